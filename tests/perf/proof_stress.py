"""Long-running create_proof byte-parity stress (by hand on the GPU box): SHA-shaped CQ circuits of random size, column
count, fill level and RNG seed against the C restatement of the reference prover.
   python3 tests/perf/proof_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cbind as OC
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import fr_to_mont
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload, small_to_mont, spread16

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = Context(0)
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    k = int(rs.choice([8, 9, 10, 11, 12, 13, 14, 15]))
    pairs = int(rs.choice([1, 2, 3, 4]))
    n = 1 << k
    full = max(1, ((n - 8) * pairs) // (3 * 8 * 64))
    blocks = int(rs.choice([1, max(1, full // 4), full]))
    wl = ShaCqWorkload(ctx, k, pairs=pairs, blocks=blocks, seed=int(rs.randint(1, 1 << 30)))
    sd = int(rs.randint(1, 1 << 30))
    proof = wl.prove(seed=sd)
    g, gl = wl.params.download()
    tl, t0_ = wl.cfg.download()
    idx = np.arange(wl.cfg.size)
    cproof = OC.create_proof(k, 2 * pairs, [[(2 * p, 0), (2 * p + 1, 1)] for p in range(pairs)],
                             [small_to_mont(idx), small_to_mont(spread16(idx))], [wl.dense.download_qs(), wl.spread.download_qs()],
                             g, gl, tl, t0_, g[1:], OC.keygen_l_active(k, 5), fr_to_mont(0xC0FFEE + k),
                             [c.download((n, 4)) for c in wl.cols], sd)
    assert proof == cproof, (k, pairs, blocks, sd)
    cases += 1
print("proof stress: %d proofs byte-identical to the C oracle in %.0f s (seed %d)" % (cases, time.time() - t0, seed))
