"""Times the general-PLONK variant of the SHA-shaped circuit (gate, selector, copy constraints; bench.py's
`plonk_variant` leg) -- for rocprofv3 kernel traces.   python3 tools/prove_plonk.py [k] [gwc|shplonk] [plookup]
(`plookup`: one halo2 permutation-based lookup more -- its permute_expression_pair sorts 2 x 2^k values on the device)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload, ShaPlonkWorkload

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
opener = sys.argv[2] if len(sys.argv) > 2 else "gwc"
ctx = Context(0)
wl = ShaCqWorkload(ctx, k)
ShaPlonkWorkload.legacy_lookup = len(sys.argv) > 3 and sys.argv[3] == "plookup"
pw = ShaPlonkWorkload(ctx, k, seed=0x5348413243515F, share=wl)
pw.pk.set_opener(opener)
for i in range(4):
    t = time.time(); proof = pw.prove(seed=2 + i); dt = time.time() - t
    print("plonk variant k=%d %s: %.1f ms  proof %d bytes" % (k, opener, dt * 1e3, len(proof)), flush=True)
