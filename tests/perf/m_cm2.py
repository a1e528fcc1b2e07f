import os
import sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import bn254 as B
from tests.test_prover_gpu import _setup, _prove_both
from sha2_on_cq_halo2_amd import Context
ctx = Context(0)
tv = {"table": [0, 1, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32],
      "table_2": [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]}
for pre in (True, False):
    ctx.set_msm_precompute(pre)
    env = _setup(ctx, 3, tv, [[(0, "table"), (1, "table_2")]], 2, 0x6371)
    tr, proof = _prove_both(env, [[30, 6], [15, 3]], 7)
    names = ["adv0", "adv1", "f", "m", "a", "qa", "a0", "b0", "p", "random", "h0", "h1"]
    print("pre", pre, "equal", proof == tr.proof)
    for i, nm in enumerate(names):
        print("  ", nm, proof[32*i:32*i+32] == tr.proof[32*i:32*i+32])
ctx.close()
