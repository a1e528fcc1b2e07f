import os
import sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import bn254 as B, kzg, poly as OP
from sha2_on_cq_halo2_amd import Context, ParamsKZG
ctx = Context(0)
s = B.fr_random(B.Xoshiro256ss(0x6371)); sm = B.to_mont_limbs([s])[0]
for k in (3, 4, 6):
    n = 1 << k
    op = kzg.ParamsKZG(k, s)
    gp = ParamsKZG.setup_from_toxic_waste(ctx, k, sm)
    for name, sc in (("e2+e5", [0,0,1,0,0,1,0,0]), ("2e2", [0,0,2]+[0]*5), ("ones", [1]*8), ("iota", list(range(8))), ("-1", [B.R_MOD-1]*8),
                     ("e0", [1]+[0]*7), ("e0+e1", [1,1]+[0]*6), ("3e1+3e2", [0,3,3]+[0]*5)):
        sc = sc + [0]*(n-8)
        exp = B.jac_to_affine(OP.best_multiexp(sc, op.g_lagrange))
        got = B.jac_from_mont_limbs(gp.commit_lagrange(B.to_mont_limbs(sc)))[0]
        print(k, name, "ok" if got == exp else "MISMATCH", flush=True)
ctx.close()
