import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import fr_to_mont, fr_from_mont, FR_MODULUS
ctx = Context(0)
P = FR_MODULUS
def devmul(a, b):
    # eval [0, a] at b = a*b (Montgomery in/out)
    poly = np.stack([fr_to_mont(0), fr_to_mont(a)])
    out = ctx.eval_polynomial(poly, fr_to_mont(b))
    m = sum(int(x) << (64*i) for i, x in enumerate(out))
    return m
for a, b in [(1, 1), (1, 2), (2, 3), (5, 7), (P-1, P-1), (123456789, 987654321), ((1<<200)+5, (1<<100)+7)]:
    m = devmul(a, b)
    exp = (a*b % P) * (1 << 256) % P
    print(hex(a)[:20], hex(b)[:20], "ok" if m == exp else "BAD", hex(m), hex(exp), "diff/p=", (m-exp)/P if m!=exp else 0)
