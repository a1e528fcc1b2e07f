"""TEST INFRASTRUCTURE ONLY -- the BN254 ("bn256") optimal ate pairing in plain Python, for the acceptance oracle.

The reference verifier finishes with pairing checks (`tests/my_test.rs:249-258`, `plonk/static_lookup/verifier.rs:138-177`,
`poly/kzg/multiopen/gwc/verifier.rs:76-128`) over `halo2curves::bn256::Bn256` (`arithmetic/curves/src/bn256/engine.rs`,
`fq2.rs`, `fq6.rs`, `fq12.rs`).  This module restates the PAIRING, not the reference's tower arithmetic: extension
fields are polynomial quotient rings (Fq2 = Fq[i]/(i^2+1), Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), i = w^6 - 9, i.e.
xi = 9 + i = w^6), G2 is untwisted into E(Fq12), the Miller loop runs over 6x+2 with x = BN_X (engine.rs:18) plus the two
Frobenius line evaluations (engine.rs:430-441), and the final exponentiation is the plain power (q^12 - 1)/r.  It is
slow (about a second per pairing) and meant for the smallest proofs only.

Pins: the G2 generator and curve constant are the reference's (`bn256/curve.rs:85-129`, checked to be on the twist
and of order r); the properties the reference's own tests assert -- `test_pairing`, `random_bilinearity_tests`
(engine.rs:662-762) -- are asserted in tests/test_oracle_pairing.py.  No reference test holds a Gt value, and every
use here is of the form "product of pairings == 1", which is invariant under the choice of pairing normalisation.
"""
from __future__ import annotations

from .bn254 import G1_GEN, Q_MOD, R_MOD

Q = Q_MOD
BN_X = 4965661367192848881  # engine.rs:18
ATE_LOOP_COUNT = 6 * BN_X + 2


def _limbs(l):
    return sum(v << (64 * i) for i, v in enumerate(l))


class FQP:
    """Element of Fq[t]/(modulus); `mod_coeffs` are the low coefficients of the monic modulus."""

    mod_coeffs = ()
    degree = 0

    def __init__(self, coeffs):
        assert len(coeffs) == self.degree
        self.c = [v % Q for v in coeffs]

    @classmethod
    def one(cls):
        return cls([1] + [0] * (cls.degree - 1))

    @classmethod
    def zero(cls):
        return cls([0] * cls.degree)

    def __add__(self, o):
        return type(self)([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return type(self)([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return type(self)([-a for a in self.c])

    def __eq__(self, o):
        return isinstance(o, type(self)) and self.c == o.c

    def __mul__(self, o):
        if isinstance(o, int):
            return type(self)([a * o for a in self.c])
        d = self.degree
        b = [0] * (2 * d - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for exp in range(2 * d - 2, d - 1, -1):  # t^exp = -sum mod_coeffs[i] t^(exp - d + i)
            top = b[exp] % Q
            if top:
                for i, mc in enumerate(self.mod_coeffs):
                    if mc:
                        b[exp - d + i] -= top * mc
        return type(self)(b[:d])

    __rmul__ = __mul__

    def __pow__(self, e):
        res, base = type(self).one(), self
        while e:
            if e & 1:
                res = res * base
            base = base * base
            e >>= 1
        return res

    def inv(self):
        """Extended Euclid over Fq[t]."""
        d = self.degree
        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = self.c + [0], list(self.mod_coeffs) + [1]

        def deg(p):
            k = len(p) - 1
            while k and p[k] % Q == 0:
                k -= 1
            return k

        while deg(low):
            r = _poly_rounded_div(high, low)
            r += [0] * (d + 1 - len(r))
            nm, new = list(hm), list(high)
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] -= lm[i] * r[j]
                    new[i + j] -= low[i] * r[j]
            nm = [x % Q for x in nm]
            new = [x % Q for x in new]
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, Q)
        return type(self)([x * inv0 for x in lm[:d]])

    def __truediv__(self, o):
        return self * o.inv()


def _poly_rounded_div(a, b):
    dega = len(a) - 1
    while dega and a[dega] % Q == 0:
        dega -= 1
    degb = len(b) - 1
    while degb and b[degb] % Q == 0:
        degb -= 1
    temp = [x % Q for x in a]
    o = [0] * len(a)
    inv_lead = pow(b[degb], -1, Q)
    for i in range(dega - degb, -1, -1):
        o[i] = (o[i] + temp[degb + i] * inv_lead) % Q
        for c_ in range(degb + 1):
            temp[c_ + i] = (temp[c_ + i] - o[i] * b[c_]) % Q
    k = len(o) - 1
    while k and o[k] == 0:
        k -= 1
    return o[: k + 1]


class FQ2(FQP):
    mod_coeffs = (1, 0)  # i^2 + 1
    degree = 2


class FQ12(FQP):
    mod_coeffs = (82, 0, 0, 0, 0, 0, -18 % Q, 0, 0, 0, 0, 0)  # w^12 - 18 w^6 + 82
    degree = 12


# ---- curves ---------------------------------------------------------------------------------------------------
B2 = FQ2([_limbs([0x3267E6DC24A138E5, 0xB5B4C5E559DBEFA3, 0x81BE18991BE06AC3, 0x2B149D40CEB8AAAE]),
          _limbs([0xE4A2BD0685C315D2, 0xA74FA084E52D1852, 0xCD2CAFADEED8FDF4, 0x009713B03AF0FED4])])  # curve.rs:85-98 = 3/(9+i)
G2_GEN = (FQ2([_limbs([0x46DEBD5CD992F6ED, 0x674322D4F75EDADD, 0x426A00665E5C4479, 0x1800DEEF121F1E76]),
               _limbs([0x97E485B7AEF312C2, 0xF1AA493335A9E712, 0x7260BFB731FB5D25, 0x198E9393920D483A])]),
          FQ2([_limbs([0x4CE6CC0166FA7DAA, 0xE3D1E7690C43D37B, 0x4AAB71808DCB408F, 0x12C85EA5DB8C6DEB]),
               _limbs([0x55ACDADCD122975B, 0xBC4B313370B38EF3, 0xEC9E99AD690C3395, 0x090689D0585FF075])]))  # curve.rs:100-129


def ec_double(pt):
    if pt is None:
        return None
    x, y = pt
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    return (nx, m * (x - nx) - y)


def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return ec_double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def ec_mul(pt, k):
    acc, base = None, pt
    while k:
        if k & 1:
            acc = ec_add(acc, base)
        base = ec_double(base)
        k >>= 1
    return acc


def ec_neg(pt):
    return None if pt is None else (pt[0], -pt[1])


def g2_mul(k):
    """[k]_2 as the reference's setup computes its G2 elements (kzg/commitment.rs:253-256)."""
    return ec_mul(G2_GEN, k % R_MOD)


def is_on_twist(pt):
    x, y = pt
    return y * y == x * x * x + B2


_W = FQ12([0, 1] + [0] * 10)


def twist(pt):
    """E'(Fq2) -> E(Fq12): (x, y) -> (x w^2, y w^3) with Fq2 embedded through i = w^6 - 9."""
    if pt is None:
        return None
    x, y = pt
    xc = [x.c[0] - x.c[1] * 9, x.c[1]]
    yc = [y.c[0] - y.c[1] * 9, y.c[1]]
    nx = FQ12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = FQ12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return (nx * (_W * _W), ny * (_W * _W * _W))


def _cast_g1(pt):
    x, y = pt
    return (FQ12([x] + [0] * 11), FQ12([y] + [0] * 11))


def _linefunc(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1 * 3) / (y1 * 2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q2, p1):
    """f_{6x+2,Q}(P) times the two Frobenius lines (engine.rs:413-443), before the final exponentiation.
    q2: G2 point (FQ2 coordinates) or None; p1: G1 affine (ints) or None."""
    if q2 is None or p1 is None:
        return FQ12.one()
    qq, pp = twist(q2), _cast_g1(p1)
    r = qq
    f = FQ12.one()
    for i in range(ATE_LOOP_COUNT.bit_length() - 2, -1, -1):
        f = f * f * _linefunc(r, r, pp)
        r = ec_double(r)
        if (ATE_LOOP_COUNT >> i) & 1:
            f = f * _linefunc(r, qq, pp)
            r = ec_add(r, qq)
    q1 = (qq[0] ** Q, qq[1] ** Q)
    nq2 = (q1[0] ** Q, -(q1[1] ** Q))
    f = f * _linefunc(r, q1, pp)
    r = ec_add(r, q1)
    f = f * _linefunc(r, nq2, pp)
    return f


def final_exponentiation(f):
    return f ** ((Q ** 12 - 1) // R_MOD)


def pairing(p1, q2):
    """e(P, Q), P in G1 (affine ints or None), Q in G2."""
    return final_exponentiation(miller_loop(q2, p1))


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 with one final exponentiation (`multi_miller_loop`, engine.rs:571-640)."""
    f = FQ12.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exponentiation(f) == FQ12.one()


G1_GENERATOR = G1_GEN
