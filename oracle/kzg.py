"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the reference's KZG SRS / CQ table
preprocessing (`poly/kzg/commitment.rs`, `plonk/static_lookup.rs`).  G1 only:
the G2 side (table commitment `t`, `zv`, `x_b0_bound`) is verifier material and
out of scope.
"""
from __future__ import annotations

from .bn254 import Q_MOD
from .bn254 import (
    FR_ROOT_OF_UNITY_INV,
    FR_S,
    G1_GEN,
    JAC_ID,
    R_MOD,
    batch_to_affine,
    g1_add,
    g1_mul,
    g1_neg,
    inv_mod,
    jac_mul,
    to_jac,
)
from .poly import EvaluationDomain, best_multiexp, kate_division
from .bn254 import jac_to_affine

P = R_MOD


def _root_for(k: int) -> int:
    """kzg/commitment.rs:123-128,237-240: ROOT_OF_UNITY_INV^-1 squared S-k times."""
    root = inv_mod(FR_ROOT_OF_UNITY_INV, P)
    for _ in range(k, FR_S):
        root = root * root % P
    return root


def _powers_g1(s: int, n: int):
    """[s^i]_1 for i<n (kzg/commitment.rs:82-93,218-226)."""
    out = []
    cur = 1
    for _ in range(n):
        out.append(jac_mul(to_jac(G1_GEN), cur))
        cur = cur * s % P
    return batch_to_affine(out)


def _lagrange_g1(s: int, k: int):
    """[L_i(s)]_1 via the closed form (kzg/commitment.rs:130-142,241-251)."""
    n = 1 << k
    root = _root_for(k)
    n_inv = inv_mod(n % P, P)
    multiplier = (pow(s, n, P) - 1) * n_inv % P
    out = []
    for i in range(n):
        rp = pow(root, i, P)
        scalar = multiplier * rp % P * inv_mod((s - rp) % P, P) % P
        out.append(jac_mul(to_jac(G1_GEN), scalar))
    return batch_to_affine(out)


def g_to_lagrange(g, k: int):
    """arithmetic.rs:277-301: inverse FFT over G1 (`best_fft` with omega^-1, arithmetic.rs:171-234: bit-reversal, then
    radix-2 butterflies (a, b) -> (a + w b, a - w b)), every point times n^-1.  Affine in, affine out."""
    from .bn254 import JAC_ID, batch_to_affine, jac_add, jac_mul, to_jac

    n = 1 << k
    assert len(g) == n
    root = _root_for(k)
    omega_inv = inv_mod(root, P)
    a = [to_jac(p_) for p_ in g]
    for i in range(n):  # :186-192
        r = int(format(i, "0%db" % k)[::-1], 2) if k else 0
        if i < r:
            a[i], a[r] = a[r], a[i]
    m = 2
    while m <= n:  # :202-231 (serial shape)
        half = m // 2
        step = pow(omega_inv, n // m, P)
        for start in range(0, n, m):
            w = 1
            for j in range(half):
                t = jac_mul(a[start + j + half], w)
                u = a[start + j]
                a[start + j] = jac_add(u, t)
                neg_t = (t[0], (-t[1]) % Q_MOD, t[2])
                a[start + j + half] = jac_add(u, neg_t)
                w = w * step % P
        m *= 2
    n_inv = inv_mod(n % P, P)
    return batch_to_affine([jac_mul(p_, n_inv) for p_ in a])


class ParamsKZG:
    """kzg/commitment.rs:209-276 `setup_from_toxic_waste` (G1 part)."""

    def __init__(self, k: int, s: int):
        assert k <= FR_S
        self.k = k
        self.n = 1 << k
        self.g = _powers_g1(s, self.n)
        self.g_lagrange = _lagrange_g1(s, k)

    def commit(self, poly):
        """kzg/commitment.rs:539-543 (blind ignored)."""
        return best_multiexp(poly, self.g[: len(poly)])

    def commit_lagrange(self, poly):
        """kzg/commitment.rs:496-504 (blind ignored)."""
        return best_multiexp(poly, self.g_lagrange[: len(poly)])


class TableSRS:
    """kzg/commitment.rs:73-178 `TableSRS::setup_from_toxic_waste` (G1 part)."""

    def __init__(self, max_g1_power: int, s: int):
        g1_len = max_g1_power + 1
        assert g1_len & (g1_len - 1) == 0
        k = g1_len.bit_length() - 1
        self.size = g1_len
        self.g1 = _powers_g1(s, g1_len)
        self.g1_lagrange = _lagrange_g1(s, k)
        root = _root_for(k)
        n_inv = inv_mod(g1_len % P, P)
        # [(L_i(x) - L_i(0))/x]_1 = w^-i [L_i(x)]_1 - (1/N)[x^(N-1)]_1   (:156-170)
        last_scaled = g1_mul(self.g1[-1], n_inv)
        neg_last = g1_neg(last_scaled)
        self.g_lagrange_opening_at_0 = []
        for i, li in enumerate(self.g1_lagrange):
            w_inv_i = inv_mod(pow(root, i, P), P)
            self.g_lagrange_opening_at_0.append(g1_add(g1_mul(li, w_inv_i), neg_last))


class StaticTableValues:
    """plonk/static_lookup.rs:77-126: value->index map and cached quotients `qs`."""

    def __init__(self, values, srs_g1, qs=None):
        size = len(values)
        assert size & (size - 1) == 0
        self.size = size
        self.values = list(values)
        self.value_index_mapping = {v: i for i, v in enumerate(values)}
        assert len(self.value_index_mapping) == size  # unique values (:82-85)
        if qs is not None:
            self.qs = qs
            return
        log = size.bit_length() - 1
        dom = EvaluationDomain(2, log)
        n_inv = inv_mod(size % P, P)
        coeffs = dom.lagrange_to_coeff(values)
        self.qs = []
        g = 1
        for _ in range(size):
            quotient = kate_division(coeffs, g)
            quotient = [v * g % P * n_inv % P for v in quotient]
            self.qs.append(jac_to_affine(best_multiexp(quotient, srs_g1[: len(quotient)])))
            g = g * dom.omega % P

    @staticmethod
    def qs_closed_form(values, s: int):
        """Not in the reference: with the toxic waste known, Q_i = [(T(s)-T(w^i))/(s-w^i) * w^i/N]_1.
        Used to build large tables quickly; cross-checked against the O(N^2) path in tests."""
        size = len(values)
        log = size.bit_length() - 1
        dom = EvaluationDomain(2, log)
        coeffs = dom.lagrange_to_coeff(values)
        ts = 0
        for c in reversed(coeffs):
            ts = (ts * s + c) % P
        n_inv = inv_mod(size % P, P)
        out = []
        g = 1
        for i in range(size):
            sc = (ts - values[i]) % P * inv_mod((s - g) % P, P) % P * g % P * n_inv % P
            out.append(jac_mul(to_jac(G1_GEN), sc))
            g = g * dom.omega % P
        return batch_to_affine(out)
