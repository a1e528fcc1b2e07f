"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the reference's L1/L2 arithmetic
(`halo2_proofs/src/arithmetic.rs`, `poly/domain.rs`).  Canonical-int restatement;
see `oracle/bn254.py` for the import rules and pinning status.
"""
from __future__ import annotations

import math

from .bn254 import (
    FR_ROOT_OF_UNITY,
    FR_S,
    FR_ZETA,
    JAC_ID,
    R_MOD,
    inv_mod,
    jac_add,
    jac_add_affine,
    jac_double,
    jac_to_affine,
    to_repr,
)

P = R_MOD


def bitreverse(n: int, l: int) -> int:
    """arithmetic.rs:172-179."""
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def best_fft(a, omega: int, log_n: int):
    """arithmetic.rs:171-234: in-place radix-2 DIT, natural order in/out.
    `a` is a list of canonical ints; returns the same list object."""
    n = len(a)
    assert n == 1 << log_n  # arithmetic.rs:184
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    # arithmetic.rs:194-200 twiddles w^i, i < n/2
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % P
    chunk = 2
    twiddle_chunk = n // 2
    for _ in range(log_n):
        half = chunk // 2
        for start in range(0, n, chunk):
            for i in range(half):
                t = a[start + half + i] * tw[i * twiddle_chunk] % P
                u = a[start + i]
                a[start + i] = (u + t) % P
                a[start + half + i] = (u - t) % P
        chunk *= 2
        twiddle_chunk //= 2
    return a


def batch_invert(vals):
    """ff::BatchInvert (ff 0.12, absent from /root/reference): Montgomery's trick,
    zeros are left as zero.  Mathematically determined; pinned by identity tests."""
    out = list(vals)
    acc = 1
    pref = []
    for v in out:
        pref.append(acc)
        if v != 0:
            acc = acc * v % P
    acc = inv_mod(acc, P)
    for i in range(len(out) - 1, -1, -1):
        v = out[i]
        if v == 0:
            continue
        out[i] = pref[i] * acc % P
        acc = acc * v % P
    return out


def eval_polynomial(poly, x: int) -> int:
    """arithmetic.rs:304-329 (Horner; chunking does not change the value)."""
    acc = 0
    for c in reversed(poly):
        acc = (acc * x + c) % P
    return acc


def kate_division(a, b: int):
    """arithmetic.rs:351-387: (a(X) - a(b)) / (X - b); length len(a)-1."""
    nb = (-b) % P
    q = [0] * (len(a) - 1)
    tmp = 0
    for i in range(len(a) - 2, -1, -1):
        lead = (a[i + 1] - tmp) % P
        q[i] = lead
        tmp = lead * nb % P
    return q


def multiexp_window(n_bases: int) -> int:
    """arithmetic.rs:16-22."""
    if n_bases < 4:
        return 1
    if n_bases < 32:
        return 3
    return int(math.ceil(math.log(float(n_bases))))


def get_at(segment: int, c: int, repr32: bytes) -> int:
    """arithmetic.rs:24-42."""
    skip_bits = segment * c
    skip_bytes = skip_bits // 8
    if skip_bytes >= 32:
        return 0
    v = repr32[skip_bytes : skip_bytes + 8].ljust(8, b"\0")
    tmp = int.from_bytes(v, "little")
    tmp >>= skip_bits - skip_bytes * 8
    return tmp % (1 << c)


def multiexp_serial(coeffs, bases, acc=JAC_ID):
    """arithmetic.rs:13-101 (Pippenger, unsigned windows, summation by parts).
    coeffs canonical ints, bases affine (x,y)|None; returns Jacobian."""
    reprs = [to_repr(c) for c in coeffs]
    c = multiexp_window(len(bases))
    segments = 256 // c + 1
    for seg in range(segments - 1, -1, -1):
        for _ in range(c):
            acc = jac_double(acc)
        buckets = [JAC_ID] * ((1 << c) - 1)
        for r, base in zip(reprs, bases):
            d = get_at(seg, c, r)
            if d != 0:
                buckets[d - 1] = jac_add_affine(buckets[d - 1], base)
        running = JAC_ID
        for b in reversed(buckets):
            running = jac_add(running, b)
            acc = jac_add(acc, running)
    return acc


def best_multiexp(coeffs, bases):
    """arithmetic.rs:132-159.  Thread chunking only re-associates the sum; result identical."""
    assert len(coeffs) == len(bases)  # arithmetic.rs:133
    return multiexp_serial(coeffs, bases)


def msm_affine(coeffs, bases):
    return jac_to_affine(best_multiexp(coeffs, bases))


class EvaluationDomain:
    """poly/domain.rs:39-142 (constants) and :238-374 (transforms)."""

    def __init__(self, j: int, k: int):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        w = FR_ROOT_OF_UNITY
        for _ in range(ek, FR_S):
            w = w * w % P
        self.extended_omega = w
        for _ in range(k, ek):
            w = w * w % P
        self.omega = w
        self.omega_inv = inv_mod(self.omega, P)
        self.extended_omega_inv = inv_mod(self.extended_omega, P)
        self.g_coset = FR_ZETA
        self.g_coset_inv = FR_ZETA * FR_ZETA % P
        # domain.rs:84-107
        orig = pow(FR_ZETA, self.n, P)
        step = pow(self.extended_omega, self.n, P)
        t = []
        cur = orig
        while True:
            t.append(cur)
            cur = cur * step % P
            if cur == orig:
                break
        assert len(t) == 1 << (ek - k)
        self.t_evaluations = [inv_mod((v - 1) % P, P) for v in t]
        self.ifft_divisor = inv_mod(self.n % P, P)
        self.extended_ifft_divisor = inv_mod((1 << ek) % P, P)
        self.barycentric_weight = inv_mod(self.n % P, P)

    @property
    def extended_len(self) -> int:
        return 1 << self.extended_k

    @staticmethod
    def ifft(a, omega_inv: int, log_n: int, divisor: int):
        """domain.rs:366-374."""
        best_fft(a, omega_inv, log_n)
        for i in range(len(a)):
            a[i] = a[i] * divisor % P
        return a

    def lagrange_to_coeff(self, a):
        """domain.rs:238-248."""
        a = list(a)
        assert len(a) == self.n
        return self.ifft(a, self.omega_inv, self.k, self.ifft_divisor)

    def _distribute_powers_zeta(self, a, into_coset: bool):
        """domain.rs:347-363."""
        cp = [self.g_coset, self.g_coset_inv] if into_coset else [self.g_coset_inv, self.g_coset]
        for i in range(len(a)):
            m = i % 3
            if m:
                a[i] = a[i] * cp[m - 1] % P

    def coeff_to_extended(self, a):
        """domain.rs:252-266."""
        a = list(a)
        assert len(a) == self.n
        self._distribute_powers_zeta(a, True)
        a += [0] * (self.extended_len - self.n)
        return best_fft(a, self.extended_omega, self.extended_k)

    def extended_to_coeff(self, a):
        """domain.rs:293-315."""
        a = list(a)
        assert len(a) == self.extended_len
        self.ifft(a, self.extended_omega_inv, self.extended_k, self.extended_ifft_divisor)
        self._distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        """domain.rs:319-338."""
        assert len(a) == self.extended_len
        tl = len(self.t_evaluations)
        return [v * self.t_evaluations[i % tl] % P for i, v in enumerate(a)]

    def rotate_omega(self, value: int, rotation: int) -> int:
        """domain.rs:414-424."""
        if rotation >= 0:
            return value * pow(self.omega, rotation, P) % P
        return value * pow(self.omega_inv, -rotation, P) % P
