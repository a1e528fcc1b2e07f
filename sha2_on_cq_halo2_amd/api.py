"""Host-side mirror of the reference's operator interface over the C ABI (ctypes).

Names, argument meaning and error behaviour follow the Rust functions they stand for
(halo2_proofs/src/arithmetic.rs, poly/domain.rs, poly/kzg/commitment.rs); contract
violations that panic in Rust raise `CqError` here.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from ._lib import CqError, load


def _fr(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == 4, "expected uint64[n,4] Montgomery limbs"
    return a


class DevBuf:
    """Device allocation owned by a Context."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = nbytes
        p = C.c_void_p()
        ctx._chk(ctx.lib.cq_dev_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._children.add(self)

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.cq_dev_upload(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.cq_dev_download(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.cq_dev_free(self.ctx.h, self.ptr)
        self.ptr = None

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.cq_ctx_create(device, stream, C.byref(h))
        if rc != 0:
            raise CqError(rc, "cq_ctx_create failed (no usable GPU?)")
        self.h = h
        self._children = weakref.WeakSet()  # device objects that must die before the context

    def close(self):
        if self.h:
            for ch in list(self._children):
                ch.close()
            self.lib.cq_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise CqError(rc, self.lib.cq_last_error(self.h).decode())

    def sync(self):
        self._chk(self.lib.cq_ctx_sync(self.h))

    @property
    def stream(self) -> int:
        return self.lib.cq_ctx_stream(self.h)

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DevBuf:
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, max(arr.nbytes, 1)).upload(arr)

    # ---- arithmetic.rs ----
    def best_fft(self, a: np.ndarray, omega: np.ndarray, log_n: int) -> np.ndarray:
        """`best_fft(a, omega, log_n)` (arithmetic.rs:171): returns the transformed copy."""
        a = _fr(a).copy()
        if a.shape[0] != 1 << log_n:  # arithmetic.rs:184 assert
            raise CqError(-1, "best_fft: len != 1 << log_n")
        om = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
        self._chk(self.lib.cq_best_fft(self.h, a.ctypes.data, log_n, om.ctypes.data))
        return a

    def best_fft_dev(self, src: DevBuf, dst: DevBuf, omega: np.ndarray, log_n: int):
        om = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
        self._chk(self.lib.cq_best_fft_dev(self.h, src.ptr, dst.ptr, log_n, om.ctypes.data))


def _g1(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == 8, "expected uint64[n,8] affine points"
    return a


def _ctx_best_multiexp(self, coeffs: np.ndarray, bases: np.ndarray) -> np.ndarray:
    """`best_multiexp(coeffs, bases)` (arithmetic.rs:132): Jacobian result, uint64[12]."""
    coeffs = _fr(coeffs) if len(coeffs) else np.zeros((0, 4), dtype=np.uint64)
    bases = _g1(bases) if len(bases) else np.zeros((0, 8), dtype=np.uint64)
    if coeffs.shape[0] != bases.shape[0]:  # arithmetic.rs:133 assert_eq
        raise CqError(-1, "best_multiexp: coeffs.len() != bases.len()")
    out = np.zeros(12, dtype=np.uint64)
    self._chk(self.lib.cq_best_multiexp(self.h, coeffs.ctypes.data, bases.ctypes.data, coeffs.shape[0], out.ctypes.data))
    return out


def _ctx_best_multiexp_dev(self, coeffs: DevBuf, bases: DevBuf, n: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    self._chk(self.lib.cq_best_multiexp_dev(self.h, coeffs.ptr, bases.ptr, n, out.ctypes.data))
    return out


def _ctx_msm_batch_dev(self, coeff_ptrs, bases_ptr: int, n: int) -> np.ndarray:
    ptrs = (C.c_void_p * len(coeff_ptrs))(*coeff_ptrs)
    out = np.zeros((len(coeff_ptrs), 12), dtype=np.uint64)
    self._chk(self.lib.cq_msm_batch_dev(self.h, ptrs, bases_ptr, n, len(coeff_ptrs), out.ctypes.data))
    return out


def _ctx_set_msm_window(self, bits: int):
    self._chk(self.lib.cq_msm_set_window(self.h, bits))


Context.best_multiexp = _ctx_best_multiexp
Context.best_multiexp_dev = _ctx_best_multiexp_dev
Context.msm_batch_dev = _ctx_msm_batch_dev
Context.set_msm_window = _ctx_set_msm_window


class ParamsKZG:
    """`ParamsKZG<Bn256>` G1 part (poly/kzg/commitment.rs:31-39), SRS resident on the GPU."""

    @classmethod
    def setup_from_toxic_waste(cls, ctx: Context, k: int, s: np.ndarray) -> "ParamsKZG":
        """`ParamsKZG::setup_from_toxic_waste(k, s)` (commitment.rs:209-276), built on the GPU."""
        self = cls.__new__(cls)
        self.ctx, self.k, self.n = ctx, k, 1 << k
        sm = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_params_setup_from_toxic_waste(ctx.h, k, sm.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        return self

    def download(self):
        """(g, g_lagrange) as uint64[n,8] host arrays."""
        g = np.empty((self.n, 8), dtype=np.uint64)
        gl = np.empty((self.n, 8), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_dev_download(self.ctx.h, g.ctypes.data, self.g_dev, g.nbytes))
        self.ctx._chk(self.ctx.lib.cq_dev_download(self.ctx.h, gl.ctypes.data, self.g_lagrange_dev, gl.nbytes))
        return g, gl

    def __init__(self, ctx: Context, k: int, g: np.ndarray, g_lagrange: np.ndarray):
        g = _g1(g)
        g_lagrange = _g1(g_lagrange)
        assert g.shape[0] == 1 << k and g_lagrange.shape[0] == 1 << k
        self.ctx = ctx
        self.k = k
        self.n = 1 << k
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_params_create(ctx.h, k, g.ctypes.data, g_lagrange.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    @property
    def g_dev(self) -> int:
        return self.ctx.lib.cq_params_g_dev(self.h)

    @property
    def g_lagrange_dev(self) -> int:
        return self.ctx.lib.cq_params_g_lagrange_dev(self.h)

    def _commit(self, fn, poly):
        poly = _fr(poly)
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(fn(self.h, poly.ctypes.data, poly.shape[0], out.ctypes.data))
        return out

    def commit(self, poly: np.ndarray) -> np.ndarray:
        """`ParamsKZG::commit` (commitment.rs:539-543)."""
        return self._commit(self.ctx.lib.cq_commit, poly)

    def commit_lagrange(self, poly: np.ndarray) -> np.ndarray:
        """`ParamsKZG::commit_lagrange` (commitment.rs:496-504)."""
        return self._commit(self.ctx.lib.cq_commit_lagrange, poly)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.cq_params_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PROF_MSM_ACCUMULATE = 1
PROF_NTT_PASS = 2


def _ctx_profile_enable(self, on: bool = True):
    self._chk(self.lib.cq_profile_enable(self.h, 1 if on else 0))


def _ctx_profile_read(self, which: int):
    """(total milliseconds, launches) of the bracketed kernel since the last read."""
    ms = C.c_double()
    calls = C.c_uint64()
    self._chk(self.lib.cq_profile_read(self.h, which, C.byref(ms), C.byref(calls)))
    return ms.value, calls.value


Context.profile_enable = _ctx_profile_enable
Context.profile_read = _ctx_profile_read


# ---- small host-side field helpers (Python ints; constants from bn256/fr.rs:29-83) ----------------
FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_S = 28
FR_ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def fr_to_mont(v: int) -> np.ndarray:
    """canonical int -> uint64[4] Montgomery limbs (the layout every entry point takes)."""
    m = (v % FR_MODULUS) * (1 << 256) % FR_MODULUS
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fr_from_mont(limbs) -> int:
    m = sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)))
    return m * pow(1 << 256, -1, FR_MODULUS) % FR_MODULUS


def domain_omega_int(k: int) -> int:
    """2^k-th root of unity as `EvaluationDomain::new` derives it (poly/domain.rs:54-75)."""
    w = FR_ROOT_OF_UNITY
    for _ in range(k, FR_S):
        w = w * w % FR_MODULUS
    return w


def domain_omega(k: int) -> np.ndarray:
    return fr_to_mont(domain_omega_int(k))


def _ctx_eval_polynomial(self, poly: np.ndarray, point: np.ndarray) -> np.ndarray:
    """`eval_polynomial(poly, point)` (arithmetic.rs:304)."""
    poly = _fr(poly) if len(poly) else np.zeros((0, 4), dtype=np.uint64)
    pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    self._chk(self.lib.cq_eval_polynomial(self.h, poly.ctypes.data, poly.shape[0], pt.ctypes.data, out.ctypes.data))
    return out


def _ctx_kate_division(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """`kate_division(a, b)` (arithmetic.rs:351): n-1 quotient coefficients."""
    a = _fr(a)
    bm = np.ascontiguousarray(b, dtype=np.uint64).reshape(4)
    q = np.zeros((max(a.shape[0] - 1, 0), 4), dtype=np.uint64)
    self._chk(self.lib.cq_kate_division(self.h, a.ctypes.data, a.shape[0], bm.ctypes.data, q.ctypes.data))
    return q


def _ctx_batch_invert(self, a: np.ndarray) -> np.ndarray:
    """`BatchInvert::batch_invert` (ff 0.12): returns the inverted copy, zeros untouched."""
    a = _fr(a).copy() if len(a) else np.zeros((0, 4), dtype=np.uint64)
    self._chk(self.lib.cq_batch_invert(self.h, a.ctypes.data, a.shape[0]))
    return a


Context.eval_polynomial = _ctx_eval_polynomial
Context.kate_division = _ctx_kate_division
Context.batch_invert = _ctx_batch_invert


class EvaluationDomain:
    """`EvaluationDomain<Fr>` (poly/domain.rs:19-34)."""

    def __init__(self, ctx: Context, j: int, k: int):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_domain_create(ctx.h, j, k, C.byref(h)))
        self.h = h
        self.j = j
        self.k = k
        self.extended_k = ctx.lib.cq_domain_extended_k(h)
        self.n = 1 << k
        ctx._children.add(self)

    @property
    def extended_len(self) -> int:
        return 1 << self.extended_k

    def constants(self):
        arrs = [np.zeros(4, dtype=np.uint64) for _ in range(4)]
        self.ctx._chk(self.ctx.lib.cq_domain_constants(self.h, *[a.ctypes.data for a in arrs]))
        return dict(zip(("omega", "omega_inv", "extended_omega", "ifft_divisor"), arrs))

    def lagrange_to_coeff(self, a: np.ndarray) -> np.ndarray:
        a = _fr(a).copy()
        if a.shape[0] != self.n:  # domain.rs:239 assert
            raise CqError(-1, "lagrange_to_coeff: wrong length")
        self.ctx._chk(self.ctx.lib.cq_lagrange_to_coeff(self.h, a.ctypes.data))
        return a

    def coeff_to_extended(self, a: np.ndarray) -> np.ndarray:
        a = _fr(a)
        if a.shape[0] != self.n:  # domain.rs:256 assert
            raise CqError(-1, "coeff_to_extended: wrong length")
        out = np.zeros((self.extended_len, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_coeff_to_extended(self.h, a.ctypes.data, out.ctypes.data))
        return out

    def extended_to_coeff(self, a: np.ndarray) -> np.ndarray:
        a = _fr(a)
        if a.shape[0] != self.extended_len:  # domain.rs:294 assert
            raise CqError(-1, "extended_to_coeff: wrong length")
        out = np.zeros((self.n * (self.j - 1), 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_extended_to_coeff(self.h, a.ctypes.data, out.ctypes.data))
        return out

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.cq_domain_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
