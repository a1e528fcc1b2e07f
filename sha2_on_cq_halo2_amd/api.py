"""Host-side mirror of the reference's operator interface over the C ABI (ctypes).

Names, argument meaning and error behaviour follow the Rust functions they stand for
(halo2_proofs/src/arithmetic.rs, poly/domain.rs, poly/kzg/commitment.rs); contract
violations that panic in Rust raise `CqError` here.
"""
from __future__ import annotations

import ctypes as C

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
        if self.ptr:
            self.ctx.lib.cq_dev_free(self.ctx.h, self.ptr)
            self.ptr = None

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

    def close(self):
        if self.h:
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
        if self.h:
            self.ctx.lib.cq_params_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
