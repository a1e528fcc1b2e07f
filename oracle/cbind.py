"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libcq_oracle.so (plain-C restatement).

Arrays are numpy uint64[..., 4] Montgomery limbs, byte-identical to the reference's `[u64;4]`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "cq_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "libcq_oracle.so"], check=True, capture_output=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        vp = C.c_void_p
        _lib.cqo_best_multiexp.argtypes = [vp, vp, C.c_size_t, vp]
        _lib.cqo_best_fft.argtypes = [vp, vp, C.c_uint32]
        _lib.cqo_eval_polynomial.argtypes = [vp, C.c_size_t, vp, vp]
        _lib.cqo_kate_division.argtypes = [vp, C.c_size_t, vp, vp]
        _lib.cqo_batch_invert.argtypes = [vp, C.c_size_t]
        _lib.cqo_invert_each.argtypes = [vp, C.c_size_t]
        _lib.cqo_fr_mul.argtypes = [vp, vp, vp]
        _lib.cqo_fq_mul.argtypes = [vp, vp, vp]
        _lib.cqo_fr_inv.argtypes = [vp, vp]
        _lib.cqo_g1_mul.argtypes = [vp, vp, vp]
        _lib.cqo_g1_to_affine.argtypes = [vp, vp]
        _lib.cqo_g1_add.argtypes = [vp, vp, vp]
        _lib.cqo_ifft.argtypes = [vp, vp, C.c_uint32, vp]
        _lib.cqo_distribute_powers.argtypes = [vp, C.c_size_t, vp, vp]
        _lib.cqo_mul_periodic.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        _lib.cqo_cq_quotient_term.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp]
        _lib.cqo_num_threads.restype = C.c_int
        _lib.cqo_set_num_threads.argtypes = [C.c_int]
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def best_multiexp(coeffs, bases) -> np.ndarray:
    coeffs, bases = _a(coeffs).reshape(-1, 4), _a(bases).reshape(-1, 8)
    assert coeffs.shape[0] == bases.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    lib().cqo_best_multiexp(coeffs.ctypes.data, bases.ctypes.data, coeffs.shape[0], out.ctypes.data)
    return out


def best_fft(a, omega, log_n: int) -> np.ndarray:
    a = _a(a).reshape(-1, 4).copy()
    assert a.shape[0] == 1 << log_n
    om = _a(omega).reshape(4)
    lib().cqo_best_fft(a.ctypes.data, om.ctypes.data, log_n)
    return a


def eval_polynomial(poly, x) -> np.ndarray:
    poly = _a(poly).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    xm = _a(x).reshape(4)
    lib().cqo_eval_polynomial(poly.ctypes.data, poly.shape[0], xm.ctypes.data, out.ctypes.data)
    return out


def kate_division(a, b) -> np.ndarray:
    a = _a(a).reshape(-1, 4)
    q = np.zeros((a.shape[0] - 1, 4), dtype=np.uint64)
    bm = _a(b).reshape(4)
    lib().cqo_kate_division(a.ctypes.data, a.shape[0], bm.ctypes.data, q.ctypes.data)
    return q


def batch_invert(v) -> np.ndarray:
    v = _a(v).reshape(-1, 4).copy()
    lib().cqo_batch_invert(v.ctypes.data, v.shape[0])
    return v


def invert_each(v) -> np.ndarray:
    v = _a(v).reshape(-1, 4).copy()
    lib().cqo_invert_each(v.ctypes.data, v.shape[0])
    return v


def g1_to_affine(j) -> np.ndarray:
    out = np.zeros(8, dtype=np.uint64)
    jm = _a(j).reshape(12)
    lib().cqo_g1_to_affine(jm.ctypes.data, out.ctypes.data)
    return out


def g1_mul(p_aff, scalar) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    pm, sm = _a(p_aff).reshape(8), _a(scalar).reshape(4)
    lib().cqo_g1_mul(pm.ctypes.data, sm.ctypes.data, out.ctypes.data)
    return out


def ifft(a, omega_inv, log_n, divisor) -> np.ndarray:
    a = _a(a).reshape(-1, 4).copy()
    om, dm = _a(omega_inv).reshape(4), _a(divisor).reshape(4)
    lib().cqo_ifft(a.ctypes.data, om.ctypes.data, log_n, dm.ctypes.data)
    return a
