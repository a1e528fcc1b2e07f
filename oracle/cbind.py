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


def build_native() -> str:
    """The same source built `-O3 -march=native` ON THE MACHINE THAT RUNS IT (the CPU baseline's
    flags): the portable library above is built in the authoring container and travels, so it may not assume the GPU box's
    instruction set; this one is keyed by the host's CPU flags and never reused elsewhere.  Falls back to the portable
    build when no compiler is at hand."""
    import hashlib

    try:
        flags = next(line for line in open("/proc/cpuinfo") if line.startswith("flags"))
    except (OSError, StopIteration):
        flags = "unknown"
    src = os.path.join(HERE, "cq_oracle.c")
    tag = hashlib.sha256((flags + open(src).read()).encode()).hexdigest()[:10]
    out = os.path.join(HERE, "libcq_oracle_native_%s.so" % tag)
    if not os.path.exists(out):
        r = subprocess.run(["make", "-C", HERE, "-B", "native", "NATIVE_OUT=" + out], capture_output=True)
        if r.returncode != 0 or not os.path.exists(out):
            return build()
    return out


def lib(native: bool = False) -> C.CDLL:
    """native=True (bench.py's cpu_baseline leg): the -march=native build of this host replaces the loaded library.
    CQ_ORACLE_LIB in the environment names another build of the same source (the sanitizer pass: `make asan`)."""
    global _lib
    if native and not getattr(_lib, "_cq_native", False):
        return _load(build_native(), True)
    if _lib is None:
        override = os.environ.get("CQ_ORACLE_LIB")
        if override:
            return _load(override, False)
        if not os.path.exists(LIB):
            build()
        return _load(LIB, False)
    return _lib


def _load(path: str, native: bool) -> C.CDLL:
    global _lib
    if True:
        _lib = C.CDLL(path)
        _lib._cq_native = native
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
        _lib.cqo_pairing_product.restype = C.c_int
        _lib.cqo_pairing_product.argtypes = [vp, vp, C.c_size_t, vp]
        _lib.cqo_g2_mul.restype = C.c_int
        _lib.cqo_g2_mul.argtypes = [vp, vp]
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


def keygen_l_active(k: int, blinding_factors: int) -> np.ndarray:
    """keygen.rs:344-373: l_active_row on the extended coset (degree-3 circuit => 2n values)."""
    L = lib()
    L.cqo_keygen_l_active.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.zeros((2 << k, 4), dtype=np.uint64)
    L.cqo_keygen_l_active(k, blinding_factors, out.ctypes.data)
    return out


def xoshiro_state(seed: int) -> np.ndarray:
    """splitmix64-seeded xoshiro256** state (same as oracle.bn254.Xoshiro256ss)."""
    from .bn254 import Xoshiro256ss

    return np.array(Xoshiro256ss(seed).s, dtype=np.uint64)


def create_proof(k, num_advice, lookups, table_values, table_qs, g, g_lagrange, t_g1_lagrange, t_open0, b0_bound,
                 l_active, vk_repr, advice, seed) -> bytes:
    """C restatement of create_proof (CQ-only).  lookups: list of lists of (advice col, table index);
    table_values / table_qs: lists of uint64[N,4] / uint64[N,8]; advice: list of uint64[n,4]."""
    L = lib()
    vp = C.c_void_p
    L.cqo_create_proof.restype = C.c_int
    L.cqo_create_proof.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_size_t] + [vp] * 11 + [
        vp, C.POINTER(C.c_size_t)]
    widths = np.array([len(lk) for lk in lookups] or [0], dtype=np.uint32)
    cols = np.array([c for lk in lookups for c, _ in lk] or [0], dtype=np.uint32)
    tids = np.array([t for lk in lookups for _, t in lk] or [0], dtype=np.uint32)
    N = table_values[0].shape[0]
    tv = _a(np.concatenate([_a(t).reshape(N, 4) for t in table_values]))
    tq = _a(np.concatenate([_a(t).reshape(N, 8) for t in table_qs]))
    adv = _a(np.concatenate([_a(a).reshape(1 << k, 4) for a in advice])) if num_advice else np.zeros((1, 4), dtype=np.uint64)
    g, g_lagrange, t_g1_lagrange, t_open0, b0_bound, l_active = map(_a, (g, g_lagrange, t_g1_lagrange, t_open0, b0_bound, l_active))
    vk = _a(vk_repr).reshape(4)
    st = xoshiro_state(seed)
    points = num_advice + 7 * len(lookups) + 1 + 2 + 1
    scalars = len({c for lk in lookups for c, _ in lk}) + 1 + 3 * len(lookups)
    proof = np.zeros(32 * (points + scalars) + 64, dtype=np.uint8)
    plen = C.c_size_t()
    rc = L.cqo_create_proof(k, num_advice, len(lookups), widths.ctypes.data, cols.ctypes.data, tids.ctypes.data,
                            len(table_values), N, tv.ctypes.data, tq.ctypes.data, g.ctypes.data, g_lagrange.ctypes.data,
                            t_g1_lagrange.ctypes.data, t_open0.ctypes.data, b0_bound.ctypes.data, l_active.ctypes.data,
                            vk.ctypes.data, adv.ctypes.data, st.ctypes.data, proof.ctypes.data, C.byref(plen))
    if rc != 0:
        raise RuntimeError(f"cqo_create_proof failed: {rc}")
    return bytes(proof[: plen.value])


def pairing_product(g1_points, g2_scalars, want_value: bool = False):
    """prod_i e(P_i, [k_i]_2) with one final exponentiation (cq_oracle.c, the pairing section): `g1_points` affine
    (x, y) canonical ints or None, `g2_scalars` ints.  Returns True iff the product is one; with want_value also the
    12 coefficients of the value (canonical ints, the basis of oracle/pairing.py's FQ12)."""
    from . import bn254 as B

    pts = np.zeros((len(g1_points), 8), dtype=np.uint64)
    for i, pt in enumerate(g1_points):
        if pt is not None:
            pts[i] = B.points_to_mont_limbs([pt])[0]
    ks = np.zeros((len(g2_scalars), 4), dtype=np.uint64)
    for i, k in enumerate(g2_scalars):
        k %= B.R_MOD
        ks[i] = [(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    out = np.zeros((12, 4), dtype=np.uint64)
    ok = bool(lib().cqo_pairing_product(pts.ctypes.data, ks.ctypes.data, len(g1_points), out.ctypes.data))
    if not want_value:
        return ok
    rinv = pow(1 << 256, -1, B.Q_MOD)
    vals = [sum(int(out[c, j]) << (64 * j) for j in range(4)) * rinv % B.Q_MOD for c in range(12)]
    return ok, vals
