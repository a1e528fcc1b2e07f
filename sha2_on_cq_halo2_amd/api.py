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
            # proving keys first: a key refers to the params / table config it was built on until it is destroyed
            for ch in sorted(list(self._children), key=lambda o: 0 if type(o).__name__ == "ProvingKey" else 1):
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


def _ctx_set_msm_table_window(self, bits: int):
    """Window width of the per-window tables built from now on (8..20, 0 = automatic); see cq_msm_set_table_window."""
    self._chk(self.lib.cq_msm_set_table_window(self.h, bits))


def _ctx_permute_expression_pair(self, k: int, input_values: np.ndarray, table_values: np.ndarray):
    """`permute_expression_pair` (plonk/lookup/prover.rs:400-502) on the device, without the blinding rows: returns the
    permuted input and table (Montgomery limbs, `usable` rows each).  Raises on an input value the table does not hold."""
    a, t = _fr(input_values), _fr(table_values)
    assert a.shape == t.shape
    usable = a.shape[0]
    da, dt = self.to_device(a), self.to_device(t)
    oa, ot = self.alloc(usable * 32), self.alloc(usable * 32)
    self._chk(self.lib.cq_permute_expression_pair_dev(self.h, k, usable, da.ptr, dt.ptr, oa.ptr, ot.ptr))
    return oa.download((usable, 4), np.uint64), ot.download((usable, 4), np.uint64)


Context.set_msm_table_window = _ctx_set_msm_table_window
Context.permute_expression_pair = _ctx_permute_expression_pair
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
PROF_MSM_ENTRIES = 3


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


# ---- CQ static lookup objects and the prover ----------------------------------------------------------
class _Handle:
    _destroy = None

    def close(self):
        if self.h and self.ctx.h:
            getattr(self.ctx.lib, self._destroy)(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TableConfig(_Handle):
    """`StaticTableConfig` (plonk/static_lookup.rs:47-66)."""

    _destroy = "cq_table_config_destroy"

    def __init__(self, ctx: Context, g1_lagrange: np.ndarray, g_lagrange_opening_at_0: np.ndarray):
        a, b = _g1(g1_lagrange), _g1(g_lagrange_opening_at_0)
        assert a.shape == b.shape
        self.ctx, self.size = ctx, a.shape[0]
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_table_config_create(ctx.h, self.size, a.ctypes.data, b.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    @classmethod
    def setup_from_toxic_waste(cls, ctx: Context, size: int, s: np.ndarray) -> "TableConfig":
        self = cls.__new__(cls)
        self.ctx, self.size = ctx, size
        sm = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_table_config_setup_from_toxic_waste(ctx.h, size, sm.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        return self

    def download(self):
        a = np.empty((self.size, 8), dtype=np.uint64)
        b = np.empty((self.size, 8), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_table_config_download(self.h, a.ctypes.data, b.ctypes.data))
        return a, b


class StaticTable(_Handle):
    """`StaticTableValues` (plonk/static_lookup.rs:68-75)."""

    _destroy = "cq_static_table_destroy"

    def __init__(self, ctx: Context, values: np.ndarray, qs_affine: np.ndarray):
        v, q = _fr(values), _g1(qs_affine)
        assert v.shape[0] == q.shape[0]
        self.ctx, self.size = ctx, v.shape[0]
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_static_table_create(ctx.h, self.size, v.ctypes.data, q.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    @classmethod
    def setup_from_toxic_waste(cls, ctx: Context, values: np.ndarray, s: np.ndarray) -> "StaticTable":
        self = cls.__new__(cls)
        v = _fr(values)
        self.ctx, self.size = ctx, v.shape[0]
        sm = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        h = C.c_void_p()
        ctx._chk(ctx.lib.cq_static_table_setup_from_toxic_waste(ctx.h, self.size, v.ctypes.data, sm.ctypes.data, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        return self

    def download_qs(self) -> np.ndarray:
        q = np.empty((self.size, 8), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_static_table_download_qs(self.h, q.ctypes.data))
        return q


class _CqPlonk(C.Structure):
    _fields_ = [
        ("num_fixed", C.c_uint32),
        ("num_instance", C.c_uint32),
        ("fixed", C.POINTER(C.c_void_p)),
        ("cs_degree", C.c_uint32),
        ("blinding_factors", C.c_uint32),
        ("num_advice_queries", C.c_uint32),
        ("advice_query_columns", C.POINTER(C.c_uint32)),
        ("advice_query_rotations", C.POINTER(C.c_int32)),
        ("num_fixed_queries", C.c_uint32),
        ("fixed_query_columns", C.POINTER(C.c_uint32)),
        ("fixed_query_rotations", C.POINTER(C.c_int32)),
        ("num_gate_polys", C.c_uint32),
        ("gate_program_lens", C.POINTER(C.c_uint32)),
        ("gate_programs", C.POINTER(C.c_uint32)),
        ("num_constants", C.c_uint32),
        ("constants", C.POINTER(C.c_uint64)),
        ("num_perm_columns", C.c_uint32),
        ("perm_column_kinds", C.POINTER(C.c_uint32)),
        ("perm_column_indices", C.POINTER(C.c_uint32)),
        ("perm_mapping", C.POINTER(C.c_uint32)),
        ("lookup_input_program_lens", C.POINTER(C.c_uint32)),
        ("lookup_input_programs", C.POINTER(C.c_uint32)),
        ("num_legacy_lookups", C.c_uint32),
        ("legacy_lookup_widths", C.POINTER(C.c_uint32)),
        ("legacy_program_lens", C.POINTER(C.c_uint32)),
        ("legacy_programs", C.POINTER(C.c_uint32)),
        ("advice_column_phases", C.POINTER(C.c_uint8)),
        ("num_challenges", C.c_uint32),
        ("challenge_phases", C.POINTER(C.c_uint8)),
    ]


class _CqCircuit(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("num_advice", C.c_uint32),
        ("num_lookups", C.c_uint32),
        ("lookup_widths", C.POINTER(C.c_uint32)),
        ("lookup_columns", C.POINTER(C.c_uint32)),
        ("lookup_tables", C.POINTER(C.c_void_p)),
        ("vk_repr", C.c_uint64 * 4),
        ("plonk", C.POINTER(_CqPlonk)),
    ]


def _lower_plonk(cs, fixed, mapping, keep: list, from_raw: bool = False) -> _CqPlonk:
    """`ConstraintSystem` (+ the keygen outputs `fixed_values`, `Assembly.mapping`) -> cq_plonk arrays.
    from_raw: the key's polynomials come from a serialized ProvingKey, only the shape is lowered."""
    pl = _CqPlonk()
    n_fixed = cs.num_fixed_columns
    fixed = [_fr(f) for f in (fixed or [])]
    pl.num_fixed, pl.num_instance = n_fixed, cs.num_instance_columns
    if not from_raw:
        assert len(fixed) == n_fixed, "one value column per fixed column (pk.fixed_values)"
        fptrs = (C.c_void_p * max(n_fixed, 1))(*[f.ctypes.data for f in fixed])
        keep += [fixed, fptrs]
        pl.fixed = C.cast(fptrs, C.POINTER(C.c_void_p))
    pl.cs_degree, pl.blinding_factors = cs.degree(), cs.blinding_factors()

    def u32(vals):
        a = (C.c_uint32 * max(len(vals), 1))(*vals)
        keep.append(a)
        return C.cast(a, C.POINTER(C.c_uint32))

    def i32(vals):
        a = (C.c_int32 * max(len(vals), 1))(*vals)
        keep.append(a)
        return C.cast(a, C.POINTER(C.c_int32))

    pl.num_advice_queries = len(cs.advice_queries)
    pl.advice_query_columns, pl.advice_query_rotations = u32([c for c, _ in cs.advice_queries]), i32([r for _, r in cs.advice_queries])
    pl.num_fixed_queries = len(cs.fixed_queries)
    pl.fixed_query_columns, pl.fixed_query_rotations = u32([c for c, _ in cs.fixed_queries]), i32([r for _, r in cs.fixed_queries])
    constants, lens, words = [], [], []
    for g in cs.gates:
        prog = g.compile(constants)
        lens.append(len(prog))
        words += prog
    pl.num_gate_polys, pl.gate_program_lens, pl.gate_programs = len(lens), u32(lens), u32(words)
    if cs.static_lookup_inputs is not None:  # expression-valued lookup inputs
        llens, lwords = [], []
        for e in cs.static_lookup_inputs:
            prog = e.compile(constants)
            llens.append(len(prog))
            lwords += prog
        pl.lookup_input_program_lens, pl.lookup_input_programs = u32(llens), u32(lwords)
    if cs.lookups:  # legacy lookups: per lookup the input programs, then the table programs
        widths, plens, pwords = [], [], []
        for ins, tabs in cs.lookups:
            widths.append(len(ins))
            for e in list(ins) + list(tabs):
                prog = e.compile(constants)
                plens.append(len(prog))
                pwords += prog
        pl.num_legacy_lookups = len(widths)
        pl.legacy_lookup_widths, pl.legacy_program_lens, pl.legacy_programs = u32(widths), u32(plens), u32(pwords)
    if any(cs.advice_column_phase) or cs.challenge_phase:
        ap = (C.c_uint8 * max(cs.num_advice_columns, 1))(*cs.advice_column_phase)
        cp = (C.c_uint8 * max(len(cs.challenge_phase), 1))(*cs.challenge_phase)
        keep += [ap, cp]
        pl.advice_column_phases = C.cast(ap, C.POINTER(C.c_uint8))
        pl.num_challenges, pl.challenge_phases = len(cs.challenge_phase), C.cast(cp, C.POINTER(C.c_uint8))
    cst = np.zeros((max(len(constants), 1), 4), dtype=np.uint64)
    for i, v in enumerate(constants):
        cst[i] = fr_to_mont(v)
    keep.append(cst)
    pl.num_constants, pl.constants = len(constants), cst.ctypes.data_as(C.POINTER(C.c_uint64))
    pl.num_perm_columns = len(cs.permutation_columns)
    pl.perm_column_kinds = u32([c.kind for c in cs.permutation_columns])
    pl.perm_column_indices = u32([c.index for c in cs.permutation_columns])
    if mapping is not None and len(cs.permutation_columns):  # (an Assembly over no columns has an empty mapping)
        m = np.ascontiguousarray(mapping, dtype=np.uint32)
        if m.ndim != 3 or m.shape[0] != len(cs.permutation_columns) or m.shape[2] != 2:
            raise CqError(-1, "permutation mapping: expected [columns][rows][2]")
        keep.append(m)
        pl.perm_mapping = m.ctypes.data_as(C.POINTER(C.c_uint32))
    return pl


class _BufferRng(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint64)), ("pos", C.c_size_t), ("len", C.c_size_t), ("overrun", C.c_size_t)]


class ProvingKey(_Handle):
    """The slice of `ProvingKey` (plonk.rs:291-308) that `create_proof` reads.

    lookups: list of lookups, each a list of (advice column, StaticTable).  For a general circuit pass
    `cs` (a `plonk.ConstraintSystem`: gates, fixed / instance columns, permutation columns; its static
    lookups and advice column count replace `lookups` / `num_advice`), `fixed` (pk.fixed_values) and
    `permutation` (`Assembly.mapping`); table_cfg / b0_g1_bound may be None without static lookups."""

    _destroy = "cq_pk_destroy"

    def __init__(self, ctx: Context, params: ParamsKZG, k: int, num_advice: int, lookups, table_cfg: TableConfig,
                 b0_g1_bound, vk_repr: np.ndarray, cs=None, fixed=None, permutation=None, raw: bytes = None,
                 num_selectors: int = 0, checked: bool = True):
        self.ctx, self.params, self.k = ctx, params, k
        self.cs = cs
        if cs is not None:
            num_advice, lookups = cs.num_advice_columns, cs.static_lookups
        self._keep = [params, table_cfg, [t for lk in lookups for _, t in lk]]
        widths = (C.c_uint32 * max(len(lookups), 1))(*[len(lk) for lk in lookups])
        flat_cols = [c for lk in lookups for c, _ in lk]
        flat_tabs = [t.h.value if isinstance(t.h, C.c_void_p) else t.h for lk in lookups for _, t in lk]
        cols = (C.c_uint32 * max(len(flat_cols), 1))(*flat_cols)
        tabs = (C.c_void_p * max(len(flat_tabs), 1))(*flat_tabs)
        cs_ = _CqCircuit()
        cs_.k, cs_.num_advice, cs_.num_lookups = k, num_advice, len(lookups)
        cs_.lookup_widths = C.cast(widths, C.POINTER(C.c_uint32))
        cs_.lookup_columns = C.cast(cols, C.POINTER(C.c_uint32))
        cs_.lookup_tables = C.cast(tabs, C.POINTER(C.c_void_p))
        vr = np.ascontiguousarray(vk_repr, dtype=np.uint64).reshape(4)
        for i in range(4):
            cs_.vk_repr[i] = int(vr[i])
        if cs is not None:
            pl = _lower_plonk(cs, fixed, permutation, self._keep, from_raw=raw is not None)
            self._keep.append(pl)
            cs_.plonk = C.pointer(pl)
        h = C.c_void_p()
        if b0_g1_bound is None:
            b0_ptr, on_dev = None, 0
        elif isinstance(b0_g1_bound, (int,)):
            b0_ptr, on_dev = b0_g1_bound, 1
        else:
            b0 = _g1(b0_g1_bound)
            assert b0.shape[0] == (1 << k) - 1, "b0_g1_bound must hold n-1 points (arithmetic.rs:133)"
            self._b0 = b0
            b0_ptr, on_dev = b0.ctypes.data, 0
        cfg_h = table_cfg.h if table_cfg is not None else None
        if raw is None:
            ctx._chk(ctx.lib.cq_pk_create(ctx.h, params.h, C.byref(cs_), cfg_h, b0_ptr, on_dev, C.byref(h)))
        else:  # `ProvingKey::read(reader, RawBytes | RawBytesUnchecked)` (plonk.rs:379-403)
            rb = np.frombuffer(raw, dtype=np.uint8)
            ctx._chk(ctx.lib.cq_pk_read_raw(ctx.h, params.h, C.byref(cs_), cfg_h, b0_ptr, on_dev, rb.ctypes.data, rb.shape[0],
                                            num_selectors, 1 if checked else 0, C.byref(h)))
        self.h = h
        self.num_advice = num_advice
        self._num_lookups = len(lookups)
        self._table_size = table_cfg.size if table_cfg is not None else 0
        self.usable_rows = ctx.lib.cq_pk_usable_rows(h)
        self.proof_size = ctx.lib.cq_pk_proof_size(h)
        ctx._children.add(self)

    def _run(self, fn, ptrs, rng_fn, rng_state, instances=None, on_device=False) -> bytes:
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        proof = (C.c_uint8 * self.proof_size)()
        plen = C.c_size_t()
        if instances is None:
            self.ctx._chk(fn(self.h, arr, rng_fn, rng_state, proof, self.proof_size, C.byref(plen)))
        else:  # `instances: &[&[Fr]]` (prover.rs:64)
            cols = [_fr(i) if len(i) else np.zeros((0, 4), dtype=np.uint64) for i in instances]
            iptr = (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
            ilen = (C.c_size_t * max(len(cols), 1))(*[c.shape[0] for c in cols])
            self.ctx._chk(self.ctx.lib.cq_create_proof_instances(self.h, arr, 1 if on_device else 0, iptr, ilen, rng_fn, rng_state,
                                                                 proof, self.proof_size, C.byref(plen)))
        return bytes(proof[: plen.value])

    def to_bytes(self, selector_bits: bytes = b"", num_selectors: int = 0) -> bytes:
        """`ProvingKey::to_bytes(SerdeFormat::RawBytes)` (plonk.rs:405-410)."""
        size = self.ctx.lib.cq_pk_raw_size(self.h, num_selectors)
        buf = np.zeros(size, dtype=np.uint8)
        sel = np.frombuffer(selector_bits, dtype=np.uint8) if num_selectors else None
        w = C.c_size_t()
        self.ctx._chk(self.ctx.lib.cq_pk_write_raw(self.h, sel.ctypes.data if sel is not None else None, num_selectors,
                                                   buf.ctypes.data, size, C.byref(w)))
        return buf[: w.value].tobytes()

    def set_opener(self, name: str):
        """`P: Prover` of create_proof: "gwc" (ProverGWC, default) or "shplonk" (ProverSHPLONK)."""
        self.ctx._chk(self.ctx.lib.cq_pk_set_opener(self.h, {"gwc": 0, "shplonk": 1}[name]))
        self.proof_size = self.ctx.lib.cq_pk_proof_size(self.h)

    def vk_commitments(self):
        """(fixed_commitments, permutation commitments) of the matching verifying key: uint64[.,8] affine points
        (keygen.rs:247-250, permutation/keygen.rs:115-149)."""
        nf = self.cs.num_fixed_columns if self.cs is not None else 0
        npc = len(self.cs.permutation_columns) if self.cs is not None else 0
        f = np.zeros((max(nf, 1), 8), dtype=np.uint64)
        p = np.zeros((max(npc, 1), 8), dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_pk_vk_commitments(self.h, f.ctypes.data, p.ctypes.data))
        return f[:nf], p[:npc]

    def set_rng_fill(self, name):
        """cq_pk_set_rng_fill: None (per-word callbacks) or "opaque" (the harness generator's bulk form)."""
        fn = C.cast(self.ctx.lib.cq_opaque_rng_fill, C.c_void_p) if name == "opaque" else None
        self.ctx._chk(self.ctx.lib.cq_pk_set_rng_fill(self.h, fn))

    def _rng(self, rng_words=None, seed=None, opaque=False):
        lib = self.ctx.lib
        if rng_words is not None:
            w = np.ascontiguousarray(rng_words, dtype=np.uint64)
            st = _BufferRng(w.ctypes.data_as(C.POINTER(C.c_uint64)), 0, w.shape[0], 0)
            self._rng_keep = (w, st)
            return C.cast(lib.cq_buffer_rng_next_u64, C.c_void_p), C.cast(C.byref(st), C.c_void_p)
        st = (C.c_uint64 * 4)()
        lib.cq_xoshiro256ss_seed(seed, st)
        self._rng_keep = st
        # opaque: the same generator behind a function pointer the library does not recognise (a caller's RngCore)
        return C.cast(lib.cq_opaque_rng_next_u64 if opaque else lib.cq_xoshiro256ss_next_u64, C.c_void_p), C.cast(st, C.c_void_p)

    def create_proof(self, advice, rng_words=None, seed=None, instances=None) -> bytes:
        """`create_proof` (plonk/prover.rs:51) with host advice columns (uint64[n,4] each; rows
        beyond the usable rows are ignored) and public inputs `instances` (uint64[len,4] per instance
        column).  RNG: a pre-drawn u64 stream or a xoshiro256** seed."""
        cols = [np.ascontiguousarray(a, dtype=np.uint64) for a in advice]
        assert len(cols) == self.num_advice and all(c.shape == (1 << self.k, 4) for c in cols)
        fn, st = self._rng(rng_words, seed)
        return self._run(self.ctx.lib.cq_create_proof_host, [c.ctypes.data for c in cols], fn, st, instances)

    def create_proof_phases(self, advice_bufs, phase_fn, rng_words=None, seed=None, instances=None) -> bytes:
        """Multi-phase circuits (prover.rs:436-463): `advice_bufs` are DevBufs of 2^k elements; before phase p > 0 is
        committed `phase_fn(p, challenges)` is called with the user challenges so far (Python ints, canonical) and
        returns {advice column index: uint64[usable_rows.., 4] Montgomery limbs} for the columns of that phase."""
        fn, st = self._rng(rng_words, seed)
        ptrs = [b.ptr for b in advice_bufs]
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        nch = len(self.cs.challenge_phase)

        def _cb(_user, phase, ch_ptr, _adv):
            try:
                raw = np.ctypeslib.as_array(C.cast(ch_ptr, C.POINTER(C.c_uint64)), shape=(max(nch, 1), 4))
                chal = [fr_from_mont(raw[i]) for i in range(nch)]
                for col, vals in phase_fn(phase, chal).items():
                    advice_bufs[col].upload(_fr(vals))
                return 0
            except Exception:  # the C side maps a non-zero status to CQ_ERR_ARG
                import traceback

                traceback.print_exc()
                return 1

        cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p)(_cb)
        cols = [_fr(i) if len(i) else np.zeros((0, 4), dtype=np.uint64) for i in (instances or [])]
        iptr = (C.c_void_p * max(len(cols), 1))(*[c_.ctypes.data for c_ in cols])
        ilen = (C.c_size_t * max(len(cols), 1))(*[c_.shape[0] for c_ in cols])
        proof = (C.c_uint8 * self.proof_size)()
        plen = C.c_size_t()
        self.ctx._chk(self.ctx.lib.cq_create_proof_phases(self.h, arr, iptr, ilen, cb, None, fn, st, proof, self.proof_size, C.byref(plen)))
        return bytes(proof[: plen.value])

    def create_proof_dev(self, advice_ptrs, rng_words=None, seed=None, instances=None, opaque_rng=False) -> bytes:
        fn, st = self._rng(rng_words, seed, opaque_rng)
        return self._run(self.ctx.lib.cq_create_proof, list(advice_ptrs), fn, st, instances, on_device=True)


def _pk_shape(self):
    L = len(self.cs.static_lookups) if self.cs is not None else self._num_lookups
    return L, 1 << self.k, self._table_size


def _pk_cq_round1(self, advice_ptrs, theta, instance_ptrs=None, challenges=None):
    """`static_lookup::Argument::commit` (static_lookup/prover.rs:51-183) on its own (cq_cq_round1_dev): returns
    (f DevBuf [L x n], m DevBuf [L x N uint32], commitments uint64[L, 2, 8] = (f_cm, m_cm) per lookup)."""
    L, n, N = _pk_shape(self)
    ctx = self.ctx
    f, m = ctx.alloc(max(L * n * 32, 32)), ctx.alloc(max(L * N * 4, 32))
    arr = (C.c_void_p * max(len(advice_ptrs), 1))(*advice_ptrs)
    iarr = (C.c_void_p * max(len(instance_ptrs or []), 1))(*(instance_ptrs or [])) if instance_ptrs else None
    ch = np.ascontiguousarray(challenges, dtype=np.uint64) if challenges is not None else None
    th = np.ascontiguousarray(theta, dtype=np.uint64).reshape(4)
    cm = np.zeros((max(L, 1), 2, 8), dtype=np.uint64)
    ctx._chk(ctx.lib.cq_cq_round1_dev(self.h, arr, iarr, ch.ctypes.data if ch is not None else None, th.ctypes.data, f.ptr, m.ptr,
                                      cm.ctypes.data))
    return f, m, cm[:L]


def _pk_cq_round2(self, f: DevBuf, m: DevBuf, theta, beta):
    """`Committed::commit_log_derivatives` (static_lookup/prover.rs:187-342) on its own (cq_cq_round2_dev): returns
    (b_coeff DevBuf, f_coeff DevBuf, commitments uint64[L, 5, 8] = (a, q_a, a_0, b_0, p), a_at_zero uint64[L, 4])."""
    L, n, _ = _pk_shape(self)
    ctx = self.ctx
    b, fc = ctx.alloc(max(L * n * 32, 32)), ctx.alloc(max(L * n * 32, 32))
    th = np.ascontiguousarray(theta, dtype=np.uint64).reshape(4)
    be = np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)
    cm = np.zeros((max(L, 1), 5, 8), dtype=np.uint64)
    a0 = np.zeros((max(L, 1), 4), dtype=np.uint64)
    ctx._chk(ctx.lib.cq_cq_round2_dev(self.h, f.ptr, m.ptr, th.ctypes.data, be.ctypes.data, b.ptr, fc.ptr, cm.ctypes.data, a0.ctypes.data))
    return b, fc, cm[:L], a0[:L]


def _pk_cq_quotient(self, b_coeff: DevBuf, f_coeff: DevBuf, y, beta, h_in: DevBuf = None, divide: bool = True, ext: int = None) -> DevBuf:
    """The static-lookup terms of `Evaluator::evaluate_h` (evaluation.rs:533-548), optionally followed by
    `divide_by_vanishing_poly` (domain.rs:319-338) (cq_quotient_dev); `ext` = 2^extended_k elements come back."""
    ctx = self.ctx
    out = ctx.alloc(ext * 32)
    yy = np.ascontiguousarray(y, dtype=np.uint64).reshape(4)
    be = np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)
    ctx._chk(ctx.lib.cq_quotient_dev(self.h, b_coeff.ptr, f_coeff.ptr, yy.ctypes.data, be.ctypes.data, h_in.ptr if h_in else None,
                                     1 if divide else 0, out.ptr))
    return out


def _pk_create_proof_batch(self, advice_ptr_lists, seeds, lanes: int = 0, opaque_rng: bool = False):
    """cq_create_proof_batch: one proof per entry of `advice_ptr_lists` (each a list of num_advice device pointers), blinded
    from xoshiro256** streams seeded with `seeds[i]`; returns the list of proofs."""
    lib = self.ctx.lib
    B = len(advice_ptr_lists)
    assert len(seeds) == B
    cols = [(C.c_void_p * max(len(p), 1))(*p) for p in advice_ptr_lists]
    adv = (C.c_void_p * max(B, 1))(*[C.cast(c, C.c_void_p) for c in cols])
    states = [(C.c_uint64 * 4)() for _ in range(B)]
    for st, sd in zip(states, seeds):
        lib.cq_xoshiro256ss_seed(sd, st)
    st_arr = (C.c_void_p * max(B, 1))(*[C.cast(s_, C.c_void_p) for s_ in states])
    bufs = [(C.c_uint8 * self.proof_size)() for _ in range(B)]
    out = (C.c_void_p * max(B, 1))(*[C.cast(b, C.c_void_p) for b in bufs])
    lens = (C.c_size_t * max(B, 1))()
    fn = C.cast(lib.cq_opaque_rng_next_u64 if opaque_rng else lib.cq_xoshiro256ss_next_u64, C.c_void_p)
    self.ctx._chk(lib.cq_create_proof_batch(self.h, B, adv, fn, st_arr, out, self.proof_size, lens, lanes))
    return [bytes(bufs[i][: lens[i]]) for i in range(B)]


ProvingKey.create_proof_batch = _pk_create_proof_batch
ProvingKey.cq_round1 = _pk_cq_round1
ProvingKey.cq_round2 = _pk_cq_round2
ProvingKey.cq_quotient = _pk_cq_quotient


def _ctx_msm_precompute(self, bases_ptr: int, n: int):
    """Registers per-window tables for a device-resident base array (fixed-base acceleration)."""
    self._chk(self.lib.cq_msm_precompute_dev(self.h, bases_ptr, n))


def _ctx_set_msm_precompute(self, on: bool):
    self._chk(self.lib.cq_msm_set_precompute(self.h, 1 if on else 0))


Context.msm_precompute = _ctx_msm_precompute
Context.set_msm_precompute = _ctx_set_msm_precompute


def _static_table_new(cls, ctx: Context, values: np.ndarray, srs_g1: np.ndarray) -> "StaticTable":
    """`StaticTableValues::new(values, srs_g1)` (static_lookup.rs:78-126), the reference's O(N^2) construction."""
    self = cls.__new__(cls)
    v, g = _fr(values), _g1(srs_g1)
    assert v.shape[0] == g.shape[0]
    self.ctx, self.size = ctx, v.shape[0]
    h = C.c_void_p()
    ctx._chk(ctx.lib.cq_static_table_new(ctx.h, self.size, v.ctypes.data, g.ctypes.data, C.byref(h)))
    self.h = h
    ctx._children.add(self)
    return self


StaticTable.new = classmethod(_static_table_new)


def _static_table_new_fk(cls, ctx: Context, values: np.ndarray, srs_g1: np.ndarray) -> "StaticTable":
    """Same table as `StaticTable.new` (bit-identical cached quotients), built FK-style in O(N log N) group operations."""
    self = cls.__new__(cls)
    v, g = _fr(values), _g1(srs_g1)
    assert v.shape[0] == g.shape[0]
    self.ctx, self.size = ctx, v.shape[0]
    h = C.c_void_p()
    ctx._chk(ctx.lib.cq_static_table_new_fk(ctx.h, self.size, v.ctypes.data, g.ctypes.data, C.byref(h)))
    self.h = h
    ctx._children.add(self)
    return self


StaticTable.new_fk = classmethod(_static_table_new_fk)


def _params_downsize(self, k: int) -> "ParamsKZG":
    """`ParamsKZG::downsize(k)` (kzg/commitment.rs:480-492), as a new object."""
    out = ParamsKZG.__new__(ParamsKZG)
    h = C.c_void_p()
    self.ctx._chk(self.ctx.lib.cq_params_downsize(self.h, k, C.byref(h)))
    out.ctx, out.k, out.n, out.h = self.ctx, k, 1 << k, h
    self.ctx._children.add(out)
    return out


ParamsKZG.downsize = _params_downsize


def _params_read_raw(cls, ctx: Context, data: bytes, checked: bool = True) -> "ParamsKZG":
    """`ParamsKZG::read_custom(reader, RawBytes | RawBytesUnchecked)` (kzg/commitment.rs:383-459)."""
    self = cls.__new__(cls)
    buf = np.frombuffer(data, dtype=np.uint8)
    h = C.c_void_p()
    ctx._chk(ctx.lib.cq_params_read_raw(ctx.h, buf.ctypes.data, buf.shape[0], 1 if checked else 0, C.byref(h)))
    self.ctx = ctx
    self.k = int.from_bytes(data[:4], "little")
    self.n = 1 << self.k
    self.h = h
    ctx._children.add(self)
    return self


def _params_write_raw(self) -> bytes:
    """G1 part of `ParamsKZG::write_custom(writer, RawBytes)` (kzg/commitment.rs:366-379)."""
    buf = np.zeros(4 + 128 * self.n, dtype=np.uint8)
    w = C.c_size_t()
    self.ctx._chk(self.ctx.lib.cq_params_write_raw(self.h, buf.ctypes.data, buf.shape[0], C.byref(w)))
    return buf[: w.value].tobytes()


ParamsKZG.read_raw = classmethod(_params_read_raw)
ParamsKZG.write_raw = _params_write_raw

SHA_ROT0, SHA_ROT1, SHA_MAJ, SHA_CH = 0, 1, 2, 3


def _ctx_sha_synthesis_table(self, kind: int, first: int, second: int) -> np.ndarray:
    """`create_{rot0,rot1,maj,ch}_table::<L>` (sha/src/tables.rs:105-133): uint64[rows,4] = (x,y,z,f)."""
    rows = 1 << (first + 2 * second)
    buf = self.alloc(rows * 32)
    self._chk(self.lib.cq_sha_synthesis_table_dev(self.h, kind, first, second, buf.ptr))
    return buf.download((rows, 4))


def _ctx_sha_decomposition_table(self, first: int, second: int, k_bits: int) -> np.ndarray:
    """`create_decomposition_table::<L,K>` (sha/src/tables.rs:135-154): uint64[2^K,4] = (a,x,y,z)."""
    rows = 1 << k_bits
    buf = self.alloc(rows * 32)
    self._chk(self.lib.cq_sha_decomposition_table_dev(self.h, first, second, k_bits, buf.ptr))
    return buf.download((rows, 4))


Context.sha_synthesis_table = _ctx_sha_synthesis_table
Context.sha_decomposition_table = _ctx_sha_decomposition_table


_ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_BCAST_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32)
RCCL_UNIQUE_ID_BYTES = 128


def rccl_unique_id() -> bytes:
    """cq_rccl_unique_id: drawn by ONE rank and handed to the others (e.g. with a torch.distributed broadcast)."""
    buf = (C.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
    rc = load().cq_rccl_unique_id(buf)
    if rc != 0:
        raise CqError(rc, "cq_rccl_unique_id failed (librccl not loadable?)")
    return bytes(buf)


def _ctx_comm_init_rccl(self, rank: int, world: int, unique_id: bytes):
    """cq_ctx_comm_init_rccl: the context's RCCL communicator (collective: every rank calls it with the same id)."""
    assert len(unique_id) == RCCL_UNIQUE_ID_BYTES
    buf = (C.c_uint8 * RCCL_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
    self._chk(self.lib.cq_ctx_comm_init_rccl(self.h, rank, world, buf))


def _ctx_comm_init_from_torch(self, device=None, group=None):
    """Convenience for torch.distributed jobs: rank 0 draws the RCCL id, the process group broadcasts it (a CUDA tensor
    over the "nccl" backend, a CPU tensor over gloo), every rank initialises the context's communicator."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t = torch.zeros(RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8).clone()
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0, group=group)
    self.comm_init_rccl(rank, world, bytes(t.cpu().numpy().tobytes()))


def _ctx_comm_destroy(self):
    self._chk(self.lib.cq_ctx_comm_destroy(self.h))


def _ctx_msm_table_width(self, bases_ptr: int, n: int, preferred: int = 0) -> int:
    """Window width of the precomputed tables a multiexp over [bases_ptr, bases_ptr + n) would use (0: none)."""
    bits = C.c_uint32()
    self._chk(self.lib.cq_msm_table_width_dev(self.h, bases_ptr, n, preferred, C.byref(bits)))
    return bits.value


Context.msm_table_width = _ctx_msm_table_width
Context.set_hip_graphs = lambda self, on: self._chk(self.lib.cq_ctx_set_hip_graphs(self.h, 1 if on else 0))
Context.comm_init_rccl = _ctx_comm_init_rccl
Context.comm_init_from_torch = _ctx_comm_init_from_torch
Context.comm_destroy = _ctx_comm_destroy
Context.comm_selftest = lambda self: self._chk(self.lib.cq_ctx_comm_selftest(self.h))


class _CqXfer(C.Structure):
    _fields_ = [("peer", C.c_uint32), ("send", C.c_uint32), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


_EXCHANGE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_CqXfer), C.c_size_t)


def _pk_set_sharding(self, rank: int, world: int, group=None, device=None, transport: str = "callback", columns: bool = True,
                     resident: bool = False):
    """Shards `create_proof` across the ranks of a job (cq_pk_set_sharding + cq_pk_set_column_sharding): every
    commitment by point range (all-gather of the per-rank Jacobian partials) and, with `columns`, the independent column
    transforms by owner (broadcast of the transformed columns).
    transport "rccl": the library's own ncclAllGather / ncclBroadcast on device buffers, on the context's communicator
    (Context.comm_init_rccl / comm_init_from_torch first); "callback": torch.distributed collectives of `group` on host
    buffers (gloo on CPU tensors; with `device` the tensors travel through that device for the "nccl" backend).
    `resident`: cq_pk_set_resident_sharding -- transformed columns stay on their owner and slices travel point to point
    (ncclSend / ncclRecv, or torch.distributed isend / irecv on host buffers)."""
    lib = self.ctx.lib
    if world <= 1:
        # (the hooks first: the library must not keep a pointer to a callback object that is about to be dropped)
        self.ctx._chk(lib.cq_pk_set_resident_sharding(self.h, 1 if resident else 0, None, None))
        self.ctx._chk(lib.cq_pk_set_column_sharding(self.h, 1 if columns else 0, None, None))
        self.ctx._chk(lib.cq_pk_set_sharding(self.h, 0, 1, None, None))
        self._allgather_cb = self._bcast_cb = self._exchange_cb = None
        return
    if transport == "rccl":
        self._allgather_cb = self._bcast_cb = self._exchange_cb = None
        self.ctx._chk(lib.cq_pk_set_resident_sharding(self.h, 1 if resident else 0, None, None))
        self.ctx._chk(lib.cq_pk_set_column_sharding(self.h, 1 if columns else 0, None, None))
        self.ctx._chk(lib.cq_pk_set_sharding(self.h, rank, world, None, None))
        return
    import torch
    import torch.distributed as dist

    def allgather(_user, send, recv, nbytes):
        try:
            src = (C.c_uint8 * nbytes).from_address(send)
            t = torch.frombuffer(src, dtype=torch.uint8).clone()
            if device is not None:
                t = t.to(device)
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t, group=group)
            dst = (C.c_uint8 * (nbytes * world)).from_address(recv)
            flat = torch.cat([o.cpu() for o in outs]).numpy()
            C.memmove(dst, flat.ctypes.data, nbytes * world)
            return 0
        except Exception:  # never unwind into C
            return -1

    def bcast(_user, buf, nbytes, root):
        try:
            mem = (C.c_uint8 * nbytes).from_address(buf)
            t = torch.frombuffer(mem, dtype=torch.uint8)  # shares the library's host buffer
            if device is not None:
                d = t.to(device)
                dist.broadcast(d, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
                t.copy_(d.cpu())
            else:
                dist.broadcast(t, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
            return 0
        except Exception:
            return -1

    def exchange(_user, xfers, count):
        try:
            ops, keep = [], []
            for i in range(count):
                x = xfers[i]
                mem = (C.c_uint8 * x.bytes).from_address(x.buf)
                t = torch.frombuffer(mem, dtype=torch.uint8)  # shares the library's host buffer
                peer = dist.get_global_rank(group, x.peer) if group is not None else x.peer
                if device is not None:
                    d = t.to(device) if x.send else torch.empty(x.bytes, dtype=torch.uint8, device=device)
                    keep.append((None if x.send else t, d))
                    ops.append(dist.P2POp(dist.isend if x.send else dist.irecv, d, peer, group))
                else:
                    ops.append(dist.P2POp(dist.isend if x.send else dist.irecv, t, peer, group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for t, d in keep:
                if t is not None:
                    t.copy_(d.cpu())
            return 0
        except Exception:
            import traceback

            traceback.print_exc()
            return -1

    self._allgather_cb = _ALLGATHER_T(allgather)
    self._bcast_cb = _BCAST_T(bcast)
    self._exchange_cb = _EXCHANGE_T(exchange)
    self.ctx._chk(self.ctx.lib.cq_pk_set_resident_sharding(self.h, 1 if resident else 0, C.cast(self._exchange_cb, C.c_void_p), None))
    self.ctx._chk(self.ctx.lib.cq_pk_set_column_sharding(self.h, 1 if columns else 0, C.cast(self._bcast_cb, C.c_void_p), None))
    self.ctx._chk(self.ctx.lib.cq_pk_set_sharding(self.h, rank, world, C.cast(self._allgather_cb, C.c_void_p), None))


ProvingKey.set_sharding = _pk_set_sharding
