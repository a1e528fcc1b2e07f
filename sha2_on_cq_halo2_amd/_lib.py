"""ctypes loader for libcq_halo2.so.  Fails loudly if the HIP extension is missing: there is
no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcq_halo2.so")
HEADER = os.path.join(HERE, "..", "include", "cq_halo2.h")

_lib = None


class CqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cq_halo2 error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m sha2_on_cq_halo2_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def header_symbols():
    """Every function name declared in include/cq_halo2.h."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cq_[a-z0-9_]+)\s*\(", src)))


u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


def _declare(lib):
    lib.cq_version.restype = C.c_char_p
    lib.cq_last_error.restype = C.c_char_p
    lib.cq_last_error.argtypes = [vp]
    lib.cq_ctx_stream.restype = vp
    lib.cq_ctx_stream.argtypes = [vp]
    lib.cq_ctx_destroy.restype = None
    lib.cq_ctx_destroy.argtypes = [vp]
    lib.cq_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.cq_ctx_sync.argtypes = [vp]
    lib.cq_ctx_set_hip_graphs.argtypes = [vp, C.c_int]
    lib.cq_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.cq_dev_free.argtypes = [vp, vp]
    lib.cq_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    lib.cq_dev_download.argtypes = [vp, vp, vp, C.c_size_t]
    lib.cq_dev_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    lib.cq_best_fft.argtypes = [vp, vp, C.c_uint32, vp]
    lib.cq_best_fft_dev.argtypes = [vp, vp, vp, C.c_uint32, vp]
    lib.cq_bench_modmul_dev.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int]


def _declare_msm(lib):
    lib.cq_best_multiexp.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.cq_best_multiexp_dev.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.cq_msm_batch_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, vp]
    lib.cq_msm_set_window.argtypes = [vp, C.c_uint32]
    lib.cq_msm_set_table_window.argtypes = [vp, C.c_uint32]
    lib.cq_permute_expression_pair_dev.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    lib.cq_params_create.argtypes = [vp, C.c_uint32, vp, vp, C.POINTER(vp)]
    lib.cq_params_destroy.restype = None
    lib.cq_params_destroy.argtypes = [vp]
    lib.cq_params_g_dev.restype = vp
    lib.cq_params_g_dev.argtypes = [vp]
    lib.cq_params_g_lagrange_dev.restype = vp
    lib.cq_params_g_lagrange_dev.argtypes = [vp]
    for name in ("cq_commit", "cq_commit_lagrange", "cq_commit_dev", "cq_commit_lagrange_dev"):
        getattr(lib, name).argtypes = [vp, vp, C.c_size_t, vp]


_declare_base = _declare


def _declare(lib):  # noqa: F811
    _declare_base(lib)
    _declare_msm(lib)
    lib.cq_params_setup_from_toxic_waste.argtypes = [vp, C.c_uint32, vp, C.POINTER(vp)]
    lib.cq_fixed_base_mul_dev.argtypes = [vp, vp, C.c_size_t, vp]
    lib.cq_eval_polynomial.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.cq_eval_polynomial_dev.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.cq_kate_division.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.cq_kate_division_dev.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.cq_batch_invert.argtypes = [vp, vp, C.c_size_t]
    lib.cq_batch_invert_dev.argtypes = [vp, vp, C.c_size_t]
    lib.cq_domain_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    lib.cq_domain_destroy.restype = None
    lib.cq_domain_destroy.argtypes = [vp]
    lib.cq_domain_k.restype = C.c_uint32
    lib.cq_domain_k.argtypes = [vp]
    lib.cq_domain_extended_k.restype = C.c_uint32
    lib.cq_domain_extended_k.argtypes = [vp]
    lib.cq_domain_constants.argtypes = [vp, vp, vp, vp, vp]
    lib.cq_lagrange_to_coeff.argtypes = [vp, vp]
    lib.cq_coeff_to_extended.argtypes = [vp, vp, vp]
    lib.cq_extended_to_coeff.argtypes = [vp, vp, vp]
    lib.cq_lagrange_to_coeff_dev.argtypes = [vp, vp, vp, C.c_uint32]
    lib.cq_coeff_to_extended_dev.argtypes = [vp, vp, vp, C.c_uint32]
    lib.cq_extended_to_coeff_dev.argtypes = [vp, vp, vp]
    lib.cq_table_config_create.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(vp)]
    lib.cq_table_config_setup_from_toxic_waste.argtypes = [vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.cq_table_config_destroy.restype = None
    lib.cq_table_config_destroy.argtypes = [vp]
    lib.cq_table_config_download.argtypes = [vp, vp, vp]
    lib.cq_static_table_create.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(vp)]
    lib.cq_static_table_setup_from_toxic_waste.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(vp)]
    lib.cq_static_table_destroy.restype = None
    lib.cq_static_table_destroy.argtypes = [vp]
    lib.cq_static_table_download_qs.argtypes = [vp, vp]
    lib.cq_pk_create.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.POINTER(vp)]
    lib.cq_pk_set_sharding.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
    lib.cq_pk_set_column_sharding.argtypes = [vp, C.c_int, vp, vp]
    lib.cq_rccl_unique_id.argtypes = [vp]
    lib.cq_ctx_comm_init_rccl.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.cq_ctx_comm_destroy.argtypes = [vp]
    lib.cq_ctx_comm_selftest.argtypes = [vp]
    lib.cq_pk_destroy.restype = None
    lib.cq_pk_destroy.argtypes = [vp]
    lib.cq_pk_usable_rows.restype = C.c_uint32
    lib.cq_pk_usable_rows.argtypes = [vp]
    lib.cq_pk_proof_size.restype = C.c_size_t
    lib.cq_pk_proof_size.argtypes = [vp]
    lib.cq_create_proof.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cq_create_proof_batch.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t, vp, C.c_uint32]
    lib.cq_create_proof_host.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cq_create_proof_instances.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cq_pk_vk_commitments.argtypes = [vp, vp, vp]
    lib.cq_cq_round1_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cq_cq_round2_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cq_quotient_dev.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.cq_g_to_lagrange_dev.argtypes = [vp, vp, C.c_uint32, vp]
    lib.cq_params_downsize.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    lib.cq_static_table_new_fk.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(vp)]
    lib.cq_create_proof_phases.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cq_pk_set_opener.argtypes = [vp, C.c_int]
    lib.cq_pk_read_raw.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.cq_pk_raw_size.restype = C.c_size_t
    lib.cq_pk_raw_size.argtypes = [vp, C.c_uint32]
    lib.cq_pk_write_raw.argtypes = [vp, vp, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    u32p = C.POINTER(C.c_uint32)
    lib.cq_permutation_assembly_init.restype = None
    lib.cq_permutation_assembly_init.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, u32p]
    lib.cq_permutation_assembly_copy.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, u32p, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_uint32]
    lib.cq_sha_witness_fill_dev.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_size_t, vp]
    lib.cq_sha_spread_table_dev.argtypes = [vp, C.c_size_t, vp, vp]
    lib.cq_xoshiro256ss_seed.restype = None
    lib.cq_xoshiro256ss_seed.argtypes = [C.c_uint64, vp]
    lib.cq_xoshiro256ss_next_u64.restype = C.c_uint64
    lib.cq_xoshiro256ss_next_u64.argtypes = [vp]
    lib.cq_xoshiro256ss_fill.restype = None
    lib.cq_xoshiro256ss_fill.argtypes = [vp, vp, C.c_size_t, C.c_uint32]
    lib.cq_buffer_rng_next_u64.restype = C.c_uint64
    lib.cq_buffer_rng_next_u64.argtypes = [vp]
    lib.cq_opaque_rng_next_u64.restype = C.c_uint64
    lib.cq_opaque_rng_next_u64.argtypes = [vp]
    lib.cq_opaque_rng_fill.restype = None
    lib.cq_opaque_rng_fill.argtypes = [vp, vp, C.c_size_t]
    lib.cq_pk_set_rng_fill.argtypes = [vp, vp]
    lib.cq_msm_precompute_dev.argtypes = [vp, vp, C.c_size_t]
    lib.cq_msm_set_precompute.argtypes = [vp, C.c_int]
    lib.cq_msm_forget_dev.argtypes = [vp, vp]
    lib.cq_sha_synthesis_table_dev.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, vp]
    lib.cq_sha_decomposition_table_dev.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    lib.cq_static_table_new.argtypes = [vp, C.c_size_t, vp, vp, C.POINTER(vp)]
    lib.cq_params_read_raw.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.cq_params_write_raw.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cq_g1_sum.argtypes = [vp, C.c_size_t, vp]
    lib.cq_g1_to_affine.argtypes = [vp, vp]
    lib.cq_profile_enable.argtypes = [vp, C.c_int]
    lib.cq_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
