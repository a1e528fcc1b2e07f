"""MI355X (gfx950) backend for the CQ-lookup / KZG commitment hot path of
halo2_proofs::plonk::create_proof (reference: aleph-zero-foundation/sha2-on-cq-halo2).

The product is `libcq_halo2.so` (hand-written HIP kernels behind the C ABI declared in
`include/cq_halo2.h`).  This Python layer is a thin host-side mirror of the reference's
operator interface (`best_fft`, `best_multiexp`, `EvaluationDomain`, `ParamsKZG`, ...) used by
the tests and the bench; arrays are numpy uint64[..., 4] Montgomery limbs, exactly the bytes
the Rust side holds.
"""
from ._lib import CqError, load, header_symbols  # noqa: F401
from .api import (Context, DevBuf, EvaluationDomain, ParamsKZG, ProvingKey, StaticTable,  # noqa: F401
                  TableConfig)
