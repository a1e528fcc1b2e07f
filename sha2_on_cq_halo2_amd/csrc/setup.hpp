// Internal: GPU construction of test SRSs (see setup.hip).
#pragma once
#include "curve.hpp"

struct cq_ctx;

namespace cq {
Fr domain_root(uint32_t k);
int fixed_base_mul(cq_ctx* c, const Fr* scalars_dev, uint32_t n, G1Affine* out_dev);
int srs_powers_and_lagrange(cq_ctx* c, uint32_t k, const Fr& s, G1Affine* g_dev, G1Affine* g_lagrange_dev,
                            Fr* scalars_tmp_dev, Fr* lagrange_scalars_keep_dev);
int srs_opening_at_zero(cq_ctx* c, uint32_t k, const Fr& s, const Fr* lagrange_scalars_dev, Fr* scalars_tmp_dev,
                        G1Affine* out_dev);
}  // namespace cq
