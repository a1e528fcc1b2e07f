// Internal interface of the G1 FFT and the FK-style table preprocessing (g1fft.hip).
#pragma once
#include "curve.hpp"

struct cq_ctx;

namespace cq {
// in-place FFT of 2^log_n packed XYZZ points (R' form, curve29.hpp): out[i] = sum_j omega^(ij) in[j]
int g1_fft(cq_ctx* c, XYZZ* data, uint32_t log_n, const Fr& omega);
// g_to_lagrange (arithmetic.rs:277-301): affine in / affine out, device arrays of 2^k points
int g1_to_lagrange(cq_ctx* c, const G1Affine* g, uint32_t k, G1Affine* out);
// cached quotients of StaticTableValues::new (static_lookup.rs:108-119) in O(N log N) group operations
int fk_table_quotients(cq_ctx* c, const Fr* coeffs, const G1Affine* srs, uint32_t log_n, G1Affine* qs_out);
}  // namespace cq
