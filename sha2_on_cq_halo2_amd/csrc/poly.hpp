// Internal interface of the streaming polynomial kernels (poly.hip) and EvaluationDomain (domain.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "field.hpp"
#include "ntt.hpp"

struct cq_ctx;

namespace cq {

constexpr uint32_t POLY_CHUNK = 16;   // coefficients per lane in chunked Horner / division (64: 4096 lanes at n = 2^18, each kernel ~60 us of latency)
constexpr uint32_t LINCOMB_MAX = 40;  // polynomials per linear combination launch
constexpr uint32_t CQ_MAX_LOOKUPS = 16;

struct LincombArgs {
  const Fr* p[LINCOMB_MAX];
  uint32_t len[LINCOMB_MAX];
  Fr coeff[LINCOMB_MAX];
  uint32_t count;
  Fr sub_const;  // subtracted from coefficient 0
};
// x -> the canonical words of 32 x: read as 29-bit limbs that is x in the kernels' R' = 2^261 Montgomery form (field29.hpp)
inline Fr fr_to_r261(const Fr& x) { return Fr::from_u64(32) * x; }

#ifndef CQ_EVAL_TILE
#define CQ_EVAL_TILE 4096
#endif
constexpr uint32_t EVAL_TILE = CQ_EVAL_TILE;  // coefficients folded by one block of block_eval_kernel (a power of two, 256 lanes x EVAL_TILE / 256 each)
constexpr uint32_t EVAL_MAX_BATCH = 40;   // polynomials per batched evaluation
struct EvalBatchArgs {
  const Fr* p[EVAL_MAX_BATCH];
  uint32_t len[EVAL_MAX_BATCH];
  uint32_t cur_len[EVAL_MAX_BATCH];
};
struct EvalPowers {  // (as R' = 2^261 constants: fr_to_r261)
  Fr sq[8];  // x, x^2, x^4, ..., x^128: lane t composes x^t from the bits of t
  Fr x256;   // the Horner multiplier of a lane's strided run
};

struct CqQuotientArgs {
  const Fr* b[CQ_MAX_LOOKUPS];
  const Fr* f[CQ_MAX_LOOKUPS];
  uint32_t count;
  const Fr* h_in;  // terms folded so far (gates, permutation), or nullptr
  const Fr* l_active;
  const Fr* t_evals;
  uint32_t t_len;
  Fr y, beta;
  // resident sharding: this rank folds a contiguous range of the lookups only; its Horner value is then multiplied by
  // `scale` = y^(lookups after the range), so that the ranks' results add up to the whole sum (has_scale = 0: no factor)
  Fr scale;
  uint32_t has_scale = 0;
};

int poly_eval(cq_ctx* c, const Fr* a_dev, uint32_t n, const Fr& z, Fr* out_host);
int poly_eval_batch(cq_ctx* c, const Fr* const* p_dev, const uint32_t* len, uint32_t count, const Fr& z, Fr* out_host);
int poly_kate_division(cq_ctx* c, const Fr* a_dev, uint32_t n, const Fr& z, Fr* q_dev);
int poly_batch_invert(cq_ctx* c, Fr* a_dev, uint32_t n);
int poly_lincomb(cq_ctx* c, const LincombArgs& args, uint32_t n, Fr* out_dev);
int poly_from_u512(cq_ctx* c, const uint64_t* words_dev, uint32_t n, Fr* out_dev);
int poly_cq_b_denominators(cq_ctx* c, const Fr* f, uint32_t n, uint32_t u, const Fr& beta, Fr* out);
int poly_cq_quotient(cq_ctx* c, const CqQuotientArgs& args, uint32_t ext, Fr* h);
int poly_fill_usable_rows(cq_ctx* c, Fr* out, uint32_t n, uint32_t u);
// dst[j] rows [0, u) <- src[j] rows [0, u), rows [u, n) <- tails[j * (n - u) ..]: the advice columns of a phase with their
// blinding rows, one launch for up to ADVICE_FILL_MAX columns
constexpr uint32_t ADVICE_FILL_MAX = 32;
struct AdviceFillArgs {
  const Fr* src[ADVICE_FILL_MAX];
  Fr* dst[ADVICE_FILL_MAX];
  uint32_t count;
};
int poly_advice_fill(cq_ctx* c, const AdviceFillArgs& a, const Fr* tails_dev, uint32_t n, uint32_t u);
// out[i] = *src[i] (device scalars scattered over polynomials -> one contiguous run the host reads with one copy)
constexpr uint32_t GATHER_MAX = 64;
struct GatherArgs {
  const Fr* src[GATHER_MAX];
  uint32_t count;
};
int poly_gather_scalars(cq_ctx* c, const GatherArgs& a, Fr* out_dev);

}  // namespace cq

// EvaluationDomain<Fr> (poly/domain.rs:19-34): constants on the host, twiddles cached in the context.
struct cq_domain {
  cq_ctx* ctx;
  uint32_t j, k, extended_k, quotient_poly_degree;
  cq::Fr omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor,
      extended_ifft_divisor, barycentric_weight;
  std::vector<cq::Fr> t_evaluations;
  cq::Fr* t_evaluations_dev = nullptr;
  size_t n() const { return (size_t)1 << k; }
  size_t ext() const { return (size_t)1 << extended_k; }
};

namespace cq {
int domain_create(cq_ctx* c, uint32_t j, uint32_t k, cq_domain** out);
void domain_destroy(cq_domain* d);
// batch of columns, contiguous with the given strides (elements)
int domain_lagrange_to_coeff(cq_domain* d, const Fr* in, Fr* out, uint32_t batch, size_t in_stride, size_t out_stride);
int domain_coeff_to_extended(cq_domain* d, const Fr* in, Fr* out, uint32_t batch, size_t in_stride, size_t out_stride);
int domain_extended_to_coeff(cq_domain* d, const Fr* in, Fr* out /* n*(j-1) */);
int domain_fft(cq_ctx* c, const Fr* in, Fr* out, uint32_t log_n, const Fr& omega, uint32_t batch, size_t in_stride,
               size_t out_stride);
}  // namespace cq
