// Streaming polynomial kernels over BN254 Fr for gfx950: the O(n) passes `create_proof` makes
// between transforms and commitments.  Each replaces a rayon `parallelize` loop or a serial
// recurrence of the reference (cited per function).  All are HBM-bandwidth bound: one 16-byte-per-lane
// coalesced read and write per element, arithmetic fused so every vector is touched once.
#include "poly.hpp"
#include "ctx.hpp"

namespace cq {

static __device__ __forceinline__ Fr ld(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
static __device__ __forceinline__ void st(Fr* p, const Fr& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v.l[0], r.v.l[1], r.v.l[2], r.v.l[3]);
  q[1] = make_uint4(r.v.l[4], r.v.l[5], r.v.l[6], r.v.l[7]);
}

// ---- eval_polynomial (arithmetic.rs:304-329) ------------------------------------------------------
// One Horner run per chunk of L coefficients: S[c] = sum_{i<L} a[cL+i] z^i.  p(z) = S(z^L), so the
// host applies the kernel again on S with z^L until one value is left (3 levels at n = 2^18).
__global__ __launch_bounds__(256) void chunk_eval_kernel(const Fr* __restrict__ a, uint32_t n, Fr z, uint32_t L,
                                                         Fr* __restrict__ S) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nch = (n + L - 1) / L;
  if (c >= nch) return;
  const uint32_t lo = c * L;
  const uint32_t hi = min(n, lo + L);
  Fr acc = Fr::zero();
  for (uint32_t i = hi; i-- > lo;) acc = acc * z + ld(a + i);
  st(S + c, acc);
}

// ---- kate_division (arithmetic.rs:351-387) ---------------------------------------------------------
// q_{i-1} = a_i + z q_i from the top.  Chunk c needs q at the top of the chunk as carry-in; those
// carries are themselves kate_division(S, z^L) of the chunk sums, hence the same recursion as above.
__global__ __launch_bounds__(256) void kate_fill_kernel(const Fr* __restrict__ a, uint32_t n, Fr z, uint32_t L,
                                                        const Fr* __restrict__ T /*nch-1 carries*/, Fr* __restrict__ q) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nch = (n + L - 1) / L;
  if (c >= nch) return;
  const uint32_t lo = c * L;
  const uint32_t hi = min(n, lo + L);
  Fr carry = (c + 1 < nch) ? ld(T + c) : Fr::zero();
  for (uint32_t i = hi; i-- > lo;) {
    carry = ld(a + i) + z * carry;
    if (i >= 1) st(q + (i - 1), carry);
  }
}

// ---- ff::BatchInvert (Montgomery's trick), zeros stay zero -----------------------------------------
// lane t owns elements t, t+T, t+2T, ... (coalesced); one Fermat inversion per BI_CHUNK elements.
constexpr int BI_CHUNK = 8;
__global__ __launch_bounds__(256) void batch_invert_kernel(Fr* __restrict__ a, uint32_t n) {
  const uint32_t T = gridDim.x * blockDim.x;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr v[BI_CHUNK], pref[BI_CHUNK];
  Fr acc = Fr::one();
#pragma unroll
  for (int k = 0; k < BI_CHUNK; k++) {
    const uint32_t i = t + k * T;
    v[k] = i < n ? ld(a + i) : Fr::zero();
    pref[k] = acc;
    if (!v[k].is_zero()) acc = acc * v[k];
  }
  acc = acc.inv();
#pragma unroll
  for (int k = BI_CHUNK - 1; k >= 0; k--) {
    const uint32_t i = t + k * T;
    if (i < n && !v[k].is_zero()) {
      st(a + i, pref[k] * acc);
      acc = acc * v[k];
    }
  }
}

// ---- out[i] = sum_j coeff[j] * p_j[i]  (Polynomial * scalar / + of poly.rs:261-322; theta- and v-folds) ----
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs args, uint32_t n, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr acc = Fr::zero();
  for (uint32_t j = 0; j < args.count; j++) {
    const Fr x = (i < args.len[j]) ? ld(args.p[j] + i) : Fr::zero();
    acc = acc + x * args.coeff[j];
  }
  if (i == 0) acc = acc - args.sub_const;  // `&poly - eval` touches the constant term only (poly.rs:327-335)
  st(out + i, acc);
}

// ---- Fr::random stream: out[i] = from_u512(words[8i..8i+8])  (bn256/fr.rs:159-170) -------------------
__global__ __launch_bounds__(256) void from_u512_kernel(const uint64_t* __restrict__ words, uint32_t n, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t w[8];
  for (int k = 0; k < 8; k++) w[k] = words[(size_t)i * 8 + k];
  st(out + i, Fr::from_u512(w));
}

// ---- CQ round 2: B_r = f_r + beta for r < u, beta for r >= u (static_lookup/prover.rs:261-269; inverted after) ----
__global__ __launch_bounds__(256) void cq_b_denominators_kernel(const Fr* __restrict__ f, uint32_t n, uint32_t u, Fr beta,
                                                                Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(out + i, i < u ? ld(f + i) + beta : beta);
}

// ---- quotient numerator, CQ terms only, then / (X^n - 1)  (evaluation.rs:533-548, domain.rs:319-338) ----
// h = Horner_y over lookups of (b * (f * l_active + beta) - 1), times t_evaluations[i mod t_len]
__global__ __launch_bounds__(256) void cq_quotient_kernel(CqQuotientArgs args, uint32_t ext, Fr* __restrict__ h) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ext) return;
  const Fr la = ld(args.l_active + i);
  const Fr one = Fr::one();
  Fr acc = Fr::zero();
  for (uint32_t l = 0; l < args.count; l++) {
    const Fr b = ld(args.b[l] + i), f = ld(args.f[l] + i);
    acc = acc * args.y + (b * (f * la + args.beta) - one);
  }
  acc = acc * ld(args.t_evals + (i & (args.t_len - 1)));
  st(h + i, acc);
}

// ---- small helpers ------------------------------------------------------------------------------
__global__ void fill_usable_rows_kernel(Fr* out, uint32_t n, uint32_t u) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(out + i, i < u ? Fr::one() : Fr::zero());
}

// =================================================================================================
// host drivers
// =================================================================================================
static inline uint32_t blocks_for(uint32_t n) { return (n + 255) / 256; }

int poly_eval(cq_ctx* c, const Fr* a, uint32_t n, const Fr& z, Fr* out_host) {
  if (n == 0) {
    *out_host = Fr::zero();
    return CQ_OK;
  }
  void* scr;
  int rc;
  const uint32_t L = POLY_CHUNK;
  const uint32_t n1 = (n + L - 1) / L;
  if ((rc = c->ensure_scratch(5, ((size_t)n1 + L + 64) * sizeof(Fr) * 2, &scr)) != CQ_OK) return rc;
  Fr* buf[2] = {(Fr*)scr, (Fr*)scr + n1 + 32};
  const Fr* src = a;
  uint32_t len = n;
  Fr zz = z;
  int which = 0;
  while (true) {
    const uint32_t nch = (len + L - 1) / L;
    chunk_eval_kernel<<<blocks_for(nch), 256, 0, c->stream>>>(src, len, zz, L, buf[which]);
    src = buf[which];
    len = nch;
    if (len == 1) break;
    which ^= 1;
    zz = zz.pow_u64(L);
  }
  if (hipMemcpyAsync(out_host, src, sizeof(Fr), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "eval: D2H failed");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return c->fail(CQ_ERR_HIP, "eval: sync failed");
  return CQ_OK;
}

// q = (a(X) - a(z)) / (X - z), n-1 coefficients.  Recursion depth log_L(n).
static int kate_rec(cq_ctx* c, const Fr* a, uint32_t n, const Fr& z, Fr* q, Fr* scratch, size_t scratch_elems) {
  const uint32_t L = POLY_CHUNK;
  if (n <= 1) return CQ_OK;
  const uint32_t nch = (n + L - 1) / L;
  if (nch == 1) {
    kate_fill_kernel<<<1, 256, 0, c->stream>>>(a, n, z, L, nullptr, q);
    return CQ_OK;
  }
  if (scratch_elems < (size_t)2 * nch) return c->fail(CQ_ERR_INTERNAL, "kate: scratch too small");
  Fr* S = scratch;
  Fr* T = scratch + nch;
  chunk_eval_kernel<<<blocks_for(nch), 256, 0, c->stream>>>(a, n, z, L, S);
  int rc = kate_rec(c, S, nch, z.pow_u64(L), T, scratch + 2 * (size_t)nch, scratch_elems - 2 * (size_t)nch);
  if (rc != CQ_OK) return rc;
  kate_fill_kernel<<<blocks_for(nch), 256, 0, c->stream>>>(a, n, z, L, T, q);
  return CQ_OK;
}

int poly_kate_division(cq_ctx* c, const Fr* a, uint32_t n, const Fr& z, Fr* q) {
  if (n == 0) return c->fail(CQ_ERR_ARG, "kate_division of an empty polynomial");
  void* scr;
  int rc;
  const size_t elems = 4 * ((size_t)n / POLY_CHUNK + POLY_CHUNK) + 256;
  if ((rc = c->ensure_scratch(5, elems * sizeof(Fr), &scr)) != CQ_OK) return rc;
  rc = kate_rec(c, a, n, z, q, (Fr*)scr, elems);
  if (rc != CQ_OK) return rc;
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "kate launch failed");
}

int poly_batch_invert(cq_ctx* c, Fr* a, uint32_t n) {
  if (!n) return CQ_OK;
  const uint32_t threads = (n + BI_CHUNK - 1) / BI_CHUNK;
  batch_invert_kernel<<<blocks_for(threads), 256, 0, c->stream>>>(a, n);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "batch_invert launch failed");
}

int poly_lincomb(cq_ctx* c, const LincombArgs& args, uint32_t n, Fr* out) {
  if (!n) return CQ_OK;
  lincomb_kernel<<<blocks_for(n), 256, 0, c->stream>>>(args, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "lincomb launch failed");
}

int poly_from_u512(cq_ctx* c, const uint64_t* words_dev, uint32_t n, Fr* out) {
  if (!n) return CQ_OK;
  from_u512_kernel<<<blocks_for(n), 256, 0, c->stream>>>(words_dev, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "from_u512 launch failed");
}

int poly_cq_b_denominators(cq_ctx* c, const Fr* f, uint32_t n, uint32_t u, const Fr& beta, Fr* out) {
  cq_b_denominators_kernel<<<blocks_for(n), 256, 0, c->stream>>>(f, n, u, beta, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_b launch failed");
}

int poly_cq_quotient(cq_ctx* c, const CqQuotientArgs& args, uint32_t ext, Fr* h) {
  cq_quotient_kernel<<<blocks_for(ext), 256, 0, c->stream>>>(args, ext, h);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_quotient launch failed");
}

int poly_fill_usable_rows(cq_ctx* c, Fr* out, uint32_t n, uint32_t u) {
  fill_usable_rows_kernel<<<blocks_for(n), 256, 0, c->stream>>>(out, n, u);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fill launch failed");
}

}  // namespace cq
