// Streaming polynomial kernels over BN254 Fr for gfx950: the O(n) passes `create_proof` makes
// between transforms and commitments.  Each replaces a rayon `parallelize` loop or a serial
// recurrence of the reference (cited per function).  All are HBM-bandwidth bound: one 16-byte-per-lane
// coalesced read and write per element, arithmetic fused so every vector is touched once.
#include <cstdlib>
#include "poly.hpp"
#include "field29.hpp"
#include <algorithm>
#include <cstring>
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
static __device__ __forceinline__ void ld8w(const Fr* p, uint32_t* w) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
static __device__ __forceinline__ void st(Fr* p, const Fr& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v.l[0], r.v.l[1], r.v.l[2], r.v.l[3]);
  q[1] = make_uint4(r.v.l[4], r.v.l[5], r.v.l[6], r.v.l[7]);
}

// ---- eval_polynomial (arithmetic.rs:304-329) ------------------------------------------------------
// A block folds EVAL_TILE coefficients of one polynomial.  Lane t takes the coefficients t, t + 256, ...
// of the tile -- so a wave's load is 64 consecutive elements (2 KB) instead of 64 runs 256 bytes apart, which the
// address path served at a fraction of the rate (k = 18: 122 -> ~50 us for the 19 polynomials of a proof) -- runs Horner
// over them with x^256, multiplies by x^t (composed from x^(2^i) by the bits of t) and the 256 lane values are then just
// ADDED in an 8-level LDS tree.  p(x) = S(x^2048) over the block sums S, so the host applies the kernel again until one
// value per polynomial is left (2 levels at n = 2^18).  blockIdx.y selects the polynomial: all evaluations of a proof at
// one point go in one launch.
__global__ __launch_bounds__(256) void block_eval_kernel(EvalBatchArgs args, uint32_t level_stride_in,
                                                         const Fr* __restrict__ level_in, Fr* __restrict__ level_out,
                                                         uint32_t out_stride, EvalPowers pw) {
  __shared__ uint4 sh_lo[256], sh_hi[256];
  const uint32_t poly = blockIdx.y;
  const Fr* a = level_in ? level_in + (size_t)poly * level_stride_in : args.p[poly];
  const uint32_t n = level_in ? args.cur_len[poly] : args.len[poly];
  const uint32_t t = threadIdx.x;
  const uint32_t base = blockIdx.x * EVAL_TILE + t;
  Fr v = Fr::zero();
  if (base < n) {
    // on the lazy 29-bit limbs (field29.hpp): the powers are R' constants, a coefficient is read as it lies in memory, and
    // the running value stays the memory form of the partial sum -- one 222-instruction product and a limb-wise addition per
    // coefficient instead of a ~430-instruction one (the kernel was bound by them: 19 polynomials of 2^18 in 117 us)
    const Fr29 x256 = Fr29::unpack(pw.x256.v.l);
    Fr29 acc = Fr29::zero();
#pragma unroll
    for (int j = (int)(EVAL_TILE / 256) - 1; j >= 0; j--) {
      const uint32_t i = base + 256u * (uint32_t)j;
      Fr29 cf = Fr29::zero();
      if (i < n) {
        uint32_t w[8];
        ld8w(a + i, w);
        cf = Fr29::unpack(w);
      }
      acc = Fr29::mul(acc, x256) + cf;  // < 3 p, limbs < 2^30
    }
    acc.normalise();
#pragma unroll
    for (int bit = 0; bit < 8; bit++)
      if ((t >> bit) & 1u) acc = Fr29::mul(acc, Fr29::unpack(pw.sq[bit].v.l));  // x^t from the bits of t
    uint32_t w[8];
    acc.to_canonical_words(w);
    CQ_UNROLL for (int k = 0; k < 8; k++) v.v.l[k] = w[k];
  }
#pragma unroll 1
  for (uint32_t s = 0; s < 8; s++) {
    sh_lo[t] = make_uint4(v.v.l[0], v.v.l[1], v.v.l[2], v.v.l[3]);
    sh_hi[t] = make_uint4(v.v.l[4], v.v.l[5], v.v.l[6], v.v.l[7]);
    __syncthreads();
    const uint32_t d = 1u << s;
    if ((t & ((d << 1) - 1)) == 0) {
      Fr o;
      uint4 lo = sh_lo[t + d], hi4 = sh_hi[t + d];
      o.v.l[0] = lo.x; o.v.l[1] = lo.y; o.v.l[2] = lo.z; o.v.l[3] = lo.w;
      o.v.l[4] = hi4.x; o.v.l[5] = hi4.y; o.v.l[6] = hi4.z; o.v.l[7] = hi4.w;
      v = v + o;
    }
    __syncthreads();
  }
  if (t == 0) st(level_out + (size_t)poly * out_stride + blockIdx.x, v);
}

// One Horner run per chunk of L coefficients (used by the division below).
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

// ---- kate_division in three launches (round 3) ------------------------------------------------------------------------
// The recursion above is nine launches at n = 2^18 (16-coefficient chunks, five levels), each ~16 us of one lane's sixteen
// dependent products.  Here a 256-thread block owns 256 E consecutive coefficients (E = 4 .. 16 per thread):
//   A  every thread folds its E coefficients (Horner), a suffix scan over the block with the multiplier z^E (eight steps in
//      LDS) turns that into tail[t] = sum_{i >= lo_t, i in the block} a_i z^(i - lo_t); tail[0] is the block's total;
//   C  one block: the carry into every block -- the suffix scan of the totals with the multiplier z^(256 E) -- and the powers
//      z^(E k), k < 256;
//   B  thread t starts from  tail[t + 1] + z^(E (255 - t)) carry(block)  = q at the top of its coefficients and writes its E
//      quotient coefficients.
// ~20 + 20 + 6 dependent products instead of ~9 x 16.  Same values as the recursion (field arithmetic is exact).
static __device__ __forceinline__ void sh_put(uint4* lo, uint4* hi, uint32_t t, const Fr& v) {
  lo[t] = make_uint4(v.v.l[0], v.v.l[1], v.v.l[2], v.v.l[3]);
  hi[t] = make_uint4(v.v.l[4], v.v.l[5], v.v.l[6], v.v.l[7]);
}
static __device__ __forceinline__ Fr sh_get(const uint4* lo, const uint4* hi, uint32_t t) {
  const uint4 a = lo[t], b = hi[t];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
// m[s] = (the scan's multiplier)^(2^s), from the host: a step of the scans below is then one product, not two
struct KateScanPowers {
  Fr m[10];
};
// suffix scan of a linear recurrence over the first `active` threads of a block (a power of two): v_t <- sum_{u >= 0} v_(t+u) m^u
static __device__ __forceinline__ Fr block_suffix_scan(Fr v, const KateScanPowers& pw, uint32_t active, uint4* lo, uint4* hi) {
  const uint32_t t = threadIdx.x;
  uint32_t s = 0;
#pragma unroll 1
  for (uint32_t d = 1; d < active; d <<= 1, s++) {
    sh_put(lo, hi, t, v);
    __syncthreads();
    if (t + d < active) v = v + sh_get(lo, hi, t + d) * pw.m[s];
    __syncthreads();
  }
  return v;
}
__global__ __launch_bounds__(256) void kate_block_tails_kernel(const Fr* __restrict__ a, uint32_t n, Fr z, uint32_t E, KateScanPowers pw /* of z^E */,
                                                               Fr* __restrict__ tail /*[blocks][256]*/, Fr* __restrict__ total /*[blocks]*/) {
  __shared__ uint4 lo[256], hi[256];
  const uint32_t t = threadIdx.x;
  const uint32_t base = (blockIdx.x * 256 + t) * E;
  Fr v = Fr::zero();
  for (uint32_t k = E; k-- > 0;) v = v * z + (base + k < n ? ld(a + base + k) : Fr::zero());
  v = block_suffix_scan(v, pw, 256, lo, hi);
  st(tail + (size_t)blockIdx.x * 256 + t, v);
  if (t == 0) st(total + blockIdx.x, v);
}
// block 0 (`active` threads, a power of two >= nblk): carry[b] = sum_{j > b} total[j] zL^(j - b - 1) (zL = z^(256 E), its
// powers in pw); block 1: pk[k] = zE^k, k < 256
__global__ __launch_bounds__(1024) void kate_block_carries_kernel(const Fr* __restrict__ total, uint32_t nblk, uint32_t active, Fr zE,
                                                                  KateScanPowers pw, Fr* __restrict__ carry, Fr* __restrict__ pk) {
  __shared__ uint4 lo[1024], hi[1024];
  const uint32_t t = threadIdx.x;
  if (blockIdx.x == 1) {
    if (t < 256) st(pk + t, zE.pow_u64(t));
    return;
  }
  Fr v = t < nblk ? ld(total + t) : Fr::zero();
  v = block_suffix_scan(v, pw, active, lo, hi);  // inclusive: total[t] + zL total[t + 1] + ..
  sh_put(lo, hi, t, v);
  __syncthreads();
  if (t < nblk) st(carry + t, t + 1 < nblk ? sh_get(lo, hi, t + 1) : Fr::zero());
}
__global__ __launch_bounds__(256) void kate_block_fill_kernel(const Fr* __restrict__ a, uint32_t n, Fr z, uint32_t E,
                                                              const Fr* __restrict__ tail, const Fr* __restrict__ carry,
                                                              const Fr* __restrict__ pw, Fr* __restrict__ q) {
  const uint32_t t = threadIdx.x;
  const uint32_t base = (blockIdx.x * 256 + t) * E;
  if (base >= n) return;
  Fr c = ld(pw + (255 - t)) * ld(carry + blockIdx.x);
  if (t < 255) c = c + ld(tail + (size_t)blockIdx.x * 256 + t + 1);
  for (uint32_t k = E; k-- > 0;) {
    const uint32_t i = base + k;
    if (i >= n) continue;  // (coefficients beyond the end are zero and so is everything above them)
    c = ld(a + i) + z * c;
    if (i >= 1) st(q + (i - 1), c);
  }
}

// ---- ff::BatchInvert (Montgomery's trick), zeros stay zero -----------------------------------------
// An inversion is a long dependent chain on one lane (binary extended Euclid, ~750 limb steps; Fermat's ~380
// dependent products took 0.4 ms), so it is shared by a whole workgroup: 256 lanes x BI_PER_LANE
// strided (coalesced) elements.  Each lane keeps its elements in registers, parks the running prefix products in
// the output array, the lane totals are scanned in LDS (prefix and suffix), lane 0 inverts the block total, and
// every lane unwinds its own elements: ~5 products per element + one inversion per 4096 elements.
// BI_PER_LANE: 16 for the largest arrays (one inversion per 4096 elements), 8 and 4 while the array leaves CUs idle anyway (the
// chain of a lane -- its elements, two 8-level scans, the inversion -- is then what the launch waits for): poly_batch_invert.
static __device__ __forceinline__ void bi_put(uint4* lo, uint4* hi, uint32_t t, const Fr& v) {
  lo[t] = make_uint4(v.v.l[0], v.v.l[1], v.v.l[2], v.v.l[3]);
  hi[t] = make_uint4(v.v.l[4], v.v.l[5], v.v.l[6], v.v.l[7]);
}
static __device__ __forceinline__ Fr bi_get(const uint4* lo, const uint4* hi, uint32_t t) {
  const uint4 a = lo[t], b = hi[t];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
// On the lazy 29-bit limbs (field29.hpp), straight on the memory words: an element a R read as limbs is the R' value
// of a / 32, and the factors of 32 cancel by themselves -- a lazy product of k words is their product over R'^(k-1), the
// block total T has all of them, and with Y = R^2 / T (inv_safegcd of T's canonical words, read as an R = 2^256 value)
// the unwinding  out = (product of the words before) (x) Y (x) (product of the words after)  comes out as R^2 / (a R) =
// R / a, the memory word of the inverse.  222-instruction products instead of ~430 (the kernel's chain: 16 + 16 + 48 of
// them per lane around the one inversion).
static __device__ __forceinline__ void bi_put29(uint4* lo, uint4* hi, uint32_t t, const Fr29& v) {  // v < 2 p
  uint32_t w[8];
  v.pack(w);
  lo[t] = make_uint4(w[0], w[1], w[2], w[3]);
  hi[t] = make_uint4(w[4], w[5], w[6], w[7]);
}
static __device__ __forceinline__ Fr29 bi_get29(const uint4* lo, const uint4* hi, uint32_t t) {
  const uint4 a = lo[t], b = hi[t];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return Fr29::unpack(w);
}
// The chain a lane waits on (one wave per SIMD: a product is ~1 us of dependent multiply-adds) is kept short by running
// independent products in pairs (Fp29::mul_pair): the lane's elements as two half-chains, the prefix and the suffix scan of
// the lane totals level by level together, and the unwinding two elements at a time (their "before" products, then
// out / r, out / r): 8 + 1 + 8 + inversion + 1 + 24 product latencies per lane at 16 elements, down from 16 + 16 + 1 + 48.
// A zero element takes part as a product by one (and is not written).
template <int BI_PER_LANE>
__global__ __launch_bounds__(256) void batch_invert_kernel(Fr* __restrict__ a, uint32_t n) {
  CQ_CRITICAL_WAVES();
  static_assert(BI_PER_LANE >= 2 && BI_PER_LANE % 2 == 0, "two half-chains per lane");
  constexpr int H = BI_PER_LANE / 2;
  __shared__ uint4 lo[256], hi[256], lo2[256], hi2[256];
  const uint32_t t = threadIdx.x;
  const uint32_t base = blockIdx.x * (256 * BI_PER_LANE);
  uint32_t v[BI_PER_LANE][8];  // the lane's elements as they lie in memory (canonical words)
  bool nz[BI_PER_LANE];
  // (compile-time loops: `#pragma unroll` gives up on a body with the paired products' asm statements, and v[][] indexed
  // by a run-time k is 512 bytes of scratch per lane -- tools/code_object_audit.py)
  Fr29::static_for<0, BI_PER_LANE>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const uint32_t i = base + k * 256 + t;
    uint32_t o = 0;
    CQ_UNROLL for (int q = 0; q < 8; q++) v[k][q] = 0;
    if (i < n) {
      ld8w(a + i, v[k]);
      CQ_UNROLL for (int q = 0; q < 8; q++) o |= v[k][q];
    }
    nz[k] = o != 0;
  });
  auto elem = [&](int k) { return nz[k] ? Fr29::unpack(v[k]) : Fr29::one(); };
  auto park = [&](int k, const Fr29& x) {  // product of the half-chain's earlier elements (< 2 p), read back while unwinding
    if (!nz[k]) return;
    uint32_t w[8];
    x.pack(w);
    uint4* dst = reinterpret_cast<uint4*>(a + (base + k * 256 + t));
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  };
  Fr29 acc_a = Fr29::one(), acc_b = Fr29::one();  // elements 0 .. H-1 and H .. 2H-1
  Fr29::static_for<0, H>([&](auto K) {
    constexpr int k = decltype(K)::value;
    park(k, acc_a);
    park(k + H, acc_b);
    Fr29::mul_pair(acc_a, elem(k), acc_b, elem(k + H), acc_a, acc_b);
  });
  const Fr29 total_a = acc_a;
  const Fr29 acc = Fr29::mul(acc_a, acc_b);
  // inclusive prefix and suffix scans of the lane totals (Hillis-Steele), a level of both per step
  Fr29 inc = acc, suf = acc;
#pragma unroll 1
  for (uint32_t d = 1; d < 256; d <<= 1) {
    bi_put29(lo, hi, t, inc);
    bi_put29(lo2, hi2, t, suf);
    __syncthreads();
    const Fr29 below = t >= d ? bi_get29(lo, hi, t - d) : Fr29::one();
    const Fr29 above = t + d < 256 ? bi_get29(lo2, hi2, t + d) : Fr29::one();
    __syncthreads();
    Fr29::mul_pair(below, inc, suf, above, inc, suf);
  }
  bi_put29(lo, hi, t, inc);
  bi_put29(lo2, hi2, t, suf);
  __syncthreads();
  const Fr29 exc = t ? bi_get29(lo, hi, t - 1) : Fr29::one();          // everything in the lanes before this one
  const Fr29 total = bi_get29(lo, hi, 255);
  const Fr29 after = t < 255 ? bi_get29(lo2, hi2, t + 1) : Fr29::one();  // everything in the lanes after it
  __syncthreads();
  if (t == 0) {
    Fr tc;
    total.pack(tc.v.l);
    Fr::cond_sub_p(tc.v.l, 0);
#if defined(CQ_BI_EXP) && CQ_BI_EXP == 1  // timing experiment (wrong results): no inversion
    bi_put(lo, hi, 0, tc);
#else
#if defined(CQ_BI_CT)  // A/B: the constant-time division steps
    bi_put(lo, hi, 0, tc.inv_safegcd());
#else
    bi_put(lo, hi, 0, tc.inv_safegcd_var());  // Y = R^2 / T as canonical words (one lane: the variable-time form)
#endif
#endif
  }
  const Fr29 exc_b = Fr29::mul(exc, total_a);  // ... and this lane's first half-chain (while lane 0 inverts)
  __syncthreads();
  const Fr29 total_inv = bi_get29(lo, hi, 0);
  // inverse of (everything up to and including this lane) = total^-1 * (product of the later lanes)
  Fr29 r = Fr29::mul(total_inv, after);
  auto prefix = [&](int k) {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (nz[k]) ld8w(a + (base + k * 256 + t), w);
    return Fr29::unpack(w);
  };
  auto emit = [&](int k, const Fr29& out) {
    if (!nz[k]) return;
    uint32_t w[8];
    out.pack(w);
    Fr::cond_sub_p(w, 0);
    uint4* dst = reinterpret_cast<uint4*>(a + (base + k * 256 + t));
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  };
#if defined(CQ_BI_EXP) && CQ_BI_EXP == 2  // timing experiment (wrong results): no unwinding
  emit(0, r);
  return;
#endif
  Fr29::static_for<0, H>([&](auto K) {
    constexpr int k1 = BI_PER_LANE - 1 - 2 * decltype(K)::value, k0 = k1 - 1;  // two elements per step, last first
    Fr29 b1, b0, o1, o0;
    // everything before the element: the earlier lanes, (the first half-chain,) the earlier elements of its own half-chain
    Fr29::mul_pair(k1 >= H ? exc_b : exc, prefix(k1), k0 >= H ? exc_b : exc, prefix(k0), b1, b0);
    Fr29::mul_pair(b1, r, r, elem(k1), o1, r);
    Fr29::mul_pair(b0, r, r, elem(k0), o0, r);
    emit(k1, o1);
    emit(k0, o0);
  });
}

// ---- out[i] = sum_j coeff[j] * p_j[i]  (Polynomial * scalar / + of poly.rs:261-322; theta- and v-folds) ----
// On the lazy 29-bit limbs (field29.hpp): the coefficients arrive as R' constants (poly_lincomb converts them), a
// polynomial's value is read as it lies in memory, and up to six products share one reduction -- 19 polynomials cost
// 19 x 81 multiplies and 4 reductions per element instead of 19 full products of the canonical type (~8 300 -> ~2 500
// instructions; the kernel was bound by them, not by its 32 bytes per term).
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs args, uint32_t n, Fr* __restrict__ out) {
  __shared__ uint32_t cf[LINCOMB_MAX][9];
  for (uint32_t j = threadIdx.x; j < args.count; j += blockDim.x) {
    const Fr29 c = Fr29::unpack(args.coeff[j].v.l);
    CQ_UNROLL for (int l = 0; l < 9; l++) cf[j][l] = c.a[l];
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr29 acc = Fr29::zero();
  uint64_t col[18];
  for (uint32_t j0 = 0; j0 < args.count; j0 += 6) {
    CQ_UNROLL for (int k = 0; k < 18; k++) col[k] = 0;
    const uint32_t j1 = min(args.count, j0 + 6);
    for (uint32_t j = j0; j < j1; j++) {
      if (i >= args.len[j]) continue;
      uint32_t w[8];
      ld8w(args.p[j] + i, w);
      Fr29 c;
      CQ_UNROLL for (int l = 0; l < 9; l++) c.a[l] = cf[j][l];
      Fr29::mac(col, Fr29::unpack(w), c);
    }
    acc = acc + Fr29::redc(col);  // < 2 p each: at most LINCOMB_MAX / 6 of them
    acc.normalise();
  }
  uint32_t w[8];
  acc.to_canonical_words(w);
  Fr r;
  CQ_UNROLL for (int k = 0; k < 8; k++) r.v.l[k] = w[k];
  if (i == 0) r = r - args.sub_const;  // `&poly - eval` touches the constant term only (poly.rs:327-335)
  st(out + i, r);
}

// ---- Fr::random stream: out[i] = from_u512(words[8i..8i+8])  (bn256/fr.rs:159-170) -------------------
__global__ __launch_bounds__(256) void from_u512_kernel(const uint64_t* __restrict__ words, uint32_t n, Fr* __restrict__ out) {
  CQ_CRITICAL_WAVES();
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
// On the lazy 29-bit limbs (field29.hpp mac / redc).  A memory word is v R (R = 2^256); read as limbs it is the R' = 2^261
// Montgomery form of v / 32, so a product of two memory words is the memory word of their product over 32, and a product
// with a constant given in R' form is exact.  The running value is kept as the memory word of acc / 2^10:
//     u = f la (/2^5),  w = u + beta/2^5,  acc' = acc y + b w - 2^-10   (two products, ONE reduction),
// the start value h_in 2^-10 and the final factor scale 2^10 (2^15 with the division: t_evals is a memory word too) are
// exact constant products.  ~3 300 instructions per point instead of ~6 100 (the kernel was bound by them).
struct CqQuotientConsts {
  Fr y261;      // y, R' form
  Fr beta_32;   // beta / 32, as a memory word
  Fr one_1024;  // 2^-10, as a memory word
  Fr c_1024;    // 2^-10, R' form
  Fr scale261;  // scale 2^10 (no division) or scale 2^15, R' form
};
__global__ __launch_bounds__(256) void cq_quotient_kernel(CqQuotientArgs args, CqQuotientConsts k, uint32_t ext, Fr* h) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ext) return;
  uint32_t w8[8];
  auto load29 = [&](const Fr* p) {
    ld8w(p, w8);
    return Fr29::unpack(w8);
  };
  const Fr29 la = load29(args.l_active + i);
  const Fr29 y = Fr29::unpack(k.y261.v.l), beta = Fr29::unpack(k.beta_32.v.l), one = Fr29::unpack(k.one_1024.v.l);
  Fr29 acc = Fr29::zero();
  if (args.h_in) acc = Fr29::mul(load29(args.h_in + i), Fr29::unpack(k.c_1024.v.l));
  for (uint32_t l = 0; l < args.count; l++) {
    const Fr29 b = load29(args.b[l] + i), f = load29(args.f[l] + i);
    Fr29 w = Fr29::mul(f, la) + beta;  // < 3 p, limbs < 2^30
    w.normalise();
    uint64_t col[18];
    CQ_UNROLL for (int q = 0; q < 18; q++) col[q] = 0;
    Fr29::mac(col, acc, y);   // acc < 4 p
    Fr29::mac(col, b, w);
    acc = Fr29::sub<2>(Fr29::redc(col), one);  // < 4 p
  }
  acc = Fr29::mul(acc, Fr29::unpack(k.scale261.v.l));
  if (args.t_len) acc = Fr29::mul(acc, load29(args.t_evals + (i & (args.t_len - 1))));  // t_len = 0: evaluate_h alone, no division
  acc.pack(w8);  // < 2 p
  Fr::cond_sub_p(w8, 0);
  Fr r;
  CQ_UNROLL for (int q = 0; q < 8; q++) r.v.l[q] = w8[q];
  st(h + i, r);
}

// ---- small helpers ------------------------------------------------------------------------------
__global__ void fill_usable_rows_kernel(Fr* out, uint32_t n, uint32_t u) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(out + i, i < u ? Fr::one() : Fr::zero());
}

// one 16-byte half element per lane
__global__ __launch_bounds__(256) void advice_fill_kernel(AdviceFillArgs a, const Fr* __restrict__ tails, uint32_t n, uint32_t u) {
  const uint32_t col = blockIdx.y;
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;  // half-element index
  if (h >= 2 * n) return;
  const uint32_t i = h >> 1;
  const uint4* src = i < u ? reinterpret_cast<const uint4*>(a.src[col]) + h
                           : reinterpret_cast<const uint4*>(tails + (size_t)col * (n - u)) + (h - 2 * u);
  reinterpret_cast<uint4*>(a.dst[col])[h] = *src;
}
__global__ void gather_scalars_kernel(GatherArgs a, Fr* out) {
  CQ_CRITICAL_WAVES();
  const uint32_t i = threadIdx.x;
  if (i < a.count) st(out + i, ld(a.src[i]));
}

// =================================================================================================
// host drivers
// =================================================================================================
static inline uint32_t blocks_for(uint32_t n) { return (n + 255) / 256; }

// evaluates `count` polynomials (device pointers p[i], lengths len[i]) at z; results to the host
int poly_eval_batch(cq_ctx* c, const Fr* const* p, const uint32_t* len, uint32_t count, const Fr& z, Fr* out_host) {
  if (count == 0) return CQ_OK;
  if (count > EVAL_MAX_BATCH) return c->fail(CQ_ERR_ARG, "eval: batch too large");
  EvalBatchArgs args;
  uint32_t maxlen = 0;
  for (uint32_t i = 0; i < count; i++) {
    args.p[i] = p[i];
    args.len[i] = len[i];
    args.cur_len[i] = len[i];
    maxlen = std::max(maxlen, len[i]);
  }
  if (maxlen == 0) {
    for (uint32_t i = 0; i < count; i++) out_host[i] = Fr::zero();
    return CQ_OK;
  }
  const uint32_t nb0 = (maxlen + EVAL_TILE - 1) / EVAL_TILE;
  void* scr;
  int rc;
  if ((rc = c->ensure_scratch(5, ((size_t)count * (nb0 + 8) * 2) * sizeof(Fr), &scr)) != CQ_OK) return rc;
  Fr* buf[2] = {(Fr*)scr, (Fr*)scr + (size_t)count * (nb0 + 8)};
  const uint32_t stride = nb0 + 8;
  Fr x = z;
  const Fr* level_in = nullptr;
  int which = 0;
  uint32_t curmax = maxlen;
  while (true) {
    EvalPowers pw;
    Fr y = x;
    for (int s2 = 0; s2 < 8; s2++) {
      pw.sq[s2] = fr_to_r261(y);
      y = y.sqr();
    }
    pw.x256 = fr_to_r261(y);
    for (uint32_t per = EVAL_TILE / 256; per > 1; per >>= 1) y = y.sqr();  // x^EVAL_TILE: the next level's point
    const uint32_t nb = (curmax + EVAL_TILE - 1) / EVAL_TILE;
    // the last level leaves its `count` results next to each other
    block_eval_kernel<<<dim3(nb, count), 256, 0, c->stream>>>(args, stride, level_in, buf[which], nb == 1 ? 1u : stride, pw);
    for (uint32_t i = 0; i < count; i++) args.cur_len[i] = (args.cur_len[i] + EVAL_TILE - 1) / EVAL_TILE;
    level_in = buf[which];
    curmax = nb;
    if (nb == 1) break;
    which ^= 1;
    x = y;  // x^EVAL_TILE
  }
  // one plain copy into pinned memory (a strided copy into pageable memory took the host ~100 us)
  void* small;
  if ((rc = c->ensure_pinned_small(&small)) != CQ_OK) return rc;
  Fr* stage = (Fr*)((char*)small + 4096);
  static_assert(4096 + EVAL_MAX_BATCH * sizeof(Fr) <= cq_ctx::PINNED_SMALL_BYTES, "pinned page layout");
  if (hipMemcpyAsync(stage, level_in, (size_t)count * sizeof(Fr), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "eval: D2H failed");
  if (int wrc = c->wait(c->stream, "eval: sync failed")) return wrc;
  memcpy(out_host, stage, (size_t)count * sizeof(Fr));
  // empty polynomials evaluate to zero (their blocks wrote zero already)
  return CQ_OK;
}

int poly_eval(cq_ctx* c, const Fr* a, uint32_t n, const Fr& z, Fr* out_host) {
  if (n == 0) {
    *out_host = Fr::zero();
    return CQ_OK;
  }
  return poly_eval_batch(c, &a, &n, 1, z, out_host);
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
  static const bool blocks_off = getenv("CQ_KATE_BLOCKS") && atoi(getenv("CQ_KATE_BLOCKS")) == 0;  // A/B knob
  if (!blocks_off && n >= 4096 && n <= (1u << 22)) {
    uint32_t E = 4;
    while ((uint64_t)1024 * 256 * E < n) E <<= 1;  // at most 1024 blocks (one scan block for their carries)
    const uint32_t nblk = (n + 256 * E - 1) / (256 * E);
    const size_t elems = (size_t)nblk * 256 + 2 * (size_t)nblk + 256;
    if ((rc = c->ensure_scratch(5, elems * sizeof(Fr), &scr)) != CQ_OK) return rc;
    Fr* tail = (Fr*)scr;
    Fr* total = tail + (size_t)nblk * 256;
    Fr* carry = total + nblk;
    Fr* pw = carry + nblk;
    const Fr zE = z.pow_u64(E);
    KateScanPowers pE, pL;
    pE.m[0] = zE;
    for (int k = 1; k < 10; k++) pE.m[k] = pE.m[k - 1] * pE.m[k - 1];
    pL.m[0] = pE.m[8];  // z^(256 E)
    for (int k = 1; k < 10; k++) pL.m[k] = pL.m[k - 1] * pL.m[k - 1];
    uint32_t active = 64;
    while (active < nblk) active <<= 1;
    kate_block_tails_kernel<<<nblk, 256, 0, c->stream>>>(a, n, z, E, pE, tail, total);
    kate_block_carries_kernel<<<2, active < 256 ? 256 : active, 0, c->stream>>>(total, nblk, active, zE, pL, carry, pw);
    kate_block_fill_kernel<<<nblk, 256, 0, c->stream>>>(a, n, z, E, tail, carry, pw, q);
    return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "kate launch failed");
  }
  const size_t elems = 4 * ((size_t)n / POLY_CHUNK + POLY_CHUNK) + 256;
  if ((rc = c->ensure_scratch(5, elems * sizeof(Fr), &scr)) != CQ_OK) return rc;
  rc = kate_rec(c, a, n, z, q, (Fr*)scr, elems);
  if (rc != CQ_OK) return rc;
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "kate launch failed");
}

int poly_batch_invert(cq_ctx* c, Fr* a, uint32_t n) {
  if (!n) return CQ_OK;
  static const int forced = getenv("CQ_BI_PER_LANE") ? atoi(getenv("CQ_BI_PER_LANE")) : 0;
  // measured stand-alone on one box (tools/batch_invert_perf.py, profiles/r03_batch_invert_per_lane.txt): 4 elements per lane
  // up to ~5 x 2^16 (the lane's chain is what the launch waits for), 8 up to ~3 M (k=18's 1.3 M: 127 us against 140), 16 beyond
  const int per = forced ? forced : (n <= 5u * (1u << 16) ? 4 : n <= 3u * (1u << 20) ? 8 : 16);
  if (per == 4) batch_invert_kernel<4><<<(n + 256 * 4 - 1) / (256 * 4), 256, 0, c->stream>>>(a, n);
  else if (per == 8) batch_invert_kernel<8><<<(n + 256 * 8 - 1) / (256 * 8), 256, 0, c->stream>>>(a, n);
  else batch_invert_kernel<16><<<(n + 256 * 16 - 1) / (256 * 16), 256, 0, c->stream>>>(a, n);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "batch_invert launch failed");
}

int poly_lincomb(cq_ctx* c, const LincombArgs& args_in, uint32_t n, Fr* out) {
  if (!n) return CQ_OK;
  LincombArgs args = args_in;
  for (uint32_t j = 0; j < args.count; j++) args.coeff[j] = fr_to_r261(args_in.coeff[j]);  // (sub_const stays a plain value)
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
  CqQuotientConsts k;
  const Fr inv32 = Fr::from_u64(32).inv(), inv1024 = inv32 * inv32;
  k.y261 = fr_to_r261(args.y);
  k.beta_32 = args.beta * inv32;
  k.one_1024 = inv1024;
  k.c_1024 = fr_to_r261(inv1024);
  k.scale261 = fr_to_r261((args.has_scale ? args.scale : Fr::one()) * Fr::from_u64(args.t_len ? 32768 : 1024));
  cq_quotient_kernel<<<blocks_for(ext), 256, 0, c->stream>>>(args, k, ext, h);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_quotient launch failed");
}

int poly_fill_usable_rows(cq_ctx* c, Fr* out, uint32_t n, uint32_t u) {
  fill_usable_rows_kernel<<<blocks_for(n), 256, 0, c->stream>>>(out, n, u);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fill launch failed");
}

int poly_advice_fill(cq_ctx* c, const AdviceFillArgs& a, const Fr* tails_dev, uint32_t n, uint32_t u) {
  if (!a.count) return CQ_OK;
  advice_fill_kernel<<<dim3(blocks_for(2 * n), a.count), 256, 0, c->stream>>>(a, tails_dev, n, u);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "advice fill launch failed");
}

int poly_gather_scalars(cq_ctx* c, const GatherArgs& a, Fr* out_dev) {
  if (!a.count) return CQ_OK;
  gather_scalars_kernel<<<1, GATHER_MAX, 0, c->stream>>>(a, out_dev);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "gather launch failed");
}

}  // namespace cq
