// Radix-2 NTT over BN254 Fr for gfx950 -- replaces `best_fft` (halo2_proofs/src/arithmetic.rs:171-274)
// and the `EvaluationDomain` wrappers around it (poly/domain.rs:238-374).
//
// Algorithm (not the reference's bit-reverse + recursive butterflies): a Stockham auto-sort
// decomposition.  log_n is split into passes of `deg` bits.  A pass with `lgp` bits already done
// reads, for every "index" in [0, n>>deg), the 2^deg elements in[index + i*(n>>deg)], each carrying the
// twiddle w^((n>>lgp>>deg) * k * i) with k = index mod 2^lgp (the PREVIOUS pass multiplied it in when it
// stored the element, see "arithmetic of the passes"), runs a 2^deg-point decimation-in-frequency FFT
// in LDS, and writes output digit i' to out[((index-k)<<deg) + k + i'*2^lgp].  After the last pass the data is in natural order, so
// input and output orders match `best_fft` (natural in / natural out) with no separate
// bit-reversal pass over HBM.
//
// HBM access: a workgroup owns a tile of T consecutive `index` values, so every global access is a
// run of T consecutive 32-byte elements (T=16..32 -> 512 B..1 KiB runs), 16 B per lane.
// LDS: the tile lives as nine planes of 32-bit limbs (field29.hpp), consecutive elements in consecutive words ->
// conflict-free ds_read/write_b32.
// The coset shift (zeta^(i mod 3), domain.rs:347-363), zero padding to the extended domain
// (domain.rs:259), the 1/n scaling of the inverse transform (domain.rs:366-374) and the truncation
// of `extended_to_coeff` (domain.rs:311-312) are fused into the first/last pass instead of being
// separate streaming passes.
#include <cstdlib>
#include "ntt.hpp"
#include "ctx.hpp"

namespace cq {

__global__ void fr_powers_kernel(Fr* out, Fr base, uint64_t step, uint32_t count) {
  // out[j] = base^(j*step)
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  out[j] = base.pow_u64((uint64_t)j * step);
}

// ---- arithmetic of the passes ------------------------------------------------------------------------------------
// The butterflies run on the lazy 9 x 29-bit field type (field29.hpp: Montgomery form with R' = 2^261, values in
// [0, K p), no packing / conditional subtraction between operations) -- a product is 226 instructions instead of the
// ~330 of the canonical 8 x u32 type, and the additions need no compare-and-subtract.  Bounds: every table constant
// is canonical (< p, K = 1); an element entering level r of the in-LDS FFT is below B_r p with B_0 = 2, and a level
// doubles the bound (s = u + v, d = u - v + B_r p, the latter below 2 p again when it is multiplied by a root), so
// B_r = 2^(r+1) <= 64 and every product has Ka * Kb <= 128 * 1.  Each element passes through exactly one more
// product per pass, at the STORE: times the next pass's twiddle (or 1) between passes -- so the scratch holds values
// below 2 p, packed -- and times the output factor in the last pass, where the constant is the factor's R = 2^256
// limbs, which turns R' form into the caller's R form in the same product.  The first pass converts on load the
// same way (times z 2^266) only when it has a coset shift to apply; otherwise it does not touch the input: the limbs of
// a R = 2^256 value a R ARE the R' = 2^261 value of a / 32, the whole transform is linear, and the missing factor 32 is
// folded into the last pass's output constant on the host (ntt_run) -- one product per element and transform less.
// Results are the same field elements as before, stored canonically.
static __device__ __forceinline__ Fr29 lds_load29(const uint32_t* sm, uint32_t E, uint32_t i) {
  Fr29 r;
  CQ_UNROLL for (int l = 0; l < 9; l++) r.a[l] = sm[l * E + i];
  return r;
}
static __device__ __forceinline__ void lds_store29(uint32_t* sm, uint32_t E, uint32_t i, const Fr29& r) {
  CQ_UNROLL for (int l = 0; l < 9; l++) sm[l * E + i] = r.a[l];
}
static __device__ __forceinline__ Fr29 g_load29(const Fr* p) {  // 8 x u32 -> limbs (no arithmetic)
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return Fr29::unpack(w);
}
static __device__ __forceinline__ void g_store29(Fr* p, const Fr29& x, bool canonical) {  // x < 2 p, normalised
  uint32_t w[8];
  x.pack(w);
  if (canonical) Fr::cond_sub_p(w, 0);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
static __device__ __forceinline__ Fr29 const29(const uint32_t* c) {
  Fr29 r;
  CQ_UNROLL for (int l = 0; l < 9; l++) r.a[l] = c[l];
  return r;
}
// value < 2 p -> the canonical representative (< p)
static __device__ __forceinline__ Fr29 canon29(const Fr29& x) {
  uint32_t w[8];
  x.pack(w);
  Fr::cond_sub_p(w, 0);
  return Fr29::unpack(w);
}
// u - v + K p for v < K p, K = 2^(rnd+1)
static __device__ __forceinline__ Fr29 sub_level(const Fr29& u, const Fr29& v, uint32_t rnd) {
  switch (rnd) {
    case 0: return Fr29::sub<2>(u, v);
    case 1: return Fr29::sub<4>(u, v);
    case 2: return Fr29::sub<8>(u, v);
    case 3: return Fr29::sub<16>(u, v);
    case 4: return Fr29::sub<32>(u, v);
    default: return Fr29::sub<64>(u, v);
  }
}

// one-off: table of R = 2^256 Montgomery values -> canonical R' = 2^261 values, in place
__global__ void ntt_table_to29_kernel(Fr* t, uint32_t count) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  g_store29(t + j, Fr29::mul(g_load29(t + j), const29(CONSTS29<FrP>.from256)), true);
}

#ifdef CQ_NTT_TRACE  // diagnostic build (tools/ntt_wg_trace.py): every workgroup of a pass logs where and when it ran
__device__ uint64_t g_ntt_trace[4 * 65536];
__device__ uint32_t g_ntt_trace_n;
static __device__ __forceinline__ uint32_t ntt_trace_begin() {
  uint32_t slot = 0;
  if (threadIdx.x == 0) {
    slot = atomicAdd(&g_ntt_trace_n, 1u) & 65535u;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_ntt_trace[4 * slot] = wall_clock64();
    g_ntt_trace[4 * slot + 2] = ((uint64_t)xcc << 32) | hw;
    g_ntt_trace[4 * slot + 3] = ((uint64_t)blockIdx.y << 32) | blockIdx.x;
  }
  return slot;
}
static __device__ __forceinline__ void ntt_trace_end(uint32_t slot) {
  __syncthreads();
  if (threadIdx.x == 0) g_ntt_trace[4 * slot + 1] = wall_clock64();
}
extern "C" int cq_debug_ntt_trace(uint64_t* out, uint32_t cap) {  // returns the number of records, resets the log
  uint32_t n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ntt_trace_n), 4) != hipSuccess) return -1;
  if (n > cap) n = cap;
  if (n > 65536) n = 65536;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ntt_trace), (size_t)n * 32) != hipSuccess) return -1;
  const uint32_t z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ntt_trace_n), &z, 4) != hipSuccess) return -1;
  return (int)n;
}
#endif

static_assert(NTT_MAX_DEG <= 6, "sub_level: B_r = 2^(r+1) <= 64");
static __device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0;
}

// DEG / LOG_T != 0: the pass's shape is a compile-time constant (the shapes the BASELINE sizes use are instantiated:
// the butterfly levels unroll, the level's bound K p and every LDS offset become immediates); 0: read from the arguments.
template <uint32_t DEG, uint32_t LOG_T>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs a) {
  extern __shared__ uint32_t smem29[];
#ifdef CQ_NTT_TRACE
  const uint32_t trace_slot = ntt_trace_begin();
#endif
  if (a.flags & NTT_CRITICAL) CQ_CRITICAL_WAVES();
  constexpr bool DYN = DEG == 0;
  const uint32_t deg = DYN ? a.deg : DEG, log_t = DYN ? a.log_t : LOG_T;
  const uint32_t D = 1u << deg;
  const uint32_t T = 1u << log_t;
  const uint32_t E = D * T;
  uint32_t* const tw29 = smem29 + 9 * E;  // the pass's D / 2 butterfly roots, unpacked: nine planes of D / 2 words
  uint32_t* const cl29 = tw29 + 9 * (D / 2 + 1);  // per-residue (g mod 3) constants of the load, then of the store: 2 x 3 x 9
  uint32_t* const cs29 = cl29 + 27;               // words (in LDS: as private arrays indexed by g mod 3 they lived in scratch)
  const uint32_t n = 1u << a.log_n;
  const uint32_t t = n >> deg;
  const uint32_t p = 1u << a.lgp;
  const uint32_t tile = blockIdx.x;
  const uint32_t batch = blockIdx.y;
  const Fr* in = a.in + (size_t)batch * a.in_stride;
  Fr* out = a.out + (size_t)batch * a.out_stride;
  const uint32_t index0 = tile * T;
  const bool first = a.lgp == 0, last = a.next_deg == 0;
  // LDS image: element (row, column) at word at(row, column) of every limb plane.  With 16 columns a half-wave (what one
  // ds_*_b32 cycle serves from the 64 banks) is two row groups, and in the second and third radix-4 steps and in the store
  // their rows differ by a multiple of four rows = 64 words: the same 16 banks twice, every such access at half rate.
  // XORing the low two bits of the row with bits 2-3 and 4-5 puts the four row groups of a wave on four bank groups in
  // every phase (found by enumeration, tools/lds_bank_conflicts.py); wider tiles (32 / 64 columns) have no such conflicts.
  constexpr bool SWZ = !DYN && LOG_T == 4;
  auto at = [&](uint32_t row, uint32_t c) { return SWZ ? (((row ^ (((row >> 2) ^ (row >> 4)) & 3u)) << 4) + c) : row * T + c; };

  // ---- load tile: element (row i, col c) <- in[index0 + c + i*t] ----
  // first pass: R form -> R' form (times 2^266), with the coset factor zeta^(g mod 3) folded into the constant;
  // later passes: the previous pass already applied this pass's twiddle and left R' values below 2 p
  const bool load_mul = first && (a.flags & NTT_IN_COSET);
  if (load_mul && threadIdx.x < 3) {
    Fr29 cl = const29(CONSTS29<FrP>.from256);
    if (threadIdx.x) cl = Fr29::mul(Fr29::unpack(a.in_coset[threadIdx.x - 1].v.l), const29(CONSTS29<FrP>.c271));  // z 2^266, < 2 p
    CQ_UNROLL for (int l = 0; l < 9; l++) cl29[threadIdx.x * 9 + l] = cl.a[l];
  }
  if (last && threadIdx.x >= 64 && threadIdx.x < 67) {
    // R' -> R with the output factor: the constant is the factor's own R = 2^256 limbs (1 -> R mod p)
    const uint32_t m = threadIdx.x - 64;
    const Fr one = Fr::one();
    const uint32_t* src = (a.flags & NTT_OUT_MUL) ? ((a.flags & NTT_OUT_COSET) ? a.out_mul[m].v.l : a.out_mul[0].v.l) : one.v.l;
    const Fr29 cm = Fr29::unpack(src);
    CQ_UNROLL for (int l = 0; l < 9; l++) cs29[m * 9 + l] = cm.a[l];
  }
  if (load_mul) __syncthreads();
  const uint32_t half = D >> 1;
  for (uint32_t j = threadIdx.x; j < half; j += NTT_THREADS) {  // roots (w_n^(n / D))^j, j < D / 2 (canonical, K = 1)
    const Fr29 w = g_load29(a.pq + ((size_t)j << a.pq_shift));
    CQ_UNROLL for (int l = 0; l < 9; l++) tw29[l * half + j] = w.a[l];
  }
  for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
    const uint32_t c = e & (T - 1);
    const uint32_t i = e >> log_t;
    const uint32_t g = index0 + c + i * t;
    Fr29 x = Fr29::zero();
    if (g < a.in_len) {
      x = g_load29(in + g);  // without a coset shift: canonical limbs read as the R' value of a / 32 (< p)
      if (load_mul) x = Fr29::mul(x, lds_load29(cl29 + (g % 3) * 9, 1, 0));  // 1 * 2
    }
    lds_store29(smem29, E, at(i, c), x);
  }
  __syncthreads();

  // ---- 2^deg-point DIF in LDS, all T columns at once ----
  // Templated shapes: two levels at a time in registers (a radix-4 step is the same four products as two radix-2 levels,
  // but the four elements make ONE round trip through LDS instead of two, the intermediate sums are not normalised, and
  // there is one barrier per pair of levels); an odd DEG ends with a plain level.  Generic shape: level by level.
#if defined(CQ_NTT_EXP_NOSTAGES)  // timing experiment (wrong results; tools/ntt_phase_exp.sh): no butterflies in the templated shapes
  constexpr uint32_t R4 = 0, LEVELS = 0;
#elif defined(CQ_NTT_RADIX2)  // A/B knob: level by level as in the generic shape
  constexpr uint32_t R4 = 0, LEVELS = DEG;
#else
  constexpr uint32_t R4 = DEG / 2, LEVELS = DEG;  // (no radix-4 steps for the generic shape, DEG = 0)
#endif
  // (a compile-time loop: with the paired products' asm statements in the body `#pragma unroll` gives up, and the steps'
  // level bounds, root strides and LDS offsets must stay immediates)
  Fr29::static_for<0, (int)R4>([&](auto ST) {
    constexpr uint32_t st = (uint32_t) decltype(ST)::value;
    constexpr uint32_t rnd = 2 * st;
    const uint32_t bit = half >> rnd;
    // group (blk, dj), dj < bit / 2: rows r0 = 2 blk bit + dj, r1 = r0 + bit / 2, r2 = r0 + bit, r3 = r2 + bit / 2.
    // Level rnd pairs (r0, r2) with root index dj and (r1, r3) with dj + bit / 2; level rnd + 1 pairs (r0, r1) and
    // (r2, r3), both with root index dj.  dj sits in the high bits of the work index, so the waves with dj = 0 skip
    // the products by 1 as before.
    const uint32_t hb = bit >> 1;
    for (uint32_t w = threadIdx.x; w < (half >> 1) * T; w += NTT_THREADS) {
      const uint32_t c = w & (T - 1);
      const uint32_t wg = w >> log_t;
      const uint32_t dj = wg >> rnd;
      const uint32_t r0 = (((wg & ((1u << rnd) - 1u)) * bit) << 1) + dj;
      const uint32_t o0 = at(r0, c), o1 = at(r0 + hb, c), o2 = at(r0 + bit, c), o3 = at(r0 + bit + hb, c);
      const Fr29 x0 = lds_load29(smem29, E, o0), x1 = lds_load29(smem29, E, o1);
      const Fr29 x2 = lds_load29(smem29, E, o2), x3 = lds_load29(smem29, E, o3);
      const Fr29 s02 = x0 + x2, s13 = x1 + x3;  // < 2 B_r p, limbs < 2^30: not normalised (operands of + and sub only)
      Fr29 d02 = sub_level(x0, x2, rnd);        // < 2 B_r p
      Fr29 d13 = sub_level(x1, x3, rnd);
      // (independent products in pairs: Fp29::mul_pair, 17 instructions fewer per product)
      if (dj) Fr29::mul_pair(d02, lds_load29(tw29, half, dj << rnd), d13, lds_load29(tw29, half, (dj + hb) << rnd), d02, d13);
      else d13 = Fr29::mul(d13, lds_load29(tw29, half, (dj + hb) << rnd));  // < 2 p
      Fr29 y0 = s02 + s13;                      // < 4 B_r p = B_(r+2) p, limbs < 2^31
      y0.normalise();
      Fr29 y1 = sub_level(s02, s13, rnd + 1);   // < 4 B_r p
      Fr29 y2 = d02 + d13;                      // < (2 B_r + 2) p
      y2.normalise();
      Fr29 y3 = sub_level(d02, d13, rnd + 1);   // d13 < 2 p <= 2 B_r p: < 4 B_r p
      if (dj) {
        const Fr29 w2 = lds_load29(tw29, half, dj << (rnd + 1));
        Fr29::mul_pair(y1, w2, y3, w2, y1, y3);  // 4 B_r * 1 <= 128
      }
      lds_store29(smem29, E, o0, y0);
      lds_store29(smem29, E, o1, y1);
      lds_store29(smem29, E, o2, y2);
      lds_store29(smem29, E, o3, y3);
    }
    __syncthreads();
  });
#pragma unroll
  for (uint32_t rnd = 2 * R4; rnd < (DYN ? 6u : LEVELS); rnd++) {  // generic shape: every level; odd DEG: the last one
    if (DYN && rnd >= deg) break;
    const uint32_t bit = half >> rnd;
    // Work item w -> (butterfly b, column c) with the butterfly's root index di = b mod bit in the HIGH bits of w: the
    // items of a wave then share di, and the waves with di = 0 (root 1: half the butterflies of the second-to-last level,
    // a quarter of the one before, ...) skip the product instead of paying for it through divergence -- about one of the
    // pass's six levels of products in all.
    const uint32_t lg_rest = rnd;  // log2(half / bit): the bits of b above di
    for (uint32_t w = threadIdx.x; w < half * T; w += NTT_THREADS) {
      const uint32_t c = w & (T - 1);
      const uint32_t wb = w >> log_t;
      const uint32_t di = wb >> lg_rest;
      const uint32_t b = ((wb & ((1u << lg_rest) - 1u)) * bit) | di;
      const uint32_t i0 = (b << 1) - di;
      const uint32_t i1 = i0 + bit;
      const Fr29 u = lds_load29(smem29, E, at(i0, c));
      const Fr29 v = lds_load29(smem29, E, at(i1, c));
      Fr29 s = u + v;  // < 2 B_r p
      s.normalise();
      Fr29 d = sub_level(u, v, rnd);  // < 2 B_r p
      if (di) d = Fr29::mul(d, lds_load29(tw29, half, di << rnd));  // 2 B_r * 1 <= 128
      lds_store29(smem29, E, at(i0, c), s);
      lds_store29(smem29, E, at(i1, c), d);
    }
    __syncthreads();
  }

  // ---- store: output digit i' (bit-reversed LDS row) -> out[((index-k)<<deg) + k + i'*p], through one product ----
  const uint32_t lgp2 = a.lgp + deg;            // the next pass: p' = 2^lgp2, t' = n >> next_deg
  const uint32_t log_t2 = a.log_n - a.next_deg;
  // Two elements per iteration (e and e + NTT_THREADS): their products -- by the next pass's twiddle, by 1, or by the output
  // factor -- are independent and go through Fp29::mul_pair.
  auto prepare = [&](uint32_t e, Fr29& x, Fr29& w, uint32_t& g) -> bool {
    if (e >= E) return false;
    const uint32_t c = e & (T - 1);
    const uint32_t i = e >> log_t;
    const uint32_t index = index0 + c;
    const uint32_t k = index & (p - 1);
    g = ((index - k) << deg) + k + i * p;
    if (g >= a.out_len) return false;
    x = lds_load29(smem29, E, at(bitrev(i, deg), c));  // < 128 p
    if (last) {
      const uint32_t m = (a.flags & NTT_OUT_COSET) ? g % 3 : 0;
      w = lds_load29(cs29 + m * 9, 1, 0);
      return true;
    }
    // twiddle of the next pass for the element it will read at g: row i2 = g / t', index2 = g mod t',
    // exponent (n >> lgp2 >> next_deg) * (index2 mod 2^lgp2) * i2  < n
    const uint32_t i2 = g >> log_t2, index2 = g & ((1u << log_t2) - 1);
    const uint32_t ex = ((n >> lgp2) >> a.next_deg) * (index2 & ((1u << lgp2) - 1)) * i2;
    if (!ex) {
      w = const29(CONSTS29<FrP>.one);
    } else if (a.tw_full) {
      w = g_load29(a.tw_full + ex);
    } else {
      w = g_load29(a.tw_lo + (ex & ((1u << a.tw_l) - 1)));
      const uint32_t h = ex >> a.tw_l;
      if (h) w = canon29(Fr29::mul(w, g_load29(a.tw_hi + h)));
    }
    return true;
  };
  for (uint32_t e = threadIdx.x; e < E; e += 2 * NTT_THREADS) {
    Fr29 xa = Fr29::zero(), xb = Fr29::zero(), wa = Fr29::zero(), wb = Fr29::zero();
    uint32_t ga = 0, gb = 0;
    const bool oka = prepare(e, xa, wa, ga), okb = prepare(e + NTT_THREADS, xb, wb, gb);
#ifdef CQ_NTT_EXP_NOTW  // timing experiment (wrong results): no product at the store
    if (oka) g_store29(out + ga, xa, false);
    if (okb) g_store29(out + gb, xb, false);
    continue;
#endif
    Fr29 ra, rb;
    Fr29::mul_pair(xa, wa, xb, wb, ra, rb);  // 128 * 1
    if (oka) g_store29(out + ga, ra, last);  // the last pass leaves canonical values
    if (okb) g_store29(out + gb, rb, last);
  }
#ifdef CQ_NTT_TRACE
  ntt_trace_end(trace_slot);
#endif
}

// ---- wide passes: 7, 8 or 9 bits per pass ------------------------------------------------------------------------------
// A 2^18 transform in TWO passes of nine bits (2^19..2^21 in three instead of four) saves a whole trip through HBM and one
// inter-pass product per element.  Tile: 2^DEG rows x T columns = 2048 elements (73.7 KB of LDS as nine limb planes: two
// workgroups of 512 threads per CU, the same sixteen waves as four 256-thread ones); T = 4 columns at nine bits, i.e.
// every global access is a run of 128 B -- one cache line -- instead of 512 B.
// The lazy bound B_r = 2^(r+1) of the butterflies reaches 128 p after six levels, the most a product's other operand (a
// canonical root) allows.  So after the third radix-4 step every output that was not just multiplied by a root is reduced
// with one product by the field's one (a quarter of the elements plus the groups with root index 0: ~0.34 products per
// element), everything is below 4 p again, and the remaining levels run with the bounds of levels 1, 2, 3.
constexpr uint32_t NTT_WIDE_THREADS = 512, NTT_WIDE_TILE_LOG = 11, NTT_WIDE_MAX_DEG = 9;

template <uint32_t DEG, uint32_t LOG_T>
__global__ __launch_bounds__(NTT_WIDE_THREADS) void ntt_pass_wide_kernel(NttPassArgs a) {
  static_assert(DEG >= 7 && DEG <= NTT_WIDE_MAX_DEG && DEG + LOG_T == NTT_WIDE_TILE_LOG, "wide pass shapes");
  extern __shared__ uint32_t smem29[];
  constexpr uint32_t D = 1u << DEG, T = 1u << LOG_T, E = D * T, half = D >> 1, TH = NTT_WIDE_THREADS;
  uint32_t* const cl29 = smem29 + 9 * E;  // per-residue constants of the load, then of the store (no room for the roots:
  uint32_t* const cs29 = cl29 + 27;       // two 73.7-KB tiles per CU leave 12 KB -- they come from the L1 / L2-resident table)
  // LDS image: element (row, column) at word  swz(row) * T + column  of every limb plane.  A ds_read_b32 serves 32 lanes per
  // cycle from 64 banks; with T = 4 columns a half-wave is eight row groups, and the butterflies' rows differ by 128, 32
  // or 8 rows between neighbouring groups -- multiples of 64 words, i.e. the same four banks eight times over.  XORing the
  // low bits of the row with two higher bit fields spreads every one of those patterns (and the bit-reversed rows of the
  // store phase) over all the banks; consecutive rows keep consecutive places, so the load phase stays conflict-free.
  constexpr uint32_t GMASK = (64u >> LOG_T) - 1u;  // row groups per 64 words, minus one
  auto at = [&](uint32_t row, uint32_t c) { return ((row ^ (((row >> 3) ^ (row >> 5)) & GMASK)) << LOG_T) + c; };
  auto root = [&](uint32_t j) { return g_load29(a.pq + ((size_t)j << a.pq_shift)); };  // (w_n^(n / D))^j, canonical
  const uint32_t n = 1u << a.log_n;
  const uint32_t t = n >> DEG;
  const uint32_t p = 1u << a.lgp;
  const uint32_t batch = blockIdx.y;
  const Fr* in = a.in + (size_t)batch * a.in_stride;
  Fr* out = a.out + (size_t)batch * a.out_stride;
  const uint32_t index0 = blockIdx.x * T;
  const bool first = a.lgp == 0, last = a.next_deg == 0;
  const bool load_mul = first && (a.flags & NTT_IN_COSET);
  if (load_mul && threadIdx.x < 3) {
    Fr29 cl = const29(CONSTS29<FrP>.from256);
    if (threadIdx.x) cl = Fr29::mul(Fr29::unpack(a.in_coset[threadIdx.x - 1].v.l), const29(CONSTS29<FrP>.c271));
    CQ_UNROLL for (int l = 0; l < 9; l++) cl29[threadIdx.x * 9 + l] = cl.a[l];
  }
  if (last && threadIdx.x >= 64 && threadIdx.x < 67) {
    const uint32_t m = threadIdx.x - 64;
    const Fr one = Fr::one();
    const uint32_t* src = (a.flags & NTT_OUT_MUL) ? ((a.flags & NTT_OUT_COSET) ? a.out_mul[m].v.l : a.out_mul[0].v.l) : one.v.l;
    const Fr29 cm = Fr29::unpack(src);
    CQ_UNROLL for (int l = 0; l < 9; l++) cs29[m * 9 + l] = cm.a[l];
  }
  if (load_mul) __syncthreads();
  for (uint32_t e = threadIdx.x; e < E; e += TH) {
    const uint32_t g = index0 + (e & (T - 1)) + (e >> LOG_T) * t;
    Fr29 x = Fr29::zero();
    if (g < a.in_len) {
      x = g_load29(in + g);
      if (load_mul) x = Fr29::mul(x, lds_load29(cl29 + (g % 3) * 9, 1, 0));
    }
    lds_store29(smem29, E, at(e >> LOG_T, e & (T - 1)), x);
  }
  __syncthreads();

  constexpr uint32_t R4 = DEG / 2;
#pragma unroll
  for (uint32_t st = 0; st < R4; st++) {
    const uint32_t rnd = 2 * st;                      // structural level of the step's first half
    const uint32_t brnd = st < 3 ? rnd : rnd - 5;     // level whose bounds apply: 0, 2, 4, then (after the reduction) 1
    constexpr uint32_t RED_STEP = 2;                  // the step after which B would reach 128
    const bool reduce = st == RED_STEP;
    const uint32_t bit = half >> rnd;
    const uint32_t hb = bit >> 1;
    for (uint32_t w = threadIdx.x; w < (half >> 1) * T; w += TH) {
      const uint32_t c = w & (T - 1);
      const uint32_t wg = w >> LOG_T;
      const uint32_t dj = wg >> rnd;
      const uint32_t r0 = (((wg & ((1u << rnd) - 1u)) * bit) << 1) + dj;
      const uint32_t o0 = at(r0, c), o1 = at(r0 + hb, c), o2 = at(r0 + bit, c), o3 = at(r0 + bit + hb, c);
      const Fr29 x0 = lds_load29(smem29, E, o0), x1 = lds_load29(smem29, E, o1);
      const Fr29 x2 = lds_load29(smem29, E, o2), x3 = lds_load29(smem29, E, o3);
      const Fr29 s02 = x0 + x2, s13 = x1 + x3;
      Fr29 d02 = sub_level(x0, x2, brnd);
      if (dj) d02 = Fr29::mul(d02, root(dj << rnd));
      const Fr29 d13 = Fr29::mul(sub_level(x1, x3, brnd), root((dj + hb) << rnd));
      Fr29 y0 = s02 + s13;
      y0.normalise();
      Fr29 y1 = sub_level(s02, s13, brnd + 1);
      Fr29 y2 = d02 + d13;
      y2.normalise();
      Fr29 y3 = sub_level(d02, d13, brnd + 1);
      if (dj) {
        const Fr29 w2 = root(dj << (rnd + 1));
        y1 = Fr29::mul(y1, w2);
        y3 = Fr29::mul(y3, w2);
      } else if (reduce) {  // not multiplied by a root in this step: bring them back below 2 p by hand
        y1 = y1.reduced();
        y2 = y2.reduced();
        y3 = y3.reduced();
      }
      if (reduce) y0 = y0.reduced();  // < 128 p -> < 2 p
      lds_store29(smem29, E, o0, y0);
      lds_store29(smem29, E, o1, y1);
      lds_store29(smem29, E, o2, y2);
      lds_store29(smem29, E, o3, y3);
    }
    __syncthreads();
  }
  if (DEG & 1) {  // odd DEG: the last level on its own (bounds: level 6 of a 7-bit pass = level 1, level 8 of a 9-bit pass = level 3)
    constexpr uint32_t rnd = DEG - 1;
    constexpr uint32_t brnd = rnd - 5;
    const uint32_t bit = half >> rnd;
    for (uint32_t w = threadIdx.x; w < half * T; w += TH) {
      const uint32_t c = w & (T - 1);
      const uint32_t wb = w >> LOG_T;
      const uint32_t di = wb >> rnd;
      const uint32_t b = ((wb & ((1u << rnd) - 1u)) * bit) | di;
      const uint32_t i0 = (b << 1) - di;
      const uint32_t i1 = i0 + bit;
      const Fr29 u = lds_load29(smem29, E, at(i0, c));
      const Fr29 v = lds_load29(smem29, E, at(i1, c));
      Fr29 sm = u + v;
      sm.normalise();
      Fr29 d = sub_level(u, v, brnd);
      if (di) d = Fr29::mul(d, root(di << rnd));
      lds_store29(smem29, E, at(i0, c), sm);
      lds_store29(smem29, E, at(i1, c), d);
    }
    __syncthreads();
  }

  const uint32_t lgp2 = a.lgp + DEG;
  const uint32_t log_t2 = a.log_n - a.next_deg;
  for (uint32_t e = threadIdx.x; e < E; e += TH) {
    const uint32_t c = e & (T - 1);
    const uint32_t i = e >> LOG_T;
    const uint32_t index = index0 + c;
    const uint32_t k = index & (p - 1);
    const uint32_t g = ((index - k) << DEG) + k + i * p;
    if (g >= a.out_len) continue;
    const Fr29 x = lds_load29(smem29, E, at(bitrev(i, DEG), c));  // < 64 p
    if (last) {
      const uint32_t m = (a.flags & NTT_OUT_COSET) ? g % 3 : 0;
      g_store29(out + g, Fr29::mul(x, lds_load29(cs29 + m * 9, 1, 0)), true);
      continue;
    }
    const uint32_t i2 = g >> log_t2, index2 = g & ((1u << log_t2) - 1);
    const uint32_t ex = ((n >> lgp2) >> a.next_deg) * (index2 & ((1u << lgp2) - 1)) * i2;
    Fr29 w;
    if (!ex) {
      w = const29(CONSTS29<FrP>.one);
    } else if (a.tw_full) {
      w = g_load29(a.tw_full + ex);
    } else {
      w = g_load29(a.tw_lo + (ex & ((1u << a.tw_l) - 1)));
      const uint32_t h = ex >> a.tw_l;
      if (h) w = canon29(Fr29::mul(w, g_load29(a.tw_hi + h)));
    }
    g_store29(out + g, Fr29::mul(x, w), false);
  }
}

// ---- the same pass, software-pipelined over several tiles per workgroup -------------------------------------------------
// The kernel above runs one tile per workgroup: load (HBM-bound), butterflies (VALU-bound), store (products + HBM), and the
// four workgroups of a CU start in step, so a pass costs the SUM of its memory and arithmetic phases (25 + 53 us per pass of
// a 2^18 x 8 batch).  Here a workgroup is persistent: it walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and holds the
// NEXT tile's elements in registers (PER x 32 B per thread, issued right after the current tile has been written to LDS)
// while it runs the current tile's butterflies and store.  What makes that work:
//   * barriers are raw `s_barrier`s that wait for LDS traffic only (lgkmcnt): `__syncthreads()` also waits vmcnt(0), which
//     would drain the prefetch at the first butterfly level and make the stores of a tile synchronous;
//   * the per-pass constants (roots, load / store factors) go to LDS once per workgroup, not once per tile.
// Same arithmetic, same element order, bit-identical output.
#define CQ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <uint32_t DEG, uint32_t LOG_T>
#ifndef CQ_NTT_PIPE_WAVES
#define CQ_NTT_PIPE_WAVES 0  // 4: cap the kernel at 128 registers (four waves per SIMD), at the price of spills
#endif
#ifndef CQ_NTT_PIPE_LATE
#define CQ_NTT_PIPE_LATE 0   // 1: issue the next tile's loads before the store phase instead of before the butterflies
#endif
#if CQ_NTT_PIPE_WAVES
#define CQ_NTT_PIPE_ATTR __attribute__((amdgpu_waves_per_eu(CQ_NTT_PIPE_WAVES, CQ_NTT_PIPE_WAVES)))
#else
#define CQ_NTT_PIPE_ATTR
#endif
__global__ __launch_bounds__(NTT_THREADS) CQ_NTT_PIPE_ATTR void ntt_pass_pipe_kernel(NttPassArgs a, uint32_t tiles_per_col, uint32_t total_tiles) {
  extern __shared__ uint32_t smem29[];
  constexpr uint32_t D = 1u << DEG, T = 1u << LOG_T, E = D * T, PER = E / NTT_THREADS, half = D >> 1;
  static_assert(E % NTT_THREADS == 0 && PER >= 1 && PER <= 8, "tile elements per thread");
  uint32_t* const tw29 = smem29 + 9 * E;
  uint32_t* const cl29 = tw29 + 9 * (D / 2 + 1);
  uint32_t* const cs29 = cl29 + 27;
  const uint32_t n = 1u << a.log_n;
  const uint32_t t = n >> DEG;
  const uint32_t p = 1u << a.lgp;
  const bool first = a.lgp == 0, last = a.next_deg == 0;
  const bool load_mul = first && (a.flags & NTT_IN_COSET);

  // ---- once per workgroup: constants and roots ----
  if (load_mul && threadIdx.x < 3) {
    Fr29 cl = const29(CONSTS29<FrP>.from256);
    if (threadIdx.x) cl = Fr29::mul(Fr29::unpack(a.in_coset[threadIdx.x - 1].v.l), const29(CONSTS29<FrP>.c271));
    CQ_UNROLL for (int l = 0; l < 9; l++) cl29[threadIdx.x * 9 + l] = cl.a[l];
  }
  if (last && threadIdx.x >= 64 && threadIdx.x < 67) {
    const uint32_t m = threadIdx.x - 64;
    const Fr one = Fr::one();
    const uint32_t* src = (a.flags & NTT_OUT_MUL) ? ((a.flags & NTT_OUT_COSET) ? a.out_mul[m].v.l : a.out_mul[0].v.l) : one.v.l;
    const Fr29 cm = Fr29::unpack(src);
    CQ_UNROLL for (int l = 0; l < 9; l++) cs29[m * 9 + l] = cm.a[l];
  }
  for (uint32_t j = threadIdx.x; j < half; j += NTT_THREADS) {
    const Fr29 w = g_load29(a.pq + ((size_t)j << a.pq_shift));
    CQ_UNROLL for (int l = 0; l < 9; l++) tw29[l * half + j] = w.a[l];
  }

  // the next tile's elements, packed as they are in memory (zero where the input is shorter than the transform)
  uint4 pre[PER][2];
  auto issue = [&](uint32_t tile_id) {
    const uint32_t batch = tile_id / tiles_per_col, index0 = (tile_id - batch * tiles_per_col) * T;
    const Fr* in = a.in + (size_t)batch * a.in_stride;
    CQ_UNROLL for (uint32_t q = 0; q < PER; q++) {
      const uint32_t e = threadIdx.x + q * NTT_THREADS;
      const uint32_t g = index0 + (e & (T - 1)) + (e >> LOG_T) * t;
      if (g < a.in_len) {
        const uint4* src = reinterpret_cast<const uint4*>(in + g);
        pre[q][0] = src[0];
        pre[q][1] = src[1];
      } else {
        pre[q][0] = make_uint4(0, 0, 0, 0);
        pre[q][1] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  uint32_t tile_id = blockIdx.x;
  if (tile_id < total_tiles) issue(tile_id);
  CQ_LDS_BARRIER();  // constants and roots in place

  for (; tile_id < total_tiles; tile_id += gridDim.x) {
    const uint32_t batch = tile_id / tiles_per_col, index0 = (tile_id - batch * tiles_per_col) * T;
    Fr* out = a.out + (size_t)batch * a.out_stride;
    // ---- registers -> LDS limb planes (the first pass of a coset transform converts and shifts here) ----
    CQ_UNROLL for (uint32_t q = 0; q < PER; q++) {
      const uint32_t e = threadIdx.x + q * NTT_THREADS;
      const uint32_t w8[8] = {pre[q][0].x, pre[q][0].y, pre[q][0].z, pre[q][0].w, pre[q][1].x, pre[q][1].y, pre[q][1].z, pre[q][1].w};
      Fr29 x = Fr29::unpack(w8);
      if (load_mul) {
        const uint32_t g = index0 + (e & (T - 1)) + (e >> LOG_T) * t;
        if (g < a.in_len) x = Fr29::mul(x, lds_load29(cl29 + (g % 3) * 9, 1, 0));
      }
      lds_store29(smem29, E, e, x);  // row e >> LOG_T, column e & (T - 1)
    }
    CQ_LDS_BARRIER();
#if !CQ_NTT_PIPE_LATE
    {
      const uint32_t next = tile_id + gridDim.x;  // in flight through the butterflies and the store of this tile
      if (next < total_tiles) issue(next);
    }
#endif

    // ---- 2^DEG-point DIF in LDS (as in ntt_pass_kernel) ----
    constexpr uint32_t R4 = DEG / 2;
#pragma unroll
    for (uint32_t st = 0; st < R4; st++) {
      const uint32_t rnd = 2 * st;
      const uint32_t bit = half >> rnd;
      const uint32_t hb = bit >> 1;
      for (uint32_t w = threadIdx.x; w < (half >> 1) * T; w += NTT_THREADS) {
        const uint32_t c = w & (T - 1);
        const uint32_t wg = w >> LOG_T;
        const uint32_t dj = wg >> rnd;
        const uint32_t r0 = (((wg & ((1u << rnd) - 1u)) * bit) << 1) + dj;
        const uint32_t o0 = r0 * T + c, o1 = o0 + hb * T, o2 = o0 + bit * T, o3 = o2 + hb * T;
        const Fr29 x0 = lds_load29(smem29, E, o0), x1 = lds_load29(smem29, E, o1);
        const Fr29 x2 = lds_load29(smem29, E, o2), x3 = lds_load29(smem29, E, o3);
        const Fr29 s02 = x0 + x2, s13 = x1 + x3;
        Fr29 d02 = sub_level(x0, x2, rnd);
        if (dj) d02 = Fr29::mul(d02, lds_load29(tw29, half, dj << rnd));
        const Fr29 d13 = Fr29::mul(sub_level(x1, x3, rnd), lds_load29(tw29, half, (dj + hb) << rnd));
        Fr29 y0 = s02 + s13;
        y0.normalise();
        Fr29 y1 = sub_level(s02, s13, rnd + 1);
        Fr29 y2 = d02 + d13;
        y2.normalise();
        Fr29 y3 = sub_level(d02, d13, rnd + 1);
        if (dj) {
          const Fr29 w2 = lds_load29(tw29, half, dj << (rnd + 1));
          y1 = Fr29::mul(y1, w2);
          y3 = Fr29::mul(y3, w2);
        }
        lds_store29(smem29, E, o0, y0);
        lds_store29(smem29, E, o1, y1);
        lds_store29(smem29, E, o2, y2);
        lds_store29(smem29, E, o3, y3);
      }
      CQ_LDS_BARRIER();
    }
    if (DEG & 1) {  // odd DEG: the last level on its own
      constexpr uint32_t rnd = DEG - 1;
      const uint32_t bit = half >> rnd;
      for (uint32_t w = threadIdx.x; w < half * T; w += NTT_THREADS) {
        const uint32_t c = w & (T - 1);
        const uint32_t wb = w >> LOG_T;
        const uint32_t di = wb >> rnd;
        const uint32_t b = ((wb & ((1u << rnd) - 1u)) * bit) | di;
        const uint32_t i0 = (b << 1) - di;
        const uint32_t i1 = i0 + bit;
        const Fr29 u = lds_load29(smem29, E, i0 * T + c);
        const Fr29 v = lds_load29(smem29, E, i1 * T + c);
        Fr29 sm = u + v;
        sm.normalise();
        Fr29 d = sub_level(u, v, rnd);
        if (di) d = Fr29::mul(d, lds_load29(tw29, half, di << rnd));
        lds_store29(smem29, E, i0 * T + c, sm);
        lds_store29(smem29, E, i1 * T + c, d);
      }
      CQ_LDS_BARRIER();
    }

#if CQ_NTT_PIPE_LATE
    {
      const uint32_t next = tile_id + gridDim.x;  // in flight through the store of this tile only (fewer live registers)
      if (next < total_tiles) issue(next);
    }
#endif
    // ---- store (as in ntt_pass_kernel) ----
    const uint32_t lgp2 = a.lgp + DEG;
    const uint32_t log_t2 = a.log_n - a.next_deg;
    CQ_UNROLL for (uint32_t q = 0; q < PER; q++) {
      const uint32_t e = threadIdx.x + q * NTT_THREADS;
      const uint32_t c = e & (T - 1);
      const uint32_t i = e >> LOG_T;
      const uint32_t index = index0 + c;
      const uint32_t k = index & (p - 1);
      const uint32_t g = ((index - k) << DEG) + k + i * p;
      if (g >= a.out_len) continue;
      const Fr29 x = lds_load29(smem29, E, bitrev(i, DEG) * T + c);
      if (last) {
        const uint32_t m = (a.flags & NTT_OUT_COSET) ? g % 3 : 0;
        g_store29(out + g, Fr29::mul(x, lds_load29(cs29 + m * 9, 1, 0)), true);
        continue;
      }
      const uint32_t i2 = g >> log_t2, index2 = g & ((1u << log_t2) - 1);
      const uint32_t ex = ((n >> lgp2) >> a.next_deg) * (index2 & ((1u << lgp2) - 1)) * i2;
      Fr29 w;
      if (!ex) {
        w = const29(CONSTS29<FrP>.one);
      } else if (a.tw_full) {
        w = g_load29(a.tw_full + ex);
      } else {
        w = g_load29(a.tw_lo + (ex & ((1u << a.tw_l) - 1)));
        const uint32_t h = ex >> a.tw_l;
        if (h) w = canon29(Fr29::mul(w, g_load29(a.tw_hi + h)));
      }
      g_store29(out + g, Fr29::mul(x, w), false);
    }
    CQ_LDS_BARRIER();  // every read of this tile's LDS image is done (its stores stay in flight)
  }
}

// ---------------------------------------------------------------------------------------------
NttTables::~NttTables() {
  if (tw_lo) hipFree(tw_lo);
  if (tw_hi) hipFree(tw_hi);
  if (tw_full) hipFree(tw_full);
  if (pq) hipFree(pq);
}

int NttTables::build(uint32_t log_n_, const Fr& omega_, hipStream_t stream) {
  log_n = log_n_;
  omega = omega_;
  tw_l = (log_n + 1) / 2;
  const uint32_t lo_cnt = 1u << tw_l;
  const uint32_t hi_cnt = 1u << (log_n - tw_l);
  pq_log = log_n < NTT_WIDE_MAX_DEG ? log_n : NTT_WIDE_MAX_DEG;  // roots for the largest in-LDS FFT, wide passes included
  const uint32_t pq_cnt = pq_log ? (1u << (pq_log - 1)) : 1;
  if (hipMalloc(&tw_lo, sizeof(Fr) * lo_cnt) != hipSuccess) return -1;
  if (hipMalloc(&tw_hi, sizeof(Fr) * hi_cnt) != hipSuccess) return -1;
  if (hipMalloc(&pq, sizeof(Fr) * pq_cnt) != hipSuccess) return -1;
  fr_powers_kernel<<<(lo_cnt + 255) / 256, 256, 0, stream>>>(tw_lo, omega, 1, lo_cnt);
  fr_powers_kernel<<<(hi_cnt + 255) / 256, 256, 0, stream>>>(tw_hi, omega, (uint64_t)lo_cnt, hi_cnt);
  // full table (the kernels are bound by field products, not by HBM: a 32-byte load is cheaper than the product that
  // composes the two-level entry); skipped for very large domains and when memory is short
  // (CQ_NTT_NO_FULL_TABLE: lets the tests exercise the two-level path that domains above 2^24 take)
  if (log_n >= 8 && log_n <= 24 && !getenv("CQ_NTT_NO_FULL_TABLE") && hipMalloc(&tw_full, sizeof(Fr) << log_n) == hipSuccess)
    fr_powers_kernel<<<((1u << log_n) + 255) / 256, 256, 0, stream>>>(tw_full, omega, 1, 1u << log_n);
  else
    (void)hipGetLastError();
  // pq[j] = (w_n^(n / 2^pq_log))^j : roots for the largest in-LDS FFT
  fr_powers_kernel<<<(pq_cnt + 255) / 256, 256, 0, stream>>>(pq, omega, (uint64_t)1 << (log_n - pq_log), pq_cnt);
  // the passes compute on the lazy 29-bit field: tables in its (canonical) R' = 2^261 form
  ntt_table_to29_kernel<<<(lo_cnt + 255) / 256, 256, 0, stream>>>(tw_lo, lo_cnt);
  ntt_table_to29_kernel<<<(hi_cnt + 255) / 256, 256, 0, stream>>>(tw_hi, hi_cnt);
  if (tw_full) ntt_table_to29_kernel<<<((1u << log_n) + 255) / 256, 256, 0, stream>>>(tw_full, 1u << log_n);
  ntt_table_to29_kernel<<<(pq_cnt + 255) / 256, 256, 0, stream>>>(pq, pq_cnt);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Split log_n into passes.  Each pass resolves `deg` <= NTT_MAX_DEG bits; tiles are
// 2^deg rows x T columns with at most NTT_TILE_ELEMS elements.
int ntt_run(const NttTables& tb, const Fr* in, Fr* out, Fr* scratch, const NttIo& io, hipStream_t stream) {
  const uint32_t log_n = tb.log_n;
  const uint32_t n = 1u << log_n;
  uint32_t npass = log_n == 0 ? 1 : (log_n + NTT_MAX_DEG - 1) / NTT_MAX_DEG;
  uint32_t degs[8];
  // Wide passes (7..9 bits, ntt_pass_wide_kernel): fewer trips through HBM -- 2^18 in two passes (9 + 9), 2^19..2^21 in three
  // (7 + 6 + 6 .. 7 + 7 + 7).  Measured on MI355X, 8 columns, Gelem/s wide / 4..6-bit plan (profiles/r03_ntt_wide_ab.txt):
  // 2^17 8.7 / 8.3, 2^18 9.0 / 8.9, 2^19 7.8 / 7.6, 2^20 7.4 / 7.6, 2^21 7.4 / 7.9, 2^22 7.6 / 8.3 -- a pass is bound by VALU
  // issue and LDS latency, not by HBM, so a saved trip is worth little and the nine-bit pass's 128-byte runs and extra
  // reducing products take it back from 2^20 on.  Inside a proof, where the transforms share the GPU with MSM tails and
  // the inversion, the wide plan is no faster at any size (k = 18: 8.0-8.2 ms against 7.8-8.0; two 74-KB workgroups of 512
  // threads per CU co-schedule less freely than four 37-KB ones).  Default: OFF; CQ_NTT_WIDE=1 turns it on (A/B, tests).
  static const int wide_env = getenv("CQ_NTT_WIDE") ? atoi(getenv("CQ_NTT_WIDE")) : 0;
  const bool wide_on = wide_env != 0;
  const uint32_t npass_wide = (log_n + NTT_WIDE_MAX_DEG - 1) / NTT_WIDE_MAX_DEG;
  const bool wide = wide_on && NTT_MAX_DEG == 6 && NTT_TILE_ELEMS == 1024 && log_n >= 13 && npass_wide < npass;
  if (wide) {
    npass = npass_wide;
    uint32_t rem = log_n;
    for (uint32_t i = 0; i < npass; i++) {  // widest passes first: 19 = 7 + 6 + 6, 22 = 8 + 7 + 7
      degs[i] = (rem + (npass - i) - 1) / (npass - i);
      rem -= degs[i];
    }
  } else {
    uint32_t rem = log_n;
    for (uint32_t i = 0; i < npass; i++) {
      uint32_t d = (rem + (npass - i) - 1) / (npass - i);
      degs[i] = d;
      rem -= d;
    }
    // Prefer even pass widths (6 and 4 bits: whole radix-4 steps, one LDS round trip per two levels) with as few 5-bit
    // passes as the sum allows: 2^20 = 6 + 6 + 4 + 4 instead of 5 + 5 + 5 + 5 (ten round trips instead of twelve).
    if (NTT_MAX_DEG == 6 && log_n >= 12) {
      for (uint32_t c5 = 0; c5 <= npass; c5++) {
        bool found = false;
        for (uint32_t a6 = 0; a6 + c5 <= npass; a6++) {
          const uint32_t b4 = npass - c5 - a6;
          if (6 * a6 + 5 * c5 + 4 * b4 != log_n) continue;
          uint32_t i = 0;
          for (uint32_t q = 0; q < a6; q++) degs[i++] = 6;
          for (uint32_t q = 0; q < c5; q++) degs[i++] = 5;
          for (uint32_t q = 0; q < b4; q++) degs[i++] = 4;
          found = true;
          break;
        }
        if (found) break;
      }
    }
  }
  // intermediate passes ping-pong between the two halves of `scratch`; only the last pass writes
  // `out` (which may be shorter than n when the result is truncated, and may alias `in`).
  Fr* sbuf[2] = {scratch, scratch + (size_t)io.batch * n};
  const Fr* src = in;
  uint32_t lgp = 0;
  for (uint32_t ps = 0; ps < npass; ps++) {
    const bool last = (ps + 1 == npass);
    Fr* dst = last ? out : sbuf[ps & 1];
    NttPassArgs a;
    a.in = src;
    a.out = dst;
    a.in_stride = (ps == 0) ? io.in_stride : (size_t)n;
    a.out_stride = last ? io.out_stride : (size_t)n;
    a.log_n = log_n;
    a.lgp = lgp;
    a.deg = degs[ps];
    a.next_deg = last ? 0 : degs[ps + 1];
    const uint32_t t = n >> degs[ps];
    uint32_t log_t = 0;
    const bool wide_pass = degs[ps] > NTT_MAX_DEG;
    // (CQ_NTT_PIPE_TILE=512: half-size tiles for the pipelined kernel -- twice the tiles per persistent workgroup; A/B knob)
    static const uint32_t tile_elems = (getenv("CQ_NTT_PIPE_TILE") && atoi(getenv("CQ_NTT_PIPE_TILE")) == 512 && NTT_TILE_ELEMS == 1024) ? 512u : NTT_TILE_ELEMS;
    while ((1u << (log_t + 1)) <= t && ((1u << (log_t + 1)) << degs[ps]) <= (wide_pass ? (1u << NTT_WIDE_TILE_LOG) : tile_elems)) log_t++;
    a.log_t = log_t;
    a.tw_lo = tb.tw_lo;
    a.tw_hi = tb.tw_hi;
    a.tw_full = tb.tw_full;
    a.tw_l = tb.tw_l;
    a.pq = tb.pq;
    a.pq_shift = tb.pq_log - degs[ps];
    a.flags = io.critical ? NTT_CRITICAL : 0;
    a.in_len = n;
    a.out_len = n;
    if (ps == 0) {
      a.in_len = io.in_len;
      if (io.in_coset) {
        a.flags |= NTT_IN_COSET;
        a.in_coset[0] = io.in_coset_mul[0];
        a.in_coset[1] = io.in_coset_mul[1];
      }
    }
    if (last) {
      a.out_len = io.out_len;
      // a transform without a coset shift on its way in ran on a / 32 (see "arithmetic of the passes")
      const Fr fix = io.in_coset ? Fr::one() : Fr::from_u64(32);
      if (io.out_mul || !io.in_coset) {
        a.flags |= NTT_OUT_MUL;
        a.out_mul[0] = (io.out_mul ? io.out_mul_v[0] : Fr::one()) * fix;
        if (io.out_mul && io.out_coset) {
          a.flags |= NTT_OUT_COSET;
          a.out_mul[1] = io.out_mul_v[1] * fix;
          a.out_mul[2] = io.out_mul_v[2] * fix;
        }
      }
    }
    const uint32_t T = 1u << log_t;
    dim3 grid(t / T, io.batch);
    const size_t lds = ((size_t)(T << degs[ps]) + ((size_t)1 << degs[ps]) / 2 + 1) * 36 + 54 * 4;  // nine 32-bit limb planes: tile + roots; constants
    hipEvent_t pe = io.prof ? io.prof->prof_begin(CQ_PROF_NTT_PASS) : nullptr;
    // templated shapes: full tiles (2^LT elements) of 6-, 5- and 4-bit passes
    constexpr uint32_t LT = NTT_TILE_ELEMS == 512 ? 9 : NTT_TILE_ELEMS == 2048 ? 11 : 10;
    static_assert(NTT_TILE_ELEMS == (1u << LT), "NTT_TILE_ELEMS: 512, 1024 or 2048");
    // CQ_NTT_PIPE=1: the pipelined kernel on a persistent grid for full tiles of the templated shapes.  OFF by default:
    // measured on MI355X (tools/ab_ntt_pipe.sh, gpurun_out/r3d/ab_ntt.log) it is SLOWER than the plain kernel -- 2^18 x 8:
    // 8.2-8.4 against 8.75 Gelem/s, 2^20 x 8: 7.3 against 7.6 -- because the 32 registers of the prefetch take the kernel
    // from 95 to 138 VGPRs, i.e. from four to three waves per SIMD, and the LDS round trips of the butterflies need the
    // waves more than the loads need hiding; capped at 128 registers it spills (7.6), with the loads issued only before
    // the store phase 8.2-8.3, with half-size tiles 7.0.  Kept for the record and for A/B on other shapes.
    static const bool pipe = getenv("CQ_NTT_PIPE") && getenv("CQ_NTT_PIPE")[0] == '1';
    static const uint32_t resident = []() {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      const char* e = getenv("CQ_NTT_PIPE_WG_PER_CU");
      return (uint32_t)cus * (uint32_t)(e ? atoi(e) : 4);
    }();
    const uint32_t tiles_per_col = t / T, total_tiles = tiles_per_col * io.batch;
    const uint32_t pgrid = total_tiles < resident ? total_tiles : resident;
    if (wide_pass) {
      // (log_n >= 13 and deg <= 9: a full 2048-element tile always exists)  73.9 KB of dynamic LDS: above the 64-KB default
      const size_t wlds = ((size_t)1 << NTT_WIDE_TILE_LOG) * 36 + 54 * 4;
      static const bool attr_ok = []() {
        const int bytes = (1 << NTT_WIDE_TILE_LOG) * 36 + 54 * 4;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_wide_kernel<9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_wide_kernel<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_wide_kernel<7, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
      }();
      if (!attr_ok) return -1;
      if (degs[ps] == 9 && log_t == 2) ntt_pass_wide_kernel<9, 2><<<grid, NTT_WIDE_THREADS, wlds, stream>>>(a);
      else if (degs[ps] == 8 && log_t == 3) ntt_pass_wide_kernel<8, 3><<<grid, NTT_WIDE_THREADS, wlds, stream>>>(a);
      else if (degs[ps] == 7 && log_t == 4) ntt_pass_wide_kernel<7, 4><<<grid, NTT_WIDE_THREADS, wlds, stream>>>(a);
      else return -1;
    } else
    if (pipe && tile_elems == 512 && degs[ps] == 6 && log_t == 3) ntt_pass_pipe_kernel<6, 3><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (pipe && tile_elems == 512 && degs[ps] == 5 && log_t == 4) ntt_pass_pipe_kernel<5, 4><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (pipe && tile_elems == 512 && degs[ps] == 4 && log_t == 5) ntt_pass_pipe_kernel<4, 5><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (pipe && degs[ps] == 6 && log_t == LT - 6) ntt_pass_pipe_kernel<6, LT - 6><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (pipe && degs[ps] == 5 && log_t == LT - 5) ntt_pass_pipe_kernel<5, LT - 5><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (pipe && degs[ps] == 4 && log_t == LT - 4) ntt_pass_pipe_kernel<4, LT - 4><<<pgrid, NTT_THREADS, lds, stream>>>(a, tiles_per_col, total_tiles);
    else if (degs[ps] == 6 && log_t == LT - 6) ntt_pass_kernel<6, LT - 6><<<grid, NTT_THREADS, lds, stream>>>(a);
    else if (degs[ps] == 5 && log_t == LT - 5) ntt_pass_kernel<5, LT - 5><<<grid, NTT_THREADS, lds, stream>>>(a);
    else if (degs[ps] == 4 && log_t == LT - 4) ntt_pass_kernel<4, LT - 4><<<grid, NTT_THREADS, lds, stream>>>(a);
    else ntt_pass_kernel<0, 0><<<grid, NTT_THREADS, lds, stream>>>(a);
    if (io.prof) io.prof->prof_end(pe);
    src = dst;
    lgp += degs[ps];
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace cq
