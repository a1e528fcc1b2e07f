// Radix-2 NTT over BN254 Fr for gfx950 -- replaces `best_fft` (halo2_proofs/src/arithmetic.rs:171-274)
// and the `EvaluationDomain` wrappers around it (poly/domain.rs:238-374).
//
// Algorithm (not the reference's bit-reverse + recursive butterflies): a Stockham auto-sort
// decomposition.  log_n is split into passes of `deg` bits.  A pass with `lgp` bits already done
// reads, for every "index" in [0, n>>deg), the 2^deg elements in[index + i*(n>>deg)], multiplies
// element i by w^((n>>lgp>>deg) * k * i) with k = index mod 2^lgp, runs a 2^deg-point
// decimation-in-frequency FFT in LDS, and writes output digit i' to
// out[((index-k)<<deg) + k + i'*2^lgp].  After the last pass the data is in natural order, so
// input and output orders match `best_fft` (natural in / natural out) with no separate
// bit-reversal pass over HBM.
//
// HBM access: a workgroup owns a tile of T consecutive `index` values, so every global access is a
// run of T consecutive 32-byte elements (T=16..32 -> 512 B..1 KiB runs), 16 B per lane.
// LDS: the tile lives as two uint4 planes (low/high 16 bytes of each element) so that consecutive
// elements fall in consecutive 16-byte slots -> conflict-free ds_read/write_b128.
// The coset shift (zeta^(i mod 3), domain.rs:347-363), zero padding to the extended domain
// (domain.rs:259), the 1/n scaling of the inverse transform (domain.rs:366-374) and the truncation
// of `extended_to_coeff` (domain.rs:311-312) are fused into the first/last pass instead of being
// separate streaming passes.
#include "ntt.hpp"
#include "ctx.hpp"

namespace cq {

__global__ void fr_powers_kernel(Fr* out, Fr base, uint64_t step, uint32_t count) {
  // out[j] = base^(j*step)
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  out[j] = base.pow_u64((uint64_t)j * step);
}

static __device__ __forceinline__ Fr lds_load(const uint4* lo, const uint4* hi, uint32_t i) {
  Fr r;
  uint4 a = lo[i], b = hi[i];
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
static __device__ __forceinline__ void lds_store(uint4* lo, uint4* hi, uint32_t i, const Fr& r) {
  lo[i] = make_uint4(r.v.l[0], r.v.l[1], r.v.l[2], r.v.l[3]);
  hi[i] = make_uint4(r.v.l[4], r.v.l[5], r.v.l[6], r.v.l[7]);
}
static __device__ __forceinline__ Fr g_load(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
static __device__ __forceinline__ void g_store(Fr* p, const Fr& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v.l[0], r.v.l[1], r.v.l[2], r.v.l[3]);
  q[1] = make_uint4(r.v.l[4], r.v.l[5], r.v.l[6], r.v.l[7]);
}

static __device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0;
}

__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs a) {
  extern __shared__ uint4 smem[];
  const uint32_t D = 1u << a.deg;
  const uint32_t T = 1u << a.log_t;
  const uint32_t E = D * T;
  uint4* lo = smem;
  uint4* hi = smem + E;
  const uint32_t n = 1u << a.log_n;
  const uint32_t t = n >> a.deg;
  const uint32_t p = 1u << a.lgp;
  const uint32_t tile = blockIdx.x;
  const uint32_t batch = blockIdx.y;
  const Fr* in = a.in + (size_t)batch * a.in_stride;
  Fr* out = a.out + (size_t)batch * a.out_stride;
  const uint32_t index0 = tile * T;

  // ---- load tile: element (row i, col c) <- in[index0 + c + i*t] * twiddle ----
  for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
    const uint32_t c = e & (T - 1);
    const uint32_t i = e >> a.log_t;
    const uint32_t index = index0 + c;
    const uint32_t g = index + i * t;
    Fr x;
    if (g < a.in_len) {
      x = g_load(in + g);
      if (a.flags & NTT_IN_COSET) {
        const uint32_t m = g % 3;
        if (m) x = x * a.in_coset[m - 1];
      }
      if (a.lgp) {
        const uint32_t k = index & (p - 1);
        // exponent of w_n: (n >> lgp >> deg) * k * i  < n
        const uint32_t ex = ((n >> a.lgp) >> a.deg) * k * i;
        if (ex) {
          if (a.tw_full) {
            x = x * g_load(a.tw_full + ex);
          } else {
            Fr w = g_load(a.tw_lo + (ex & ((1u << a.tw_l) - 1)));
            const uint32_t h = ex >> a.tw_l;
            if (h) w = w * g_load(a.tw_hi + h);
            x = x * w;
          }
        }
      }
    } else {
      x = Fr::zero();
    }
    lds_store(lo, hi, i * T + c, x);
  }
  __syncthreads();

  // ---- 2^deg-point DIF in LDS, all T columns at once ----
  const uint32_t half = D >> 1;
  for (uint32_t rnd = 0; rnd < a.deg; rnd++) {
    const uint32_t bit = half >> rnd;
    for (uint32_t w = threadIdx.x; w < half * T; w += NTT_THREADS) {
      const uint32_t c = w & (T - 1);
      const uint32_t b = w >> a.log_t;
      const uint32_t di = b & (bit - 1);
      const uint32_t i0 = (b << 1) - di;
      const uint32_t i1 = i0 + bit;
      Fr u = lds_load(lo, hi, i0 * T + c);
      Fr v = lds_load(lo, hi, i1 * T + c);
      Fr s = u + v;
      Fr d = u - v;
      if (di) d = d * g_load(a.pq + ((size_t)(di << rnd) << a.pq_shift));
      lds_store(lo, hi, i0 * T + c, s);
      lds_store(lo, hi, i1 * T + c, d);
    }
    __syncthreads();
  }

  // ---- store: output digit i' (bit-reversed LDS row) -> out[((index-k)<<deg) + k + i'*p] ----
  for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
    const uint32_t c = e & (T - 1);
    const uint32_t i = e >> a.log_t;
    const uint32_t index = index0 + c;
    const uint32_t k = index & (p - 1);
    const uint32_t g = ((index - k) << a.deg) + k + i * p;
    if (g < a.out_len) {
      Fr x = lds_load(lo, hi, bitrev(i, a.deg) * T + c);
      if (a.flags & NTT_OUT_MUL) x = x * a.out_mul[(a.flags & NTT_OUT_COSET) ? (g % 3) : 0];
      g_store(out + g, x);
    }
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
  pq_log = log_n < NTT_MAX_DEG ? log_n : NTT_MAX_DEG;
  const uint32_t pq_cnt = pq_log ? (1u << (pq_log - 1)) : 1;
  if (hipMalloc(&tw_lo, sizeof(Fr) * lo_cnt) != hipSuccess) return -1;
  if (hipMalloc(&tw_hi, sizeof(Fr) * hi_cnt) != hipSuccess) return -1;
  if (hipMalloc(&pq, sizeof(Fr) * pq_cnt) != hipSuccess) return -1;
  fr_powers_kernel<<<(lo_cnt + 255) / 256, 256, 0, stream>>>(tw_lo, omega, 1, lo_cnt);
  fr_powers_kernel<<<(hi_cnt + 255) / 256, 256, 0, stream>>>(tw_hi, omega, (uint64_t)lo_cnt, hi_cnt);
  // full table (the kernels are bound by field products, not by HBM: a 32-byte load is cheaper than the product that
  // composes the two-level entry); skipped for very large domains and when memory is short
  if (log_n >= 8 && log_n <= 24 && hipMalloc(&tw_full, sizeof(Fr) << log_n) == hipSuccess)
    fr_powers_kernel<<<((1u << log_n) + 255) / 256, 256, 0, stream>>>(tw_full, omega, 1, 1u << log_n);
  else
    (void)hipGetLastError();
  // pq[j] = (w_n^(n / 2^pq_log))^j : roots for the largest in-LDS FFT
  fr_powers_kernel<<<(pq_cnt + 255) / 256, 256, 0, stream>>>(pq, omega, (uint64_t)1 << (log_n - pq_log), pq_cnt);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Split log_n into passes.  Each pass resolves `deg` <= NTT_MAX_DEG bits; tiles are
// 2^deg rows x T columns with at most NTT_TILE_ELEMS elements.
int ntt_run(const NttTables& tb, const Fr* in, Fr* out, Fr* scratch, const NttIo& io, hipStream_t stream) {
  const uint32_t log_n = tb.log_n;
  const uint32_t n = 1u << log_n;
  uint32_t npass = log_n == 0 ? 1 : (log_n + NTT_MAX_DEG - 1) / NTT_MAX_DEG;
  uint32_t degs[8];
  {
    uint32_t rem = log_n;
    for (uint32_t i = 0; i < npass; i++) {
      uint32_t d = (rem + (npass - i) - 1) / (npass - i);
      degs[i] = d;
      rem -= d;
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
    const uint32_t t = n >> degs[ps];
    uint32_t log_t = 0;
    while ((1u << (log_t + 1)) <= t && ((1u << (log_t + 1)) << degs[ps]) <= NTT_TILE_ELEMS) log_t++;
    a.log_t = log_t;
    a.tw_lo = tb.tw_lo;
    a.tw_hi = tb.tw_hi;
    a.tw_full = tb.tw_full;
    a.tw_l = tb.tw_l;
    a.pq = tb.pq;
    a.pq_shift = tb.pq_log - degs[ps];
    a.flags = 0;
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
      if (io.out_mul) {
        a.flags |= NTT_OUT_MUL;
        a.out_mul[0] = io.out_mul_v[0];
        if (io.out_coset) {
          a.flags |= NTT_OUT_COSET;
          a.out_mul[1] = io.out_mul_v[1];
          a.out_mul[2] = io.out_mul_v[2];
        }
      }
    }
    const uint32_t T = 1u << log_t;
    dim3 grid(t / T, io.batch);
    const size_t lds = (size_t)(T << degs[ps]) * 32;
    hipEvent_t pe = io.prof ? io.prof->prof_begin(CQ_PROF_NTT_PASS) : nullptr;
    ntt_pass_kernel<<<grid, NTT_THREADS, lds, stream>>>(a);
    if (io.prof) io.prof->prof_end(pe);
    src = dst;
    lgp += degs[ps];
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace cq
