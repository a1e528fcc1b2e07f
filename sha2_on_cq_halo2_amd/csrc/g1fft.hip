// FFT over G1 for gfx950: `best_fft::<G1>` as `g_to_lagrange` uses it (halo2_proofs/src/arithmetic.rs:171-234,
// 277-301; ParamsKZG::downsize, poly/kzg/commitment.rs:480-492), and on top of it an FK-style ("fast amortized
// KZG proofs") construction of the CQ cached quotients, O(N log N) group operations instead of the O(N^2) of
// StaticTableValues::new (plonk/static_lookup.rs:78-126).  Setup-time code: every butterfly multiplies a point by a
// 254-bit twiddle (double-and-add on the lazy 9x29-bit field, curve29.hpp), one lane per butterfly, data in HBM.
#include "g1fft.hpp"
#include <vector>
#include "ctx.hpp"
#include "curve29.hpp"
#include "plonk.hpp"
#include "poly.hpp"

namespace cq {

static inline uint32_t blocks_for(size_t n) { return (uint32_t)((n + 255) / 256); }

// scalar (canonical, 8 x u32) * P, MSB first
static __device__ __forceinline__ XYZZ29 g1_mul_canonical(const XYZZ29& p, const uint32_t* k) {
  XYZZ29 acc = XYZZ29::identity();
  int top = 7;
  while (top >= 0 && k[top] == 0) top--;
  for (int w = top; w >= 0; w--) {
    const uint32_t word = k[w];
#pragma unroll 1
    for (int b = 31; b >= 0; b--) {
      acc = xyzz29_dbl(acc);
      if ((word >> b) & 1u) xyzz29_add(acc, p);
    }
  }
  return acc;
}
static __device__ __forceinline__ void ld_canon(const uint64_t* p, uint32_t* k) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
}

// affine (R = 2^256 form) -> packed XYZZ in the kernels' R' form; entries >= n_in are the identity (zero padding)
__global__ void g1_from_affine_kernel(const G1Affine* __restrict__ in, uint32_t n_in, uint32_t n, XYZZ* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ29 v = XYZZ29::identity();
  if (i < n_in) {
    const Affine29 a = load_affine29(in + i, true);
    if (!a.is_identity()) v = {a.x, a.y, Fq29::one(), Fq29::one()};
  }
  store_xyzz29(out + i, v);
}
// packed XYZZ -> canonical affine in the reference's layout (one inversion per point: setup code)
__global__ void g1_to_affine_kernel(const XYZZ* __restrict__ in, uint32_t n, G1Affine* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const XYZZ29 v = load_xyzz29(in + i);
  G1Affine r = G1Affine::identity();
  if (!v.is_identity()) {
    const Fq x = v.x.reduced().to_mont256(), y = v.y.to_mont256(), zz = v.zz.to_mont256(), zzz = v.zzz.to_mont256();
    const Fq iv = (zz * zzz).inv();
    r.x = x * (iv * zzz);
    r.y = y * (iv * zz);
  }
  uint32_t w[8];
  for (int k = 0; k < 8; k++) w[k] = r.x.v.l[k];
  st8(&out[i].x, w);
  for (int k = 0; k < 8; k++) w[k] = r.y.v.l[k];
  st8(&out[i].y, w);
}

// bit-reversal permutation (arithmetic.rs:186-192)
__global__ void g1_bitrev_kernel(XYZZ* __restrict__ a, uint32_t log_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << log_n;
  if (i >= n) return;
  const uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  if (i < r) {
    const XYZZ29 x = load_xyzz29(a + i), y = load_xyzz29(a + r);
    store_xyzz29(a + i, y);
    store_xyzz29(a + r, x);
  }
}
// one radix-2 stage: (a, b) -> (a + w b, a - w b), w = omega^(pos * n / m)   (arithmetic.rs:202-231)
__global__ __launch_bounds__(64) void g1_butterfly_kernel(XYZZ* __restrict__ a, uint32_t log_n, uint32_t stage,
                                                          const uint64_t* __restrict__ twiddles /* omega^j canonical, j < n/2 */) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << log_n;
  if (j >= n / 2) return;
  const uint32_t half = 1u << stage, m = half << 1;
  const uint32_t pos = j & (half - 1), i0 = (j >> stage) * m + pos, i1 = i0 + half;
  const uint32_t ex = pos * (n / m);
  XYZZ29 t = load_xyzz29(a + i1);
  if (ex) {
    uint32_t k[8];
    ld_canon(twiddles + 4 * (size_t)ex, k);
    t = g1_mul_canonical(t, k);
  }
  XYZZ29 u = load_xyzz29(a + i0);
  XYZZ29 s = u;
  xyzz29_add(s, t);
  if (!t.is_identity()) t.y = Fq29::neg<4>(t.y);
  xyzz29_add(u, t);
  store_xyzz29(a + i0, s);
  store_xyzz29(a + i1, u);
}
// a[i] <- scalars[i] * a[i]  (canonical scalars)
__global__ __launch_bounds__(64) void g1_scale_kernel(XYZZ* __restrict__ a, uint32_t n, const uint64_t* __restrict__ scalars, uint32_t scalar_stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  ld_canon(scalars + 4 * (size_t)i * scalar_stride, k);
  store_xyzz29(a + i, g1_mul_canonical(load_xyzz29(a + i), k));
}

__global__ void fr_powers_canonical_kernel(Fr base, Fr first, uint32_t n, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U256 c = (first * base.pow_u64(i)).to_canonical();
  uint4* q = reinterpret_cast<uint4*>(out + 4 * (size_t)i);
  q[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
  q[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
}

int g1_fft(cq_ctx* c, XYZZ* data, uint32_t log_n, const Fr& omega) {
  const uint32_t n = 1u << log_n;
  if (log_n == 0) return CQ_OK;
  void* tw;
  int rc;
  if ((rc = c->ensure_scratch(2, (size_t)(n / 2) * 32 + 64, &tw)) != CQ_OK) return rc;
  fr_powers_canonical_kernel<<<blocks_for(n / 2), 256, 0, c->stream>>>(omega, Fr::one(), n / 2, (uint64_t*)tw);
  g1_bitrev_kernel<<<blocks_for(n), 256, 0, c->stream>>>(data, log_n);
  for (uint32_t s = 0; s < log_n; s++)
    g1_butterfly_kernel<<<(n / 2 + 63) / 64, 64, 0, c->stream>>>(data, log_n, s, (const uint64_t*)tw);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "g1_fft launch failed");
}

// g_to_lagrange (arithmetic.rs:277-301): inverse FFT with omega^-1, every point times n^-1, normalised
int g1_to_lagrange(cq_ctx* c, const G1Affine* g, uint32_t k, G1Affine* out) {
  const uint32_t n = 1u << k;
  XYZZ* buf = nullptr;
  if (hipMalloc(&buf, (size_t)n * sizeof(XYZZ)) != hipSuccess) return c->fail(CQ_ERR_HIP, "hipMalloc(g_to_lagrange)");
  g1_from_affine_kernel<<<blocks_for(n), 256, 0, c->stream>>>(g, n, n, buf);
  Fr w = fr_from_raw(FR_ROOT_OF_UNITY_RAW);
  for (uint32_t i = k; i < FR_S; i++) w = w.sqr();
  int rc = g1_fft(c, buf, k, w.inv());
  if (rc == CQ_OK) {
    void* sc;
    if ((rc = c->ensure_scratch(2, 64, &sc)) == CQ_OK) {
      const Fr n_inv = Fr::from_u64(n).inv();
      fr_powers_canonical_kernel<<<1, 256, 0, c->stream>>>(Fr::one(), n_inv, 1, (uint64_t*)sc);
      g1_scale_kernel<<<(n + 63) / 64, 64, 0, c->stream>>>(buf, n, (const uint64_t*)sc, 0);
      g1_to_affine_kernel<<<blocks_for(n), 256, 0, c->stream>>>(buf, n, out);
      if (hipGetLastError() != hipSuccess) rc = c->fail(CQ_ERR_HIP, "g_to_lagrange launch failed");
    }
  }
  hipStreamSynchronize(c->stream);
  hipFree(buf);
  return rc;
}

// reversed, zero-padded coefficient vector of the FK convolution: d[k] = coeffs[N-1-k] for k < N, 0 above
__global__ void fk_reverse_kernel(const Fr* __restrict__ coeffs, uint32_t N, Fr* __restrict__ d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * N) return;
  const uint4* q = reinterpret_cast<const uint4*>(coeffs + (N - 1 - (i < N ? i : 0)));
  uint4 a = q[0], b = q[1];
  if (i >= N) a = b = make_uint4(0, 0, 0, 0);
  uint4* o = reinterpret_cast<uint4*>(d + i);
  o[0] = a;
  o[1] = b;
}
// h[m] = conv[N-2-m] for m <= N-2, identity for m = N-1
__global__ void fk_gather_kernel(const XYZZ* __restrict__ conv, uint32_t N, XYZZ* __restrict__ h) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= N) return;
  store_xyzz29(h + m, m + 1 < N ? load_xyzz29(conv + (N - 2 - m)) : XYZZ29::identity());
}

// Cached quotients Q_i = [ (T(X) - T(w^i)) / (X - w^i) * w^i / N ]_1 for every root, FK style:
//   h_m = sum_j c_{m+1+j} [s^j]  (a Toeplitz product = one cyclic convolution of size 2N over G1),  Q = DFT_N(h), then
//   the scaling by w^i / N (static_lookup.rs:111-117).  Everything a scalar could absorb is folded into one scale pass.
int fk_table_quotients(cq_ctx* c, const Fr* coeffs /* N, device */, const G1Affine* srs /* N, device */, uint32_t log_n,
                       G1Affine* qs_out /* N, device */) {
  const uint32_t N = 1u << log_n, N2 = 2 * N;
  XYZZ *S = nullptr, *H = nullptr;
  Fr* d = nullptr;
  uint64_t* canon = nullptr;
  auto cleanup = [&]() {
    hipStreamSynchronize(c->stream);
    for (void* p : {(void*)S, (void*)H, (void*)d, (void*)canon})
      if (p) hipFree(p);
  };
  if (hipMalloc(&S, (size_t)N2 * sizeof(XYZZ)) != hipSuccess || hipMalloc(&H, (size_t)N * sizeof(XYZZ)) != hipSuccess ||
      hipMalloc(&d, (size_t)N2 * sizeof(Fr)) != hipSuccess || hipMalloc(&canon, (size_t)N2 * 32) != hipSuccess) {
    cleanup();
    return c->fail(CQ_ERR_HIP, "hipMalloc(fk)");
  }
  Fr w2 = fr_from_raw(FR_ROOT_OF_UNITY_RAW);  // omega of the size-2N domain
  for (uint32_t i = log_n + 1; i < FR_S; i++) w2 = w2.sqr();
  const Fr w1 = w2.sqr();
  int rc;
  // FFT_2N of the powers [s^j] (zero-padded) over G1, and of the reversed coefficients over Fr
  g1_from_affine_kernel<<<blocks_for(N2), 256, 0, c->stream>>>(srs, N, N2, S);
  if ((rc = g1_fft(c, S, log_n + 1, w2)) != CQ_OK) { cleanup(); return rc; }
  fk_reverse_kernel<<<blocks_for(N2), 256, 0, c->stream>>>(coeffs, N, d);
  if ((rc = domain_fft(c, d, d, log_n + 1, w2, 1, N2, N2)) != CQ_OK) { cleanup(); return rc; }
  if ((rc = fr_to_canonical(c, d, N2, canon)) != CQ_OK) { cleanup(); return rc; }
  // pointwise product, inverse FFT (its 1/2N goes into the final scale), gather h, DFT_N
  g1_scale_kernel<<<(N2 + 63) / 64, 64, 0, c->stream>>>(S, N2, canon, 1);
  if ((rc = g1_fft(c, S, log_n + 1, w2.inv())) != CQ_OK) { cleanup(); return rc; }
  fk_gather_kernel<<<blocks_for(N), 256, 0, c->stream>>>(S, N, H);
  if ((rc = g1_fft(c, H, log_n, w1)) != CQ_OK) { cleanup(); return rc; }
  // Q_i * w^i / (N * 2N)
  const Fr scale0 = (Fr::from_u64(N) * Fr::from_u64(N2)).inv();
  fr_powers_canonical_kernel<<<blocks_for(N), 256, 0, c->stream>>>(w1, scale0, N, canon);
  g1_scale_kernel<<<(N + 63) / 64, 64, 0, c->stream>>>(H, N, canon, 1);
  g1_to_affine_kernel<<<blocks_for(N), 256, 0, c->stream>>>(H, N, qs_out);
  rc = hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fk launch failed");
  cleanup();
  return rc;
}

}  // namespace cq
