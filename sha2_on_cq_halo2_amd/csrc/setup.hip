// Test/bench SRS construction on the GPU: `ParamsKZG::setup_from_toxic_waste`
// (halo2_proofs/src/poly/kzg/commitment.rs:209-276) and the G1 part of
// `TableSRS::setup_from_toxic_waste` (:73-178).  The reference marks both "FOR TESTING
// PURPOSES" / "MUST NOT be used in production"; they exist here so that benches and parity tests
// can run on a true KZG SRS ([s^i]_1 and [L_i(s)]_1) at k = 18..22 without a host round trip.
//
// Every SRS element is (some scalar) * G, so the work is a fixed-base scalar multiplication:
// a host-built table T[j][d] = d * 2^(8j) * G (32 x 256 affine points, 512 KiB, L2 resident) turns
// one multiplication into 32 mixed additions; the scalars themselves (s^i, the closed-form
// Lagrange scalars of :241-251, the opening-at-zero scalars of :156-170) come from small kernels.
#include <vector>
#include "ctx.hpp"
#include "curve.hpp"
#include "setup.hpp"

namespace cq {

__global__ void fr_powers2_kernel(Fr* out, Fr base, uint32_t count) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  out[j] = base.pow_u64(j);
}

// out[i] = multiplier * w^i / (s - w^i)      (kzg/commitment.rs:243-250)
__global__ void lagrange_scalars_kernel(Fr* out, Fr s, Fr omega, Fr multiplier, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr rp = omega.pow_u64(i);
  out[i] = multiplier * rp * (s - rp).inv();
}

// out[i] = w^-i * L_i(s) - s^(N-1)/N          (kzg/commitment.rs:156-170 in scalar form)
__global__ void opening_at_zero_scalars_kernel(Fr* out, const Fr* lagrange, Fr omega_inv, Fr last_scaled, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = omega_inv.pow_u64(i) * lagrange[i] - last_scaled;
}

__global__ __launch_bounds__(128) void fixed_base_mul_kernel(const Fr* __restrict__ scalars, uint32_t n,
                                                             const G1Affine* __restrict__ table, G1Affine* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U256 v = scalars[i].to_canonical();
  XYZZ acc = XYZZ::identity();
  for (uint32_t j = 0; j < 32; j++) {
    const uint32_t byte = (v.l[j >> 2] >> ((j & 3) * 8)) & 0xffu;
    if (byte) xyzz_add_affine(acc, table[j * 256 + byte]);
  }
  G1Affine r = G1Affine::identity();
  if (!acc.is_identity()) {
    Fq iv = (acc.zz * acc.zzz).inv();
    r.x = acc.x * (iv * acc.zzz);
    r.y = acc.y * (iv * acc.zz);
  }
  out[i] = r;
}

static const G1Affine* fixed_base_table(cq_ctx* c, int* rc) {
  if (c->fb_table) return (const G1Affine*)c->fb_table;
  // host: T[j][d] = d * 2^(8j) * G, normalised with one inversion (Montgomery's trick)
  std::vector<G1Jac> jac(32 * 256);
  G1Affine gen = {Fq::one(), Fq::from_u64(2)};  // bn256/curve.rs:66-67
  G1Jac base = jac_from_affine(gen);
  for (int j = 0; j < 32; j++) {
    jac[j * 256] = G1Jac::identity();
    G1Jac cur = base;
    for (int d = 1; d < 256; d++) {
      jac[j * 256 + d] = cur;
      cur = jac_add(cur, base);
    }
    for (int k = 0; k < 8; k++) base = jac_dbl(base);
  }
  std::vector<G1Affine> aff(32 * 256);
  std::vector<Fq> pref(32 * 256);
  Fq acc = Fq::one();
  for (size_t i = 0; i < jac.size(); i++) {
    pref[i] = acc;
    if (!jac[i].is_identity()) acc = acc * jac[i].z;
  }
  acc = acc.inv();
  for (size_t i = jac.size(); i-- > 0;) {
    if (jac[i].is_identity()) {
      aff[i] = G1Affine::identity();
      continue;
    }
    Fq zi = pref[i] * acc;
    acc = acc * jac[i].z;
    Fq zi2 = zi.sqr();
    aff[i] = {jac[i].x * zi2, jac[i].y * zi2 * zi};
  }
  void* d = nullptr;
  if (hipMalloc(&d, aff.size() * sizeof(G1Affine)) != hipSuccess) {
    *rc = c->fail(CQ_ERR_HIP, "hipMalloc(fixed-base table)");
    return nullptr;
  }
  if (hipMemcpy(d, aff.data(), aff.size() * sizeof(G1Affine), hipMemcpyHostToDevice) != hipSuccess) {
    *rc = c->fail(CQ_ERR_HIP, "hipMemcpy(fixed-base table)");
    return nullptr;
  }
  c->fb_table = d;
  return (const G1Affine*)d;
}

int fixed_base_mul(cq_ctx* c, const Fr* scalars_dev, uint32_t n, G1Affine* out_dev) {
  int rc = CQ_OK;
  const G1Affine* tb = fixed_base_table(c, &rc);
  if (!tb) return rc;
  if (n) fixed_base_mul_kernel<<<(n + 127) / 128, 128, 0, c->stream>>>(scalars_dev, n, tb, out_dev);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fixed_base_mul launch failed");
}

Fr domain_root(uint32_t k) {
  // kzg/commitment.rs:237-240: ROOT_OF_UNITY_INV^-1 squared (S-k) times == ROOT_OF_UNITY^(2^(S-k))
  Fr root = fr_from_raw(FR_ROOT_OF_UNITY_RAW);
  for (uint32_t i = k; i < FR_S; i++) root = root.sqr();
  return root;
}

// g[i] = [s^i]_1, g_lagrange[i] = [L_i(s)]_1 for the 2^k domain; scratch: n Fr
int srs_powers_and_lagrange(cq_ctx* c, uint32_t k, const Fr& s, G1Affine* g_dev, G1Affine* g_lagrange_dev,
                            Fr* scalars_tmp_dev, Fr* lagrange_scalars_keep_dev) {
  const uint32_t n = 1u << k;
  const uint32_t blocks = (n + 255) / 256;
  int rc;
  if (g_dev) {
    fr_powers2_kernel<<<blocks, 256, 0, c->stream>>>(scalars_tmp_dev, s, n);
    if ((rc = fixed_base_mul(c, scalars_tmp_dev, n, g_dev)) != CQ_OK) return rc;
  }
  if (g_lagrange_dev) {
    const Fr root = domain_root(k);
    const Fr n_inv = Fr::from_u64(n).inv();
    const Fr multiplier = (s.pow_u64(n) - Fr::one()) * n_inv;
    Fr* dst = lagrange_scalars_keep_dev ? lagrange_scalars_keep_dev : scalars_tmp_dev;
    lagrange_scalars_kernel<<<blocks, 256, 0, c->stream>>>(dst, s, root, multiplier, n);
    if ((rc = fixed_base_mul(c, dst, n, g_lagrange_dev)) != CQ_OK) return rc;
  }
  return CQ_OK;
}

int srs_opening_at_zero(cq_ctx* c, uint32_t k, const Fr& s, const Fr* lagrange_scalars_dev, Fr* scalars_tmp_dev,
                        G1Affine* out_dev) {
  const uint32_t n = 1u << k;
  const Fr root = domain_root(k);
  const Fr n_inv = Fr::from_u64(n).inv();
  const Fr last_scaled = s.pow_u64(n - 1) * n_inv;
  opening_at_zero_scalars_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(scalars_tmp_dev, lagrange_scalars_dev, root.inv(),
                                                                        last_scaled, n);
  return fixed_base_mul(c, scalars_tmp_dev, n, out_dev);
}

}  // namespace cq
