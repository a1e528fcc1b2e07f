// EvaluationDomain (halo2_proofs/src/poly/domain.rs): constants per :39-142 computed on the host;
// the transforms of :238-374 map onto the fused first/last NTT passes of ntt.hip.
#include "ctx.hpp"
#include "poly.hpp"

namespace cq {

int domain_create(cq_ctx* c, uint32_t j, uint32_t k, cq_domain** out) {
  if (j < 2 && j != 1) return c->fail(CQ_ERR_ARG, "domain: j must be >= 1");
  if (k > FR_S) return c->fail(CQ_ERR_ARG, "domain: k too large");
  cq_domain* d = new cq_domain();
  d->ctx = c;
  d->j = j;
  d->k = k;
  d->quotient_poly_degree = j - 1;
  const uint64_t n = 1ull << k;
  uint32_t ek = k;
  while ((1ull << ek) < n * d->quotient_poly_degree) ek++;  // domain.rs:49-52
  if (ek > FR_S) {
    delete d;
    return c->fail(CQ_ERR_ARG, "domain: extended_k exceeds the field's 2-adicity");
  }
  d->extended_k = ek;
  Fr w = fr_from_raw(FR_ROOT_OF_UNITY_RAW);
  for (uint32_t i = ek; i < FR_S; i++) w = w.sqr();  // :59-61
  d->extended_omega = w;
  for (uint32_t i = k; i < ek; i++) w = w.sqr();  // :70-73
  d->omega = w;
  d->omega_inv = d->omega.inv();
  d->extended_omega_inv = d->extended_omega.inv();
  d->g_coset = fr_from_raw(FR_ZETA_RAW);       // :81
  d->g_coset_inv = d->g_coset.sqr();           // :82
  // :84-107 t(X) = X^n - 1 on the coset, period 2^(ek-k)
  const Fr orig = d->g_coset.pow_u64(n);
  const Fr step = d->extended_omega.pow_u64(n);
  Fr cur = orig;
  const size_t tl = (size_t)1 << (ek - k);
  for (size_t i = 0; i < tl; i++) {
    d->t_evaluations.push_back((cur - Fr::one()).inv());
    cur = cur * step;
  }
  d->ifft_divisor = Fr::from_u64(n).inv();
  d->extended_ifft_divisor = Fr::from_u64(1ull << ek).inv();
  d->barycentric_weight = d->ifft_divisor;
  if (hipMalloc(&d->t_evaluations_dev, tl * sizeof(Fr)) != hipSuccess ||
      hipMemcpy(d->t_evaluations_dev, d->t_evaluations.data(), tl * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) {
    delete d;
    return c->fail(CQ_ERR_HIP, "domain: t_evaluations upload failed");
  }
  *out = d;
  return CQ_OK;
}

void domain_destroy(cq_domain* d) {
  if (!d) return;
  hipStreamSynchronize(d->ctx->stream);
  if (d->t_evaluations_dev) hipFree(d->t_evaluations_dev);
  delete d;
}

static int run(cq_ctx* c, uint32_t log_n, const Fr& omega, const Fr* in, Fr* out, NttIo& io) {
  int rc = CQ_OK;
  const NttTables* tb = c->tables_for(log_n, omega, &rc);
  if (!tb) return rc;
  const size_t n = (size_t)1 << log_n;
  void* scr;
  if ((rc = c->ensure_scratch(c->ntt_scratch_slot, (size_t)2 * io.batch * n * sizeof(Fr), &scr)) != CQ_OK) return rc;
  io.prof = c;
  io.critical = c->ntt_scratch_slot == 0;  // not the side stream's slot (ctx.hpp): the caller waits for this transform
  if (ntt_run(*tb, in, out, (Fr*)scr, io, c->stream) != 0) return c->fail(CQ_ERR_HIP, "ntt launch failed");
  return CQ_OK;
}

int domain_fft(cq_ctx* c, const Fr* in, Fr* out, uint32_t log_n, const Fr& omega, uint32_t batch, size_t in_stride,
               size_t out_stride) {
  NttIo io;
  io.batch = batch;
  io.in_stride = in_stride;
  io.out_stride = out_stride;
  io.in_len = io.out_len = 1u << log_n;
  return run(c, log_n, omega, in, out, io);
}

// domain.rs:238-248: iNTT then * 1/n
int domain_lagrange_to_coeff(cq_domain* d, const Fr* in, Fr* out, uint32_t batch, size_t in_stride, size_t out_stride) {
  NttIo io;
  io.batch = batch;
  io.in_stride = in_stride;
  io.out_stride = out_stride;
  io.in_len = io.out_len = (uint32_t)d->n();
  io.out_mul = true;
  io.out_mul_v[0] = d->ifft_divisor;
  return run(d->ctx, d->k, d->omega_inv, in, out, io);
}

// domain.rs:252-266: * zeta^(i mod 3), zero-pad to the extended domain, NTT with extended_omega
int domain_coeff_to_extended(cq_domain* d, const Fr* in, Fr* out, uint32_t batch, size_t in_stride, size_t out_stride) {
  NttIo io;
  io.batch = batch;
  io.in_stride = in_stride;
  io.out_stride = out_stride;
  io.in_len = (uint32_t)d->n();
  io.out_len = (uint32_t)d->ext();
  io.in_coset = true;
  io.in_coset_mul[0] = d->g_coset;
  io.in_coset_mul[1] = d->g_coset_inv;
  return run(d->ctx, d->extended_k, d->extended_omega, in, out, io);
}

// domain.rs:293-315: iNTT(ext) * 1/ext, * zeta^-(i mod 3), truncate to n*(j-1)
int domain_extended_to_coeff(cq_domain* d, const Fr* in, Fr* out) {
  NttIo io;
  io.batch = 1;
  io.in_stride = d->ext();
  io.out_stride = d->n() * d->quotient_poly_degree;
  io.in_len = (uint32_t)d->ext();
  io.out_len = (uint32_t)(d->n() * d->quotient_poly_degree);
  io.out_mul = true;
  io.out_coset = true;
  io.out_mul_v[0] = d->extended_ifft_divisor;
  io.out_mul_v[1] = d->extended_ifft_divisor * d->g_coset_inv;  // coset_powers = [g_coset_inv, g_coset] (:351)
  io.out_mul_v[2] = d->extended_ifft_divisor * d->g_coset;
  return run(d->ctx, d->extended_k, d->extended_omega_inv, in, out, io);
}

}  // namespace cq
