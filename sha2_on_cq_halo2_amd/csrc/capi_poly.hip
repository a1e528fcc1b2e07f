// C ABI: EvaluationDomain transforms, eval_polynomial, kate_division, batch inversion.
#include <cstring>
#include "ctx.hpp"
#include "poly.hpp"

using namespace cq;

namespace {
struct HostStage {
  cq_ctx* c;
  void* din = nullptr;
  void* dout = nullptr;
  int rc = CQ_OK;
  HostStage(cq_ctx* c_, size_t in_bytes, size_t out_bytes) : c(c_) {
    if ((rc = c->ensure_scratch(1, in_bytes ? in_bytes : 32, &din)) != CQ_OK) return;
    rc = c->ensure_scratch(2, out_bytes ? out_bytes : 32, &dout);
  }
};
}  // namespace

extern "C" {

int cq_domain_create(cq_ctx* c, uint32_t j, uint32_t k, cq_domain** out) {
  if (!c || !out || j < 1) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return domain_create(c, j, k, out);
}
void cq_domain_destroy(cq_domain* d) { domain_destroy(d); }
uint32_t cq_domain_k(const cq_domain* d) { return d ? d->k : 0; }
uint32_t cq_domain_extended_k(const cq_domain* d) { return d ? d->extended_k : 0; }

int cq_domain_constants(const cq_domain* d, uint64_t omega[4], uint64_t omega_inv[4], uint64_t extended_omega[4],
                        uint64_t ifft_divisor[4]) {
  if (!d) return CQ_ERR_ARG;
  if (omega) d->omega.to_limbs64(omega);
  if (omega_inv) d->omega_inv.to_limbs64(omega_inv);
  if (extended_omega) d->extended_omega.to_limbs64(extended_omega);
  if (ifft_divisor) d->ifft_divisor.to_limbs64(ifft_divisor);
  return CQ_OK;
}

int cq_lagrange_to_coeff_dev(cq_domain* d, const uint64_t* in_dev, uint64_t* out_dev, uint32_t batch) {
  if (!d || !in_dev || !out_dev || !batch) return CQ_ERR_ARG;
  CQ_HIP(d->ctx, hipSetDevice(d->ctx->device));
  return domain_lagrange_to_coeff(d, (const Fr*)in_dev, (Fr*)out_dev, batch, d->n(), d->n());
}
int cq_coeff_to_extended_dev(cq_domain* d, const uint64_t* in_dev, uint64_t* out_dev, uint32_t batch) {
  if (!d || !in_dev || !out_dev || !batch) return CQ_ERR_ARG;
  if ((const void*)in_dev == (const void*)out_dev) return d->ctx->fail(CQ_ERR_ARG, "coeff_to_extended: in == out");
  CQ_HIP(d->ctx, hipSetDevice(d->ctx->device));
  return domain_coeff_to_extended(d, (const Fr*)in_dev, (Fr*)out_dev, batch, d->n(), d->ext());
}
int cq_extended_to_coeff_dev(cq_domain* d, const uint64_t* in_dev, uint64_t* out_dev) {
  if (!d || !in_dev || !out_dev) return CQ_ERR_ARG;
  CQ_HIP(d->ctx, hipSetDevice(d->ctx->device));
  return domain_extended_to_coeff(d, (const Fr*)in_dev, (Fr*)out_dev);
}

int cq_lagrange_to_coeff(cq_domain* d, uint64_t* a) {
  if (!d || !a) return CQ_ERR_ARG;
  cq_ctx* c = d->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t bytes = d->n() * sizeof(Fr);
  HostStage st(c, bytes, bytes);
  if (st.rc != CQ_OK) return st.rc;
  CQ_HIP(c, hipMemcpyAsync(st.din, a, bytes, hipMemcpyHostToDevice, c->stream));
  int rc = domain_lagrange_to_coeff(d, (const Fr*)st.din, (Fr*)st.dout, 1, d->n(), d->n());
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(a, st.dout, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}
int cq_coeff_to_extended(cq_domain* d, const uint64_t* a, uint64_t* out) {
  if (!d || !a || !out) return CQ_ERR_ARG;
  cq_ctx* c = d->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  HostStage st(c, d->n() * sizeof(Fr), d->ext() * sizeof(Fr));
  if (st.rc != CQ_OK) return st.rc;
  CQ_HIP(c, hipMemcpyAsync(st.din, a, d->n() * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  int rc = domain_coeff_to_extended(d, (const Fr*)st.din, (Fr*)st.dout, 1, d->n(), d->ext());
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(out, st.dout, d->ext() * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}
int cq_extended_to_coeff(cq_domain* d, const uint64_t* a, uint64_t* out) {
  if (!d || !a || !out) return CQ_ERR_ARG;
  cq_ctx* c = d->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t out_bytes = d->n() * d->quotient_poly_degree * sizeof(Fr);
  HostStage st(c, d->ext() * sizeof(Fr), out_bytes);
  if (st.rc != CQ_OK) return st.rc;
  CQ_HIP(c, hipMemcpyAsync(st.din, a, d->ext() * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  int rc = domain_extended_to_coeff(d, (const Fr*)st.din, (Fr*)st.dout);
  if (rc != CQ_OK) return rc;
  if (out_bytes) CQ_HIP(c, hipMemcpyAsync(out, st.dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

int cq_eval_polynomial_dev(cq_ctx* c, const uint64_t* poly_dev, size_t n, const uint64_t point[4], uint64_t out[4]) {
  if (!c || !point || !out || (n && !poly_dev) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  Fr r;
  int rc = poly_eval(c, (const Fr*)poly_dev, (uint32_t)n, Fr::from_limbs64(point), &r);
  if (rc != CQ_OK) return rc;
  r.to_limbs64(out);
  return CQ_OK;
}
int cq_eval_polynomial(cq_ctx* c, const uint64_t* poly, size_t n, const uint64_t point[4], uint64_t out[4]) {
  if (!c || !point || !out || (n && !poly) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  HostStage st(c, n * sizeof(Fr), 32);
  if (st.rc != CQ_OK) return st.rc;
  if (n) CQ_HIP(c, hipMemcpyAsync(st.din, poly, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  return cq_eval_polynomial_dev(c, (const uint64_t*)st.din, n, point, out);
}

int cq_kate_division_dev(cq_ctx* c, const uint64_t* a_dev, size_t n, const uint64_t b[4], uint64_t* q_dev) {
  if (!c || !a_dev || !b || n == 0 || (n > 1 && !q_dev) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return poly_kate_division(c, (const Fr*)a_dev, (uint32_t)n, Fr::from_limbs64(b), (Fr*)q_dev);
}
int cq_kate_division(cq_ctx* c, const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* q) {
  if (!c || !a || !b || n == 0 || (n > 1 && !q) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  HostStage st(c, n * sizeof(Fr), n * sizeof(Fr));
  if (st.rc != CQ_OK) return st.rc;
  CQ_HIP(c, hipMemcpyAsync(st.din, a, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  int rc = poly_kate_division(c, (const Fr*)st.din, (uint32_t)n, Fr::from_limbs64(b), (Fr*)st.dout);
  if (rc != CQ_OK) return rc;
  if (n > 1) CQ_HIP(c, hipMemcpyAsync(q, st.dout, (n - 1) * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

int cq_batch_invert_dev(cq_ctx* c, uint64_t* a_dev, size_t n) {
  if (!c || (n && !a_dev) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return poly_batch_invert(c, (Fr*)a_dev, (uint32_t)n);
}
int cq_batch_invert(cq_ctx* c, uint64_t* a, size_t n) {
  if (!c || (n && !a) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  if (!n) return CQ_OK;
  HostStage st(c, n * sizeof(Fr), 32);
  if (st.rc != CQ_OK) return st.rc;
  CQ_HIP(c, hipMemcpyAsync(st.din, a, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  int rc = poly_batch_invert(c, (Fr*)st.din, (uint32_t)n);
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(a, st.din, n * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

}  // extern "C"
