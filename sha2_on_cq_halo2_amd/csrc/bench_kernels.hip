// Measurement support: chip-level 256-bit Montgomery multiplication rate.
#include "ctx.hpp"
#include "field29.hpp"
using namespace cq;

template <class F>
__global__ void modmul_bench_kernel(F* out, uint32_t iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::from_u64(tid + 3), b = F::from_u64(2 * tid + 7);
  for (uint32_t i = 0; i < iters; i++) {
    a = a * b;
    b = b * a;
  }
  out[tid] = a + b;
}

// the lazy 9 x 29-bit field the MSM kernels compute in (field29.hpp)
__global__ void modmul29_bench_kernel(Fq* out, uint32_t iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = Fq29::from_mont256(Fq::from_u64(tid + 3)), b = Fq29::from_mont256(Fq::from_u64(2 * tid + 7));
  for (uint32_t i = 0; i < iters; i++) {
    a = a * b;
    b = b * a;
  }
  out[tid] = (a + b).reduced().to_mont256();
}

extern "C" {

int cq_bench_modmul_dev(cq_ctx* c, uint64_t* out_dev, uint32_t lanes, uint32_t iters, int which) {
  if (!c || !out_dev || lanes % 256) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  if (which == 0)
    modmul_bench_kernel<Fr><<<lanes / 256, 256, 0, c->stream>>>((Fr*)out_dev, iters / 2);
  else if (which == 2)
    modmul29_bench_kernel<<<lanes / 256, 256, 0, c->stream>>>((Fq*)out_dev, iters / 2);
  else
    modmul_bench_kernel<Fq><<<lanes / 256, 256, 0, c->stream>>>((Fq*)out_dev, iters / 2);
  CQ_HIP(c, hipGetLastError());
  return CQ_OK;
}


}  // extern "C"
