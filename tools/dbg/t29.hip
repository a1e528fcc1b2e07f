// Standalone check + microbenchmark of the lazy 9x29-bit field (field29.hpp) against the host Fq arithmetic.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/t29 tools/dbg/t29.hip && /tmp/t29
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "../../sha2_on_cq_halo2_amd/csrc/field29.hpp"
using namespace cq;

__global__ void chain_kernel(Fq* io, uint32_t iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = Fq29::from_mont256(io[2 * tid]), b = Fq29::from_mont256(io[2 * tid + 1]);
  for (uint32_t i = 0; i < iters; i++) {
    a = a * b;
    b = b * a;
  }
  io[2 * tid] = a.to_mont256();
  io[2 * tid + 1] = b.to_mont256();
}
// exercises add / sub / neg with the K bounds used by the curve code: out0 = (a+b)*(a-b) , out1 = (8p - (a*a - b - 2a)) * b
__global__ void ops_kernel(Fq* io) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = Fq29::from_mont256(io[2 * tid]), b = Fq29::from_mont256(io[2 * tid + 1]);
  Fq29 s = a + b;                    // < 4p, limbs < 2^30
  Fq29 d = Fq29::sub<2>(a, b);       // < 4p
  Fq29 o0 = s * d;
  Fq29 t = Fq29::sub<2>(a.sqr(), b); // < 4p
  Fq29 u = Fq29::sub<4>(t, a + a);   // < 8p
  Fq29 w = Fq29::neg<8>(u);          // < 8p
  Fq29 o1 = w * b;
  io[2 * tid] = o0.to_mont256();
  io[2 * tid + 1] = o1.to_mont256();
}

static uint64_t rs = 88172645463325252ull;
static uint64_t xr() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

int main() {
  const int lanes = 256 * 64;
  std::vector<Fq> h(2 * lanes), h0;
  for (auto& v : h) { uint64_t w[8]; for (auto& x : w) x = xr(); v = Fq::from_u512(w); }
  // edge values
  h[0] = Fq::zero(); h[1] = Fq::one(); h[2] = Fq::zero() - Fq::one(); h[3] = Fq::zero() - Fq::one(); h[4] = Fq::one(); h[5] = Fq::zero();
  h0 = h;
  Fq* d;
  hipMalloc(&d, h.size() * sizeof(Fq));
  hipMemcpy(d, h.data(), h.size() * sizeof(Fq), hipMemcpyHostToDevice);
  chain_kernel<<<lanes / 256, 256>>>(d, 5);
  hipMemcpy(h.data(), d, h.size() * sizeof(Fq), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < lanes; i++) {
    Fq a = h0[2 * i], b = h0[2 * i + 1];
    for (int k = 0; k < 5; k++) { a = a * b; b = b * a; }
    if (!(a == h[2 * i]) || !(b == h[2 * i + 1])) bad++;
  }
  printf("chain mismatches: %d / %d\n", bad, lanes);
  hipMemcpy(d, h0.data(), h.size() * sizeof(Fq), hipMemcpyHostToDevice);
  ops_kernel<<<lanes / 256, 256>>>(d);
  hipMemcpy(h.data(), d, h.size() * sizeof(Fq), hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < lanes; i++) {
    Fq a = h0[2 * i], b = h0[2 * i + 1];
    Fq o0 = (a + b) * (a - b);
    Fq o1 = (Fq::zero() - (a * a - b - (a + a))) * b;
    if (!(o0 == h[2 * i]) || !(o1 == h[2 * i + 1])) bad++;
  }
  printf("ops mismatches: %d / %d\n", bad, lanes);
  // throughput
  const int L2 = 256 * 256 * 8;
  Fq* big;
  hipMalloc(&big, (size_t)2 * L2 * sizeof(Fq));
  hipMemset(big, 1, (size_t)2 * L2 * sizeof(Fq));
  chain_kernel<<<L2 / 256, 256>>>(big, 16);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  chain_kernel<<<L2 / 256, 256>>>(big, 2048);
  hipDeviceSynchronize();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("Fq29 mul: %.1f Gmul/s\n", (double)L2 * 4096 / dt / 1e9);
  return bad != 0;
}
