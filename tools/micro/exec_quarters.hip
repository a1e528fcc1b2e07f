// Micro-benchmark (development aid): does a wave64 whose EXEC mask covers only 16 or 32 lanes issue its VALU instructions
// faster than a full one on gfx950?  One wave per workgroup, one workgroup per CU, a long chain of dependent-free
// v_mad_u64_u32 on four accumulators.    hipcc --offload-arch=gfx950 -O3 tools/micro/exec_quarters.hip -o /tmp/eq && /tmp/eq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void chain(uint64_t* out, uint32_t active, uint32_t iters, uint32_t seed) {
  const uint32_t lane = threadIdx.x;
  if (lane >= active) return;
  uint64_t a0 = seed + lane, a1 = seed * 3 + lane, a2 = seed * 5 + lane, a3 = seed * 7 + lane;
  uint32_t x = seed | 1, y = lane * 2654435761u + 12345u;
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      a0 += (uint64_t)x * y; asm("" : "+v"(a0));
      a1 += (uint64_t)y * (uint32_t)a0; asm("" : "+v"(a1));
      a2 += (uint64_t)x * (uint32_t)a1; asm("" : "+v"(a2));
      a3 += (uint64_t)y * (uint32_t)a2; asm("" : "+v"(a3));
    }
  }
  out[blockIdx.x * 64 + lane] = a0 ^ a1 ^ a2 ^ a3;
}
int main() {
  uint64_t* d;
  hipMalloc(&d, 256 * 64 * 8 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (uint32_t blocks : {256u, 1024u, 4096u})
    for (uint32_t active : {64u, 32u, 16u, 1u}) {
      chain<<<blocks, 64>>>(d, active, 100, 7);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      chain<<<blocks, 64>>>(d, active, 20000, 7);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4u active lanes %2u: %.3f ms  (%.2f ns per multiply-add per wave)\n", blocks, active, ms, ms * 1e6 / (20000.0 * 64));
    }
  return 0;
}
