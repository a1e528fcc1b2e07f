#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../sha2_on_cq_halo2_amd/csrc/field.hpp"
using namespace cq;
__global__ void k(const Fr* a, const Fr* b, Fr* out, uint64_t* dbg) {
  Fr x = a[threadIdx.x], y = b[threadIdx.x];
  out[threadIdx.x] = x * y;
  // replicate the internals for lane 0
  if (threadIdx.x == 0) {
    constexpr uint32_t M29 = 0x1fffffffu;
    uint32_t A[9], B[9];
    Fr::unpack29(x.v.l, A); Fr::unpack29(y.v.l, B);
    for (int i = 0; i < 9; i++) { dbg[i] = A[i]; dbg[9 + i] = B[i]; }
    uint64_t c[18];
    for (int k2 = 0; k2 < 18; k2++) c[k2] = 0;
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)A[i] * B[j];
    for (int i = 0; i < 18; i++) dbg[18 + i] = c[i];
    constexpr uint32_t INV29 = FrP::INV & M29;
    for (int i = 0; i < 8; i++) {
      const uint32_t m = ((uint32_t)c[i] * INV29) & M29;
      dbg[60 + i] = m;
      for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * Fr::p29(j);
      c[i + 1] += c[i] >> 29;
    }
    { const uint32_t m = ((uint32_t)c[8] * INV29) & 0x00ffffffu; dbg[68] = m;
      for (int j = 0; j < 9; j++) c[8 + j] += (uint64_t)m * Fr::p29(j); }
    for (int i = 0; i < 18; i++) dbg[36 + i] = c[i];
    for (int j = 0; j < 9; j++) dbg[70 + j] = Fr::p29(j);
  }
}
int main() {
  Fr ha[64], hb[64], ho[64]; uint64_t hd[128];
  for (int i = 0; i < 64; i++) { ha[i] = Fr::one(); hb[i] = Fr::one(); }
  Fr *da, *db, *dout; uint64_t* dd;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho); hipMalloc(&dd, sizeof hd);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dout, dd);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  printf("out: "); for (int i = 7; i >= 0; i--) printf("%08x", ho[0].v.l[i]); printf("\n");
  Fr e = Fr::one() * Fr::one();
  printf("exp: "); for (int i = 7; i >= 0; i--) printf("%08x", e.v.l[i]); printf("\n");
  for (int i = 0; i < 79; i++) printf("dbg[%d]=%llx\n", i, (unsigned long long)hd[i]);
  return 0;
}
