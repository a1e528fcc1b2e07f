// quad_add (curve29.hpp: one general addition by four lanes) against xyzz29_add on one wave, every special case:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sha2_on_cq_halo2_amd/csrc tools/micro/quad_add_test.hip -o /tmp/qt && /tmp/qt
// (tests/test_msm_gpu.py::test_quad_add_matches_lane_serial_addition builds and runs it on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve29.hpp"
using namespace cq;

// mode 0: general; odd quads: 1 P2 = identity, 2 P1 = identity, 3 P2 = P1 (doubling), 4 P2 = -P1 (cancellation), 5 both identity
__global__ __launch_bounds__(64) void test(uint32_t* out, uint32_t mode) {
  const uint32_t lane = threadIdx.x, quad = lane >> 2, role = lane & 3;
  Affine29 g;  // the generator (1, 2) in R' form
  g.x = Fq29::one();
  g.y = Fq29::one() + Fq29::one();
  g.y.normalise();
  auto mulk = [&](uint32_t k) {
    XYZZ29 a = XYZZ29::identity();
    for (uint32_t i = 0; i < k; i++) xyzz29_add_affine(a, g);
    return a;
  };
  XYZZ29 p1 = mulk(2 + quad), p2 = mulk(40 + 3 * quad);
  if (quad & 1) {
    if (mode == 1 || mode == 5) p2 = XYZZ29::identity();
    if (mode == 2 || mode == 5) p1 = XYZZ29::identity();
    if (mode == 3) p2 = mulk(2 + quad);  // the same point (and the same coordinates)
    if (mode == 4) {
      p2 = p1;
      p2.y = Fq29::neg<4>(p2.y);
    }
  }
  XYZZ29 ref = p1;
  xyzz29_add(ref, p2);
  const Fq29 F = role == 0 ? p1.x : role == 1 ? p1.y : role == 2 ? p1.zz : p1.zzz;
  const Fq29 G = role == 0 ? p2.x : role == 1 ? p2.y : role == 2 ? p2.zz : p2.zzz;
  const Fq29 R = quad_add(F, G);
  const XYZZ29 q = {quad_perm<0, 0, 0, 0>(R), quad_perm<1, 1, 1, 1>(R), quad_perm<2, 2, 2, 2>(R), quad_perm<3, 3, 3, 3>(R)};
  // the same affine point?  x_ref zz_q == x_q zz_ref and y_ref zzz_q == y_q zzz_ref, and identity <=> identity
  const Fq a = (ref.x * q.zz).to_mont256(), b = (q.x * ref.zz).to_mont256();
  const Fq c = (ref.y * q.zzz).to_mont256(), d = (q.y * ref.zzz).to_mont256();
  out[lane] = (a == b) && (c == d) && (ref.is_identity() == q.is_identity());
}
int main() {
  uint32_t* d;
  if (hipMalloc(&d, 64 * 4) != hipSuccess) return 2;
  int total = 0;
  for (uint32_t mode = 0; mode < 6; mode++) {
    test<<<1, 64>>>(d, mode);
    uint32_t h[64];
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int i = 0; i < 64; i++) bad += !h[i];
    printf("mode %u: %d of 64 lanes disagree\n", mode, bad);
    total += bad;
  }
  printf(total ? "FAILED\n" : "ok\n");
  return total ? 1 : 0;
}
