// Standalone check of curve29.hpp against the host group law (curve.hpp compiled for the host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../sha2_on_cq_halo2_amd/csrc/curve29.hpp"
using namespace cq;

// out[0] = A + B (affine adds into identity), out[1] = (A + B) + (A + B) via xyzz add (doubling branch),
// out[2] = A + A via affine add (doubling branch), out[3] = A + (-A), out[4] = ((A+B) + C) + (A + B) general add
__global__ void k(const G1Affine* pts, G1Jac* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine29 A = load_affine29(pts + 3 * i, true), B = load_affine29(pts + 3 * i + 1, true), C = load_affine29(pts + 3 * i + 2, true);
  XYZZ29 s = XYZZ29::identity();
  xyzz29_add_affine(s, A);
  xyzz29_add_affine(s, B);
  out[5 * i] = xyzz29_to_jac(s);
  XYZZ29 d = s;
  xyzz29_add(d, s);
  out[5 * i + 1] = xyzz29_to_jac(d);
  XYZZ29 e = XYZZ29::identity();
  xyzz29_add_affine(e, A);
  xyzz29_add_affine(e, A);
  out[5 * i + 2] = xyzz29_to_jac(e);
  XYZZ29 f = XYZZ29::identity();
  xyzz29_add_affine(f, A);
  Affine29 nA = A;
  nA.y = Fq29::neg<2>(nA.y);
  xyzz29_add_affine(f, nA);
  out[5 * i + 3] = xyzz29_to_jac(f);
  XYZZ29 g = s;
  xyzz29_add_affine(g, C);
  xyzz29_add(g, s);
  out[5 * i + 4] = xyzz29_to_jac(g);
}

static uint64_t rs = 88172645463325252ull;
static uint64_t xr() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

int main() {
  const int n = 512;
  G1Affine gen = {Fq::from_u64(1), Fq::from_u64(2)};
  std::vector<G1Affine> pts(3 * n);
  G1Jac cur = jac_from_affine(gen);
  for (int i = 0; i < 3 * n; i++) {
    int steps = 1 + (xr() % 5);
    for (int s = 0; s < steps; s++) cur = jac_add(jac_dbl(cur), jac_from_affine(gen));
    pts[i] = jac_to_affine(cur);
  }
  G1Affine* dp; G1Jac* dout;
  hipMalloc(&dp, pts.size() * sizeof(G1Affine));
  hipMalloc(&dout, 5 * n * sizeof(G1Jac));
  hipMemcpy(dp, pts.data(), pts.size() * sizeof(G1Affine), hipMemcpyHostToDevice);
  k<<<(n + 63) / 64, 64>>>(dp, dout, n);
  std::vector<G1Jac> out(5 * n);
  hipMemcpy(out.data(), dout, out.size() * sizeof(G1Jac), hipMemcpyDeviceToHost);
  int bad[5] = {0, 0, 0, 0, 0};
  auto eq = [](const G1Jac& a, const G1Jac& b) {
    G1Affine x = jac_to_affine(a), y = jac_to_affine(b);
    return x.x == y.x && x.y == y.y;
  };
  for (int i = 0; i < n; i++) {
    G1Jac A = jac_from_affine(pts[3 * i]), B = jac_from_affine(pts[3 * i + 1]), C = jac_from_affine(pts[3 * i + 2]);
    G1Jac s = jac_add(A, B);
    if (!eq(out[5 * i], s)) bad[0]++;
    if (!eq(out[5 * i + 1], jac_dbl(s))) bad[1]++;
    if (!eq(out[5 * i + 2], jac_dbl(A))) bad[2]++;
    if (!out[5 * i + 3].is_identity()) bad[3]++;
    if (!eq(out[5 * i + 4], jac_add(jac_add(s, C), s))) bad[4]++;
  }
  printf("mismatches: add_affine %d, xyzz dbl-branch %d, affine dbl-branch %d, cancel %d, general %d (of %d)\n", bad[0], bad[1], bad[2], bad[3], bad[4], n);
  return 0;
}
