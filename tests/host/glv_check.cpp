// HIP-free check of csrc/glv.hpp (the host's endomorphism-split scalar multiplication used for the f commitments) against
// plain double-and-add, on random scalars, the edge scalars and a random point; built and run by tests/test_plonk_host_cpu.py.
#include <cstdio>
#include <cstdint>
#include "glv.hpp"

using namespace cq;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next64() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static G1Jac plain_mul(const G1Jac& p, const Fr& k) {
  const U256 e = k.to_canonical();
  G1Jac acc = G1Jac::identity();
  for (int w = 7; w >= 0; w--)
    for (int bit = 31; bit >= 0; bit--) {
      acc = jac_dbl(acc);
      if ((e.l[w] >> bit) & 1u) acc = jac_add(acc, p);
    }
  return acc;
}
static bool same_point(const G1Jac& a, const G1Jac& b) {
  const G1Affine x = jac_to_affine(a), y = jac_to_affine(b);
  return x.x == y.x && x.y == y.y;
}
int main() {
  G1Affine g;
  g.x = Fq::from_u64(1);
  g.y = Fq::from_u64(2);
  G1Jac p = jac_from_affine(g);
  for (int i = 0; i < 5; i++) p = jac_add(jac_dbl(p), jac_from_affine(g));  // some other point
  int bad = 0;
  auto check = [&](const Fr& k) {
    const G1Jac ref = plain_mul(p, k);
    if (!same_point(host_scalar_mul(p, k), ref)) bad++;
    if (!same_point(jac_add(host_scalar_mul_half(p, k, 0), host_scalar_mul_half(p, k, 1)), ref)) bad++;
  };
  check(Fr::zero());
  check(Fr::one());
  check(Fr::one().neg());
  check(Fr::from_u64(2));
  check(fr_from_raw(GLV_LAMBDA_RAW));
  check(fr_from_raw(GLV_LAMBDA_RAW).neg());
  check(fr_from_raw(FR_HALF_RAW));
  check(fr_from_raw(FR_HALF_RAW) + Fr::one());
  for (int i = 0; i < 300; i++) {
    uint64_t w[8];
    for (int j = 0; j < 8; j++) w[j] = next64();
    check(Fr::from_u512(w));
  }
  // the split itself: phi(P) = lambda P
  G1Jac phi = p;
  phi.x = phi.x * (Fq::from_limbs64(GLV_ZETA_RAW) * Fq::r2());
  if (!same_point(phi, plain_mul(p, fr_from_raw(GLV_LAMBDA_RAW)))) bad++;
  printf(bad ? "FAILED %d\n" : "ok\n", bad);
  return bad ? 1 : 0;
}
