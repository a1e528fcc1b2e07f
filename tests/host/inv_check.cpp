// HIP-free check of field.hpp's inversions against each other: the variable-time safegcd (the batch inversion's lone lane)
// must return the words of the constant-time one and of Fermat's x^(p-2), for both fields, on random elements, small values,
// values next to p, powers of two; and x * inv(x) = 1.  Built and run by tests/test_plonk_host_cpu.py.
#include <cstdio>
#include <cstdint>
#include "field.hpp"

using namespace cq;

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint64_t next64() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
template <class F>
static int run(const char* name) {
  int bad = 0, n = 0;
  auto check = [&](const F& x) {
    n++;
    const F a = x.inv_safegcd(), b = x.inv_safegcd_var();
    if (!(a == b)) bad++;
    if (!x.is_zero() && !((x * b) == F::one())) bad++;
    if (x.is_zero() && !b.is_zero()) bad++;
  };
  check(F::zero());
  check(F::one());
  check(F::one().neg());
  for (uint64_t v = 2; v < 200; v++) {
    check(F::from_u64(v));
    check(F::from_u64(v).neg());
  }
  for (int sh = 0; sh < 254; sh++) {  // 2^sh and 2^sh - 1 as field elements
    F p2 = F::one();
    for (int i = 0; i < sh; i++) p2 = p2.dbl();
    check(p2);
    check(p2 - F::one());
  }
  for (int i = 0; i < 20000; i++) {
    uint64_t w[4] = {next64(), next64(), next64(), next64() >> 3};
    F x = F::from_limbs64(w);          // some 253-bit pattern read as a Montgomery value (always < p: p > 2^253)
    check(x);
  }
  {  // and against Fermat on a few
    F x = F::from_u64(0x123456789abcdefull);
    for (int i = 0; i < 50; i++) {
      x = x * x + F::from_u64(i + 3);
      if (!(x.inv_safegcd_var() == x.inv_fermat())) bad++;
    }
  }
  printf("%s: %d elements, %d disagreements\n", name, n, bad);
  return bad;
}
int main() {
  int bad = run<Fr>("Fr") + run<Fq>("Fq");
  printf(bad ? "FAIL\n" : "ok\n");
  return bad ? 1 : 0;
}
