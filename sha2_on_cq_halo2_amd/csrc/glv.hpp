// Host-side scalar multiplication of a G1 point with the curve's endomorphism (HIP-free: also built by
// tests/test_sanitizers_cpu.py's host unit).
#pragma once
#include "curve.hpp"

namespace cq {

// k * P on the host: used where a commitment is a known linear combination of commitments already computed --
// f = sum_j theta^(w-1-j) e_j over plain advice columns gives [f] = sum_j theta^(w-1-j) [e_j], the same group element as
// the n-term MSM the reference runs.  The four of a proof sit on its critical path (theta -> beta), so the scalar is split
// with the curve's endomorphism (bn256/curve.rs:69-83, 137-166: phi(x, y) = (zeta x, y) = lambda (x, y)):
// k = k1 + k2 lambda with |k1|, |k2| < 2^127, and k1 P + k2 phi(P) runs as ONE 4-bit-window chain of 128 doublings instead
// of 256 (Straus).  The lattice basis (a1, b1), (a2, b2) -- a_i + b_i lambda = 0 mod r, a1 b2 - a2 b1 = r -- is the one the
// extended Euclid on (r, lambda) stops at; c_i = floor(g_i k / 2^256) with g1 = floor(2^256 b2 / r), g2 = floor(-2^256 b1 / r)
// approximates the rounded quotients, and k2 = -(c1 b1 + c2 b2), k1 = k - k2 lambda are then EXACT in the field whatever
// the rounding (checked over 20 000 scalars: both below 2^127 in magnitude).
static const uint64_t GLV_LAMBDA_RAW[4] = {0x8b17ea66b99c90ddull, 0x5bfc41088d8daaa7ull, 0xb3c4d79d41a91758ull, 0x0ull};
static const uint64_t GLV_ZETA_RAW[4] = {0x5763473177fffffeull, 0xd4f263f1acdb5c4full, 0x59e26bcea0d48bacull, 0x0ull};  // fq.rs ZETA
static const uint64_t GLV_G1[3] = {0xd91d232ec7e0b3d7ull, 0x2ull, 0x0ull};
static const uint64_t GLV_G2[3] = {0x7a7bd9d4391eb18dull, 0x4ccef014a773d2cfull, 0x2ull};
static const uint64_t GLV_MINUS_B1[4] = {0x8211bbeb7d4f1128ull, 0x6f4d8248eeb859fcull, 0x0ull, 0x0ull};
static const uint64_t GLV_B2[4] = {0x89d3256894d213e3ull, 0x0ull, 0x0ull, 0x0ull};

// floor(g k / 2^256) for a three-word g and a four-word k, as a raw four-word integer
static inline void glv_mul_hi(const uint64_t* g, const uint64_t* k, uint64_t* out) {
  typedef unsigned __int128 u128;
  uint64_t prod[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 3; i++) {
    u128 carry = 0;
    for (int j = 0; j < 4; j++) {
      const u128 t = (u128)g[i] * k[j] + prod[i + j] + carry;
      prod[i + j] = (uint64_t)t;
      carry = t >> 64;
    }
    prod[i + 4] = (uint64_t)carry;
  }
  for (int i = 0; i < 4; i++) out[i] = i + 4 < 7 ? prod[i + 4] : 0;
}
// |v| as two words and its sign, for a field element that stands for a small signed integer
static inline void glv_small(const Fr& v, uint64_t* mag, bool& neg) {
  static const Fr half = fr_from_raw(FR_HALF_RAW);
  uint64_t w[4];
  const U256 c = v.to_canonical();
  for (int i = 0; i < 4; i++) w[i] = (uint64_t)c.l[2 * i] | ((uint64_t)c.l[2 * i + 1] << 32);
  uint64_t hw[4];
  const U256 hc = half.to_canonical();
  for (int i = 0; i < 4; i++) hw[i] = (uint64_t)hc.l[2 * i] | ((uint64_t)hc.l[2 * i + 1] << 32);
  neg = false;
  for (int i = 3; i >= 0; i--)
    if (w[i] != hw[i]) {
      neg = w[i] > hw[i];
      break;
    }
  if (neg) {
    const U256 n = v.neg().to_canonical();
    for (int i = 0; i < 4; i++) w[i] = (uint64_t)n.l[2 * i] | ((uint64_t)n.l[2 * i + 1] << 32);
  }
  mag[0] = w[0];
  mag[1] = w[1];
}
// the split of k and, per half, the 4-bit-window chain over 128 bits: half 0 = k1 P, half 1 = k2 phi(P); their sum is k P.
// (Two halves so that two worker threads can share one multiplication: host_scalar_mul_half.)
struct GlvSplit {
  uint64_t m[2][2];
  bool neg[2];
};
inline GlvSplit glv_split(const Fr& k) {
  static const Fr lambda = fr_from_raw(GLV_LAMBDA_RAW), minus_b1 = fr_from_raw(GLV_MINUS_B1), b2 = fr_from_raw(GLV_B2);
  uint64_t kw[4], c1w[4], c2w[4];
  const U256 kc = k.to_canonical();
  for (int i = 0; i < 4; i++) kw[i] = (uint64_t)kc.l[2 * i] | ((uint64_t)kc.l[2 * i + 1] << 32);
  glv_mul_hi(GLV_G1, kw, c1w);
  glv_mul_hi(GLV_G2, kw, c2w);
  const Fr c1 = fr_from_raw(c1w), c2 = fr_from_raw(c2w);
  const Fr k2 = c1 * minus_b1 - c2 * b2;  // -(c1 b1 + c2 b2)
  const Fr k1 = k - k2 * lambda;
  GlvSplit sp;
  glv_small(k1, sp.m[0], sp.neg[0]);
  glv_small(k2, sp.m[1], sp.neg[1]);
  return sp;
}
// 1..15 times P (half 0) or phi(P) (half 1: x scaled by zeta), the sign of the half's scalar folded in
inline void glv_table(const G1Jac& p, const GlvSplit& sp, int half, G1Jac* t) {
  static const Fq zeta = Fq::from_limbs64(GLV_ZETA_RAW) * Fq::r2();
  t[0] = G1Jac::identity();
  t[1] = p;
  for (int i = 2; i < 16; i++) t[i] = (i & 1) ? jac_add(t[i - 1], p) : jac_dbl(t[i / 2]);
  for (int i = 1; i < 16; i++) {
    if (half) t[i].x = t[i].x * zeta;
    if (sp.neg[half]) t[i].y = t[i].y.neg();
  }
}
inline G1Jac host_scalar_mul_half(const G1Jac& p, const Fr& k, int half) {
  const GlvSplit sp = glv_split(k);
  G1Jac t[16];
  glv_table(p, sp, half, t);
  G1Jac acc = G1Jac::identity();
  for (int nib = 31; nib >= 0; nib--) {
    for (int d = 0; d < 4; d++) acc = jac_dbl(acc);
    const uint32_t v = (uint32_t)(sp.m[half][nib >> 4] >> ((nib & 15) * 4)) & 15u;
    if (v) acc = jac_add(acc, t[v]);
  }
  return acc;
}
inline G1Jac host_scalar_mul(const G1Jac& p, const Fr& k) {
  const GlvSplit sp = glv_split(k);
  G1Jac t1[16], t2[16];
  glv_table(p, sp, 0, t1);
  glv_table(p, sp, 1, t2);
  G1Jac acc = G1Jac::identity();
  for (int nib = 31; nib >= 0; nib--) {  // one chain of doublings for both halves (Straus)
    for (int d = 0; d < 4; d++) acc = jac_dbl(acc);
    const uint32_t v1 = (uint32_t)(sp.m[0][nib >> 4] >> ((nib & 15) * 4)) & 15u;
    const uint32_t v2 = (uint32_t)(sp.m[1][nib >> 4] >> ((nib & 15) * 4)) & 15u;
    if (v1) acc = jac_add(acc, t1[v1]);
    if (v2) acc = jac_add(acc, t2[v2]);
  }
  return acc;
}

}  // namespace cq
