// BN254 G1 group law on the lazy 9 x 29-bit field (field29.hpp) for the MSM kernels.
//
// Same formulas as curve.hpp (XYZZ: madd-2008-s, add-2008-s, dbl-2008-s-1) -- results are the same group
// elements, so nothing that reaches the transcript changes -- but coordinates stay in limb form in registers
// and range over [0, K p).  Invariant of every XYZZ29 held in registers, LDS or memory:
//     x < 8 p,  y < 4 p,  zz, zzz < 2 p,  limbs normalised;  identity <=> all limbs of zz are zero
// (zz of a finite point is non-zero mod p, so its limbs cannot all vanish; the special cases that produce the
// identity set it explicitly).  The bound of every intermediate is given as a multiple of p in the comments; a
// product needs (bound a) * (bound b) <= 128.
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace cq {

struct XYZZ29 {
  Fq29 x, y, zz, zzz;
  static __device__ __forceinline__ XYZZ29 identity() { return {Fq29::zero(), Fq29::zero(), Fq29::zero(), Fq29::zero()}; }
  __device__ __forceinline__ bool is_identity() const { return zz.limbs_zero(); }
};

// affine point in R' form (x, y < 2 p); identity = (0, 0) with all limbs zero
struct Affine29 {
  Fq29 x, y;
  __device__ __forceinline__ bool is_identity() const { return x.limbs_zero() && y.limbs_zero(); }
};

// ---- memory <-> registers ---------------------------------------------------------------------------------
static __device__ __forceinline__ void ld8(const void* p, uint32_t* w) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
static __device__ __forceinline__ void st8(void* p, const uint32_t* w) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// table / SRS point.  `mont256`: the array holds the reference's R = 2^256 values (converted with one product per
// coordinate); otherwise it is a library-built table already in R' form.
static __device__ __forceinline__ Affine29 load_affine29(const G1Affine* p, bool mont256) {
  uint32_t wx[8], wy[8];
  ld8(&p->x, wx);
  ld8(&p->y, wy);
  Affine29 r;
  r.x = Fq29::unpack(wx);
  r.y = Fq29::unpack(wy);
  if (mont256) {
    Fq29 f;
    CQ_UNROLL for (int i = 0; i < 9; i++) f.a[i] = CONSTS29<FqP>.from256[i];
    r.x = Fq29::mul(r.x, f);  // 0 stays 0 (identity)
    r.y = Fq29::mul(r.y, f);
  }
  return r;
}
// partial sums / buckets: 4 x (8 x u32) holding R'-form values below 2^256 (x is reduced first: 8 p > 2^256)
static __device__ __forceinline__ XYZZ29 load_xyzz29(const XYZZ* p) {
  uint32_t w[8];
  XYZZ29 r;
  ld8(&p->x, w);   r.x = Fq29::unpack(w);
  ld8(&p->y, w);   r.y = Fq29::unpack(w);
  ld8(&p->zz, w);  r.zz = Fq29::unpack(w);
  ld8(&p->zzz, w); r.zzz = Fq29::unpack(w);
  return r;
}
// the same in two steps, so that a serial sum can have the next operand's 128 bytes in flight while it adds the current one
struct XYZZRaw {
  uint32_t w[32];
};
static __device__ __forceinline__ XYZZRaw load_xyzz_raw(const XYZZ* p) {
  XYZZRaw r;
  ld8(&p->x, r.w);
  ld8(&p->y, r.w + 8);
  ld8(&p->zz, r.w + 16);
  ld8(&p->zzz, r.w + 24);
  return r;
}
static __device__ __forceinline__ XYZZ29 unpack_xyzz29(const XYZZRaw& r) {
  return {Fq29::unpack(r.w), Fq29::unpack(r.w + 8), Fq29::unpack(r.w + 16), Fq29::unpack(r.w + 24)};
}
static __device__ __forceinline__ void store_xyzz29(XYZZ* p, const XYZZ29& v) {
  uint32_t w[8];
  const Fq29 xr = v.x.reduced();  // < 2 p
  xr.pack(w);    st8(&p->x, w);
  v.y.pack(w);   st8(&p->y, w);   // 4 p < 2^256
  v.zz.pack(w);  st8(&p->zz, w);
  v.zzz.pack(w); st8(&p->zzz, w);
}

// ---- group law ----------------------------------------------------------------------------------------------
// 2 * (ax, ay), affine in (x, y < 2 p)
static __device__ __forceinline__ XYZZ29 xyzz29_dbl_affine(const Affine29& a) {
  if (a.is_identity()) return XYZZ29::identity();
  const Fq29 u = a.y + a.y;                    // < 4
  const Fq29 v = u.sqr();                      // 16 -> < 2
  const Fq29 w = u * v;                        // 8
  const Fq29 s = a.x * v;                      // 4
  const Fq29 x2 = a.x.sqr();                   // 4
  Fq29 m = x2 + x2 + x2;                       // < 6, limbs < 3 * 2^29
  m.normalise();
  const Fq29 x3 = Fq29::sub<4>(m.sqr(), s + s);                   // 36; s + s < 4  ->  x3 < 6
  const Fq29 y3 = Fq29::mul2(m, Fq29::sub<8>(s, x3), w, Fq29::neg<2>(a.y));  // m (s - x3) - w y: 6 * 10 + 2 * 2 = 64  ->  y3 < 2
  return {x3, y3, v, w};
}

static __device__ __forceinline__ XYZZ29 xyzz29_dbl(const XYZZ29& p) {
  if (p.is_identity()) return XYZZ29::identity();
  const Fq29 u = p.y + p.y;                    // < 8, limbs < 2^30
  const Fq29 v = u.sqr();                      // 64
  const Fq29 w = u * v;                        // 16
  const Fq29 s = p.x * v;                      // 16
  const Fq29 x2 = p.x.sqr();                   // 64
  Fq29 m = x2 + x2 + x2;                       // < 6
  m.normalise();
  const Fq29 x3 = Fq29::sub<4>(m.sqr(), s + s);                   // x3 < 6
  const Fq29 y3 = Fq29::mul2(m, Fq29::sub<8>(s, x3), w, Fq29::neg<4>(p.y));  // 6 * 10 + 2 * 4 = 68  ->  y3 < 2
  return {x3, y3, v * p.zz, w * p.zzz};
}

// acc += a, complete (identity operands, a == acc, a == -acc)
static __device__ __forceinline__ void xyzz29_add_affine(XYZZ29& acc, const Affine29& a) {
  if (a.is_identity()) return;
  if (acc.is_identity()) {
    acc = {a.x, a.y, Fq29::one(), Fq29::one()};
    return;
  }
  // the independent products go in pairs (Fp29::mul_pair: 17 instructions fewer per product than one at a time)
  Fq29 u2, s2, pp, rr, ppp, q;
  Fq29::mul_pair(a.x, acc.zz, a.y, acc.zzz, u2, s2);  // 4, 4
  const Fq29 p = Fq29::sub<8>(u2, acc.x);      // < 10
  const Fq29 r = Fq29::sub<4>(s2, acc.y);      // < 6
  Fq29::sqr_pair(p, r, pp, rr);                // 100, 36
  if (pp.is_zero_mod_p()) {                    // same x: doubling or cancellation
    if (rr.is_zero_mod_p()) acc = xyzz29_dbl_affine(a);
    else acc = XYZZ29::identity();
    return;
  }
  Fq29::mul_pair(p, pp, acc.x, pp, ppp, q);    // 20, 16
  const Fq29 x3 = Fq29::sub<6, 31>(rr, ppp + q + q);                    // subtrahend < 6, limbs < 3 * 2^29  ->  x3 < 8
  // r (q - x3) - y1 ppp with one reduction (6 * 10 + 4 * 2 = 68  ->  y3 < 2), interleaved with zz pp and zzz ppp (4, 4)
  Fq29 y3, zz3, zzz3;
  Fq29::mul2_mul_mul(r, Fq29::sub<8>(q, x3), Fq29::neg<4>(acc.y), ppp, acc.zz, pp, acc.zzz, ppp, y3, zz3, zzz3);
  acc.x = x3;
  acc.y = y3;
  acc.zz = zz3;
  acc.zzz = zzz3;
}

// acc += b, complete
static __device__ __forceinline__ void xyzz29_add(XYZZ29& acc, const XYZZ29& b) {
  if (b.is_identity()) return;
  if (acc.is_identity()) {
    acc = b;
    return;
  }
  Fq29 u1, u2, s1, s2, pp, rr, ppp, q, zz, zzz;
  Fq29::mul_pair(acc.x, b.zz, b.x, acc.zz, u1, u2);      // 16, 16
  Fq29::mul_pair(acc.y, b.zzz, b.y, acc.zzz, s1, s2);    // 8, 8
  const Fq29 p = Fq29::sub<2>(u2, u1);         // < 4
  const Fq29 r = Fq29::sub<2>(s2, s1);         // < 4
  Fq29::sqr_pair(p, r, pp, rr);                // 16, 16
  if (pp.is_zero_mod_p()) {
    if (rr.is_zero_mod_p()) acc = xyzz29_dbl(acc);
    else acc = XYZZ29::identity();
    return;
  }
  Fq29::mul_pair(p, pp, u1, pp, ppp, q);       // 8, 4
  Fq29::mul_pair(acc.zz, b.zz, acc.zzz, b.zzz, zz, zzz);  // 4, 4
  const Fq29 x3 = Fq29::sub<6, 31>(rr, ppp + q + q);                   // x3 < 8
  const Fq29 y3 = Fq29::mul2(r, Fq29::sub<8>(q, x3), Fq29::neg<2>(s1), ppp);  // 4 * 10 + 2 * 2 = 44  ->  y3 < 2
  acc.x = x3;
  acc.y = y3;
  Fq29::mul_pair(zz, pp, zzz, ppp, acc.zz, acc.zzz);     // 4, 4
}

// Jacobian representative (X ZZ, Y ZZZ, ZZ) in the reference's layout: canonical R = 2^256 Montgomery values
static __device__ __forceinline__ G1Jac xyzz29_to_jac(const XYZZ29& v) {
  if (v.is_identity()) return G1Jac::identity();
  return {(v.x * v.zz).to_mont256(), (v.y * v.zzz).to_mont256(), v.zz.to_mont256()};  // 16, 8
}

static __device__ __forceinline__ XYZZ29 xyzz29_shfl_down(const XYZZ29& a, int delta) {
  XYZZ29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    r.x.a[k] = __shfl_down(a.x.a[k], delta, 64);
    r.y.a[k] = __shfl_down(a.y.a[k], delta, 64);
    r.zz.a[k] = __shfl_down(a.zz.a[k], delta, 64);
    r.zzz.a[k] = __shfl_down(a.zzz.a[k], delta, 64);
  }
  return r;
}

// ---- one addition by FOUR lanes ----------------------------------------------------------------------------------
// The tails of an MSM launch (row / column sums, bit-plane trees) are chains of dependent general additions on a handful
// of waves: what counts there is the length of ONE addition -- 14 products one after the other for a lone wave, ~5 us --
// not the work.  But the 14 products are only four deep.  A QUAD (four adjacent lanes) holds a point one coordinate per
// lane -- lane r of the quad: x, y, zz, zzz for r = 0..3 -- and computes P1 + P2 in four steps of ONE product each, every
// lane multiplying a different pair of operands it picked up from its neighbours with DPP quad permutes:
//     step 1   x1 zz2 | zz1 x2 | y1 zzz2 | zzz1 y2          = u1 | u2 | s1 | s2
//     step 2   p p    | r r    | zz1 zz2 | zzz1 zzz2        p = u2 - u1,  r = s2 - s1
//     step 3   p pp   | u1 pp  | zz12 pp | --               = ppp | q | zz3 | --
//     step 4   s1 ppp | r (q - x3) | --  | zzz12 ppp        x3 = rr - ppp - 2 q (in lane 1);  = . | . | -- | zzz3
//     y3 = r (q - x3) - s1 ppp (lane 1);  result x3 | y3 | zz3 | zzz3
// ~1 300 instructions against ~2 700, at a quarter of the lanes' throughput -- for launches that leave the SIMDs idle.
// Same formulas and bounds as xyzz29_add (every lane computes every line; the lanes a line is not meant for hold values
// that are never used).  Identity operands are selected around; equal or opposite x (doubling, cancellation) falls back
// to xyzz29_add on the gathered points, for the whole wave when any quad needs it (needs_slow is wave-uniform).
template <int P0, int P1, int P2, int P3>
static __device__ __forceinline__ Fq29 quad_perm(const Fq29& a) {
  Fq29 r;
  CQ_UNROLL for (int k = 0; k < 9; k++) {
    r.a[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.a[k], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
    // (opaque: hipcc 7.2 folds the move into the instruction that uses it, and "a' - a''" with two DIFFERENT permutes of
    // the same register came out with one of them applied to both -- tools/micro/quad_steps_test.hip)
    asm volatile("" : "+v"(r.a[k]));
  }
  return r;
}
// one word from lane K of the quad to its four lanes
template <int K>
static __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, true);
  asm volatile("" : "+v"(r));
  return r;
}
static __device__ __forceinline__ Fq29 select29(bool c, const Fq29& a, const Fq29& b) {
  Fq29 r;
  CQ_UNROLL for (int k = 0; k < 9; k++) r.a[k] = c ? a.a[k] : b.a[k];
  return r;
}
// F: this quad's point, G: the other point, both one coordinate per lane (x | y | zz | zzz).  Returns F + G in that form.
static __device__ __forceinline__ Fq29 quad_add(const Fq29& F, const Fq29& G) {
  const uint32_t role = threadIdx.x & 3u;
  // identity <=> all limbs of zz (lane 2) are zero
  uint32_t zf = 0, zg = 0;
  CQ_UNROLL for (int k = 0; k < 9; k++) {
    zf |= F.a[k];
    zg |= G.a[k];
  }
  const bool id1 = quad_bcast<2>(zf) == 0;
  const bool id2 = quad_bcast<2>(zg) == 0;
  // step 1
  const Fq29 A1 = quad_perm<0, 2, 1, 3>(F);   // x1 | zz1 | y1 | zzz1
  const Fq29 B1 = quad_perm<2, 0, 3, 1>(G);   // zz2 | x2 | zzz2 | y2
  const Fq29 T1 = Fq29::mul(A1, B1);          // u1 | u2 | s1 | s2     (16, 16, 8, 8)
  // step 2
  const Fq29 D = Fq29::sub<2>(quad_perm<1, 3, 1, 3>(T1), quad_perm<0, 2, 0, 2>(T1));  // p | r | (p | r), < 4
  const Fq29 T2 = Fq29::mul(select29(role < 2, D, F), select29(role < 2, D, G));     // pp | rr | zz12 | zzz12   (16, 16, 4, 4)
  const bool same_x = quad_bcast<0>(T2.is_zero_mod_p() ? 1u : 0u) != 0;
  // step 3
  const Fq29 PP = quad_perm<0, 0, 0, 0>(T2);
  const Fq29 U1 = quad_perm<0, 0, 0, 0>(T1);
  const Fq29 T3 = Fq29::mul(role == 0 ? D : role == 1 ? U1 : T2, PP);                // ppp | q | zz3 | (zzz12 pp)   (8, 4, 4)
  // step 4
  const Fq29 PPP = quad_perm<0, 0, 0, 0>(T3);
  const Fq29 X3 = Fq29::sub<6, 31>(T2, PPP + T3 + T3);                               // lane 1: rr - ppp - 2 q, < 8
  const Fq29 W = Fq29::sub<8>(T3, X3);                                               // lane 1: q - x3, < 10
  const Fq29 S1 = quad_perm<2, 2, 2, 2>(T1);
  const Fq29 T4 = Fq29::mul(role == 0 ? S1 : role == 1 ? D : T2, role == 0 ? T3 : role == 1 ? W : PPP);  // s1 ppp | r (q - x3) | . | zzz3   (4, 40, ., 8)
  const Fq29 Y3 = Fq29::sub<2>(T4, quad_perm<0, 0, 0, 0>(T4));                       // lane 1: < 4
  const Fq29 X3all = quad_perm<1, 1, 1, 1>(X3);  // (every lane must execute a permute: not inside the conditional below)
  Fq29 R = role == 0 ? X3all : role == 1 ? Y3 : role == 2 ? T3 : T4;
  const bool needs_slow = same_x && !id1 && !id2;
  if (__any(needs_slow)) {  // doubling or cancellation somewhere in the wave: the plain formulas on the gathered points
    XYZZ29 a = {quad_perm<0, 0, 0, 0>(F), quad_perm<1, 1, 1, 1>(F), quad_perm<2, 2, 2, 2>(F), quad_perm<3, 3, 3, 3>(F)};
    const XYZZ29 b = {quad_perm<0, 0, 0, 0>(G), quad_perm<1, 1, 1, 1>(G), quad_perm<2, 2, 2, 2>(G), quad_perm<3, 3, 3, 3>(G)};
    xyzz29_add(a, b);
    const Fq29 S = role == 0 ? a.x : role == 1 ? a.y : role == 2 ? a.zz : a.zzz;
    R = select29(needs_slow, S, R);
  }
  return select29(id2, F, select29(id1, G, R));
}

}  // namespace cq
