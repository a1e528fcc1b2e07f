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
  const Fq29 y3 = Fq29::mul2(r, Fq29::sub<8>(q, x3), Fq29::neg<4>(acc.y), ppp);  // r (q - x3) - y1 ppp, one reduction: 6 * 10 + 4 * 2 = 68  ->  y3 < 2
  acc.x = x3;
  acc.y = y3;
  Fq29::mul_pair(acc.zz, pp, acc.zzz, ppp, acc.zz, acc.zzz);  // 4, 4
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

}  // namespace cq
