// BN254 G1 (y^2 = x^3 + 3, bn256/curve.rs:66-68) group law for gfx950 and the host side.
//
// Layouts match the reference: affine = {x,y} 64 B with (0,0) = identity, Jacobian = {x,y,z} 96 B
// with z = 0 = identity (derive/curve.rs:157-168,453-463,696-705).  Bucket accumulators use
// extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is
// 8M+2S and needs no field inversion.  Only canonical affine encodings ever reach the
// transcript (derive/curve.rs:635-646), so the choice of coordinates cannot change proof bytes.
#pragma once
#include "field.hpp"

namespace cq {

struct alignas(16) G1Affine {
  Fq x, y;
  CQ_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
  static CQ_HD G1Affine identity() { return {Fq::zero(), Fq::zero()}; }
  CQ_HD G1Affine neg() const { return {x, y.is_zero() ? y : y.neg()}; }
};

struct alignas(16) G1Jac {
  Fq x, y, z;
  static CQ_HD G1Jac identity() { return {Fq::zero(), Fq::zero(), Fq::zero()}; }
  CQ_HD bool is_identity() const { return z.is_zero(); }
};

struct XYZZ {
  Fq x, y, zz, zzz;
  static CQ_HD XYZZ identity() { return {Fq::zero(), Fq::zero(), Fq::zero(), Fq::zero()}; }
  CQ_HD bool is_identity() const { return zz.is_zero(); }
  static CQ_HD XYZZ from_affine(const G1Affine& a) {
    if (a.is_identity()) return identity();
    return {a.x, a.y, Fq::one(), Fq::one()};
  }
  // (X*ZZ, Y*ZZZ, ZZ) is a Jacobian representative of the same point.
  CQ_HD G1Jac to_jac() const {
    if (is_identity()) return G1Jac::identity();
    return {x * zz, y * zzz, zz};
  }
};

// affine doubling -> XYZZ (mdbl-2008-s-1)
static CQ_HD XYZZ xyzz_dbl_affine(const G1Affine& a) {
  if (a.is_identity() || a.y.is_zero()) return XYZZ::identity();
  Fq u = a.y.dbl();
  Fq v = u.sqr();
  Fq w = u * v;
  Fq s = a.x * v;
  Fq x2 = a.x.sqr();
  Fq m = x2.dbl() + x2;
  Fq x3 = m.sqr() - s.dbl();
  Fq y3 = m * (s - x3) - w * a.y;
  return {x3, y3, v, w};
}

// dbl-2008-s-1
static CQ_HD XYZZ xyzz_dbl(const XYZZ& p) {
  if (p.is_identity() || p.y.is_zero()) return XYZZ::identity();
  Fq u = p.y.dbl();
  Fq v = u.sqr();
  Fq w = u * v;
  Fq s = p.x * v;
  Fq x2 = p.x.sqr();
  Fq m = x2.dbl() + x2;
  Fq x3 = m.sqr() - s.dbl();
  Fq y3 = m * (s - x3) - w * p.y;
  return {x3, y3, v * p.zz, w * p.zzz};
}

// acc += a  (madd-2008-s), complete: handles identity operands, a == acc and a == -acc
static CQ_HD void xyzz_add_affine(XYZZ& acc, const G1Affine& a) {
  if (a.is_identity()) return;
  if (acc.is_identity()) {
    acc = {a.x, a.y, Fq::one(), Fq::one()};
    return;
  }
  Fq u2 = a.x * acc.zz;
  Fq s2 = a.y * acc.zzz;
  Fq p = u2 - acc.x;
  Fq r = s2 - acc.y;
  if (p.is_zero()) {
    if (r.is_zero()) {
      acc = xyzz_dbl_affine(a);
    } else {
      acc = XYZZ::identity();
    }
    return;
  }
  Fq pp = p.sqr();
  Fq ppp = p * pp;
  Fq q = acc.x * pp;
  Fq x3 = r.sqr() - ppp - q.dbl();
  Fq y3 = r * (q - x3) - acc.y * ppp;
  acc.x = x3;
  acc.y = y3;
  acc.zz = acc.zz * pp;
  acc.zzz = acc.zzz * ppp;
}

// acc += b  (add-2008-s), complete
static CQ_HD void xyzz_add(XYZZ& acc, const XYZZ& b) {
  if (b.is_identity()) return;
  if (acc.is_identity()) {
    acc = b;
    return;
  }
  Fq u1 = acc.x * b.zz;
  Fq u2 = b.x * acc.zz;
  Fq s1 = acc.y * b.zzz;
  Fq s2 = b.y * acc.zzz;
  Fq p = u2 - u1;
  Fq r = s2 - s1;
  if (p.is_zero()) {
    if (r.is_zero()) {
      acc = xyzz_dbl(acc);
    } else {
      acc = XYZZ::identity();
    }
    return;
  }
  Fq pp = p.sqr();
  Fq ppp = p * pp;
  Fq q = u1 * pp;
  Fq x3 = r.sqr() - ppp - q.dbl();
  Fq y3 = r * (q - x3) - s1 * ppp;
  acc.x = x3;
  acc.y = y3;
  acc.zz = acc.zz * b.zz * pp;
  acc.zzz = acc.zzz * b.zzz * ppp;
}

// ---- Jacobian ops (host side: folding window sums, normalising results) ----
static CQ_HD G1Jac jac_dbl(const G1Jac& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Jac::identity();
  Fq a = p.x.sqr();
  Fq b = p.y.sqr();
  Fq c = b.sqr();
  Fq d = ((p.x + b).sqr() - a - c).dbl();
  Fq e = a.dbl() + a;
  Fq f = e.sqr();
  Fq x3 = f - d.dbl();
  Fq y3 = e * (d - x3) - c.dbl().dbl().dbl();
  Fq z3 = (p.y * p.z).dbl();
  return {x3, y3, z3};
}

static CQ_HD G1Jac jac_add(const G1Jac& p, const G1Jac& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq z1z1 = p.z.sqr();
  Fq z2z2 = q.z.sqr();
  Fq u1 = p.x * z2z2;
  Fq u2 = q.x * z1z1;
  Fq s1 = p.y * q.z * z2z2;
  Fq s2 = q.y * p.z * z1z1;
  if (u1 == u2) {
    if (s1 == s2) return jac_dbl(p);
    return G1Jac::identity();
  }
  Fq h = u2 - u1;
  Fq r = s2 - s1;
  Fq hh = h.sqr();
  Fq hhh = h * hh;
  Fq v = u1 * hh;
  Fq x3 = r.sqr() - hhh - v.dbl();
  Fq y3 = r * (v - x3) - s1 * hhh;
  Fq z3 = p.z * q.z * h;
  return {x3, y3, z3};
}

static CQ_HD G1Jac jac_from_affine(const G1Affine& a) {
  if (a.is_identity()) return G1Jac::identity();
  return {a.x, a.y, Fq::one()};
}

// `to_affine` (derive/curve.rs:399-412)
static CQ_HD G1Affine jac_to_affine(const G1Jac& p) {
  if (p.is_identity()) return G1Affine::identity();
  Fq zi = p.z.inv();
  Fq zi2 = zi.sqr();
  return {p.x * zi2, p.y * zi2 * zi};
}

}  // namespace cq
