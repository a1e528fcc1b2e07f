// Device-only "lazy" field type for the elliptic-curve kernels: the value lives in registers as 9 unsaturated
// 29-bit limbs in Montgomery form with R' = 2^261 (nine whole digits), not 8 x 32 with R = 2^256.
//
// Why: with Fp (field.hpp) every product unpacks two operands to 29-bit limbs, repacks the result and ends
// with a conditional subtraction -- about as many instructions as the 162 multiplies themselves.  Keeping
// operands in limb form between operations, and letting values range over [0, K p) instead of [0, p),
// removes all of that: a product is 81 + 81 v_mad_u64_u32, 9 digit multiplies and the carry shifts.
//
// Contract (K-bounds are tracked in comments at every call site in curve29.hpp):
//   * "normalised": limbs 0..7 < 2^29 (the top limb holds whatever is left).  mul() needs limb products to fit
//     64-bit columns: limbs of one operand < 2^30 and of the other < 2^30 (9 * 2^60 + 9 * 2^58 + carries < 2^64).
//   * mul(a, b) with a < Ka p, b < Kb p, Ka * Kb <= 128: result normalised and < 2 p   (R' = 2^261 > 128 p).
//   * add() is limb-wise (no carries); sub<K>(a, b) = a + (K p - b) with K p in a borrow-proof limb form, then
//     normalised; requires b < K p with normalised-or-sum limbs (< 2^30).
// Memory formats stay what they were: R = 2^256 Montgomery values (the reference's bytes) are converted with one
// product on load (from_mont256) / store (to_mont256); library-internal arrays (window tables, partial sums)
// hold the R' form packed into 8 x u32 (values < 2^256 there).
#pragma once
#include <type_traits>
#include "field.hpp"

// acc += a * b as the NEXT link of a chain of v_mad_u64_u32 (mul_pair / sqr_pair below): the empty asm makes the sum
// opaque, so the compiler cannot re-associate a column's additions (it would sum the products first and bring the
// carry in with a separate 64-bit addition -- the very instruction the chains are there to save)
#if defined(__HIP_DEVICE_COMPILE__)
#define CQ_CHAIN(acc, a, b)       \
  do {                            \
    (acc) += (uint64_t)(a) * (b); \
    asm("" : "+v"(acc));          \
  } while (0)
#else
#define CQ_CHAIN(acc, a, b) ((acc) += (uint64_t)(a) * (b))
#endif

namespace cq {

// ---- compile-time limb constants ----------------------------------------------------------------------------
struct Consts29 {
  uint32_t one[9];      // R' mod p      (the field's 1)
  uint32_t to256[9];    // 2^256 mod p   : mul(x, to256) turns x R' into x R
  uint32_t from256[9];  // 2^266 mod p   : mul(y, from256) turns y = x R into x R'
  uint32_t c271[9];     // 2^271 mod p   : mul(z R, c271) = z 2^266, the constant that converts AND scales by z (ntt.hip)
};
struct Limbs29 {
  uint32_t l[9];
};

// 2^e mod p by repeated doubling on 9 x 29-bit limbs
template <class P>
constexpr void pow2_mod_p29(int e, uint32_t* out) {
  constexpr uint32_t M29 = 0x1fffffffu;
  uint32_t v[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < e; s++) {
    uint32_t carry = 0;
    for (int i = 0; i < 9; i++) {
      const uint32_t t = (v[i] << 1) | carry;
      carry = t >> 29;
      v[i] = t & M29;
    }
    bool ge = true;  // v < 2p: subtract p once if v >= p
    for (int i = 8; i >= 0; i--) {
      if (v[i] != Fp<P>::p29(i)) {
        ge = v[i] > Fp<P>::p29(i);
        break;
      }
    }
    if (ge) {
      uint32_t borrow = 0;
      for (int i = 0; i < 9; i++) {
        const uint32_t sub_ = Fp<P>::p29(i) + borrow;
        if (v[i] >= sub_) {
          v[i] -= sub_;
          borrow = 0;
        } else {
          v[i] = v[i] + (1u << 29) - sub_;
          borrow = 1;
        }
      }
    }
  }
  for (int i = 0; i < 9; i++) out[i] = v[i];
}
template <class P>
constexpr Consts29 make_consts29() {
  Consts29 c{};
  pow2_mod_p29<P>(261, c.one);
  pow2_mod_p29<P>(256, c.to256);
  pow2_mod_p29<P>(266, c.from256);
  pow2_mod_p29<P>(271, c.c271);
  return c;
}
template <class P>
inline constexpr Consts29 CONSTS29 = make_consts29<P>();

// K*p in "borrow-proof" limb form: every limb below the top one carries an extra 2^PAD that the next limb pays for
// (2^PAD * 2^(29 j) = 2^(PAD-29) * 2^(29 (j+1))), so  limb_j - b_j  cannot go negative for b_j < 2^PAD.  The limbs
// still sum to K p.
template <class P, uint32_t K, uint32_t PAD>
constexpr Limbs29 make_kp29() {
  Limbs29 r{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    const uint64_t t = (uint64_t)Fp<P>::p29(i) * K + carry;
    r.l[i] = i < 8 ? (uint32_t)(t & 0x1fffffffu) : (uint32_t)t;
    carry = t >> 29;
  }
  for (int i = 0; i < 8; i++) {
    r.l[i] += 1u << PAD;
    r.l[i + 1] -= 1u << (PAD - 29);
  }
  return r;
}
template <class P, uint32_t K, uint32_t PAD>
inline constexpr Limbs29 KP29 = make_kp29<P, K, PAD>();

template <class P>
struct Fp29 {
  uint32_t a[9];

  static constexpr uint32_t M29 = 0x1fffffffu;
  static constexpr uint32_t pl(int j) { return Fp<P>::p29(j); }
  // -p^-1 mod 2^29
  static constexpr uint32_t NINV = P::INV & M29;

  static __device__ __forceinline__ Fp29 zero() {
    Fp29 r;
    CQ_UNROLL for (int i = 0; i < 9; i++) r.a[i] = 0;
    return r;
  }
  __device__ __forceinline__ bool limbs_zero() const {
    uint32_t o = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) o |= a[i];
    return o == 0;
  }
  // value == 0 (mod p) for a normalised value < 2 p: the limbs are all zero or spell p
  __device__ __forceinline__ bool is_zero_mod_p() const {
    uint32_t o = 0, q = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      o |= a[i];
      q |= a[i] ^ pl(i);
    }
    return o == 0 || q == 0;
  }

  // carry propagation: limbs 0..7 back below 2^29
  __device__ __forceinline__ void normalise() {
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      a[i + 1] += a[i] >> 29;
      a[i] &= M29;
    }
  }

  // Montgomery product, divisor 2^261
  static __device__ __forceinline__ Fp29 mul(const Fp29& x, const Fp29& y) {
    uint64_t c[18];
    CQ_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)x.a[i] * y.a[j];
    }
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      const uint32_t m = ((uint32_t)c[i] * NINV) & M29;
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * pl(j);
      c[i + 1] += c[i] >> 29;
    }
    Fp29 r;
    CQ_UNROLL for (int k = 0; k < 8; k++) {
      r.a[k] = (uint32_t)c[9 + k] & M29;
      c[10 + k] += c[9 + k] >> 29;
    }
    r.a[8] = (uint32_t)c[17];
    return r;
  }
  // x y + z w with ONE reduction (81 multiplies fewer than two products and an addition).  All four operands
  // normalised (limbs < 2^29: 18 * 2^58 + 9 * 2^58 + carries < 2^63); bound: Kx Ky + Kz Kw <= 128  ->  result < 2 p.
  static __device__ __forceinline__ Fp29 mul2(const Fp29& x, const Fp29& y, const Fp29& z, const Fp29& w) {
    uint64_t c[18];
    CQ_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)x.a[i] * y.a[j];
    }
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)z.a[i] * w.a[j];
    }
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      const uint32_t m = ((uint32_t)c[i] * NINV) & M29;
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * pl(j);
      c[i + 1] += c[i] >> 29;
    }
    Fp29 r;
    CQ_UNROLL for (int k = 0; k < 8; k++) {
      r.a[k] = (uint32_t)c[9 + k] & M29;
      c[10 + k] += c[9 + k] >> 29;
    }
    r.a[8] = (uint32_t)c[17];
    return r;
  }
  __device__ __forceinline__ Fp29 operator*(const Fp29& o) const { return mul(*this, o); }
  // ---- sums of products with one reduction ------------------------------------------------------------------------
  // mac(): c += x y as eighteen 64-bit columns (no reduction); redc(): the Montgomery reduction of such columns (< 2 p
  // when the summed bounds stay <= 128).  All operands normalised (limbs < 2^29): a column then holds 9 limb products below
  // 2^58 per product and the reduction's 9 -- up to SIX products fit a reduction (6 x 9 + 9 = 63 < 64).  For the pointwise
  // kernels over R = 2^256 values (poly.hip): a memory word a R read as limbs IS the R' value of a / 32, so the product of
  // such a value with a constant given in R' form (the host passes 32 x, canonical) is the memory form of the true product
  // -- sums of them need no conversion at either end.
  static __device__ __forceinline__ void mac(uint64_t* c, const Fp29& x, const Fp29& y) {
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)x.a[i] * y.a[j];
    }
  }
  static __device__ __forceinline__ Fp29 redc(uint64_t* c) {
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      const uint32_t m = ((uint32_t)c[i] * NINV) & M29;
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * pl(j);
      c[i + 1] += c[i] >> 29;
    }
    Fp29 r;
    CQ_UNROLL for (int k = 0; k < 8; k++) {
      r.a[k] = (uint32_t)c[9 + k] & M29;
      c[10 + k] += c[9 + k] >> 29;
    }
    r.a[8] = (uint32_t)c[17];
    return r;
  }
  // value < 64 p, normalised -> the canonical 8 x u32 words (one product by the field's one, pack, conditional subtraction)
  __device__ __forceinline__ void to_canonical_words(uint32_t* w) const {
    const Fp29 r = mul(*this, one());
    r.pack(w);
    Fp<P>::cond_sub_p(w, 0);
  }
  // ---- two independent products at once, column by column ------------------------------------------------------
  // mul() above is written row by row (c[i + j] += x_i y_j over eighteen column registers): the carry of every column is
  // a shift and a 64-bit addition, 2 x 17 instructions per product.  Column by column (product scanning) ONE 64-bit
  // accumulator takes the column's limb products, the reduction's products m_i p_(k-i) of the digits already chosen, its
  // own digit's m_k p_0, and is shifted down into the next column: every addition is the addend of a v_mad_u64_u32, and
  // the 17 additions go (222 -> 205 instructions per product).  But such a chain is one long dependence, and hipcc pads a
  // v_mad_u64_u32 that accumulates in place right behind another one with s_nop (gfx950 hazard recogniser): one product
  // written that way came out as 1 467 multiply-adds + 1 237 s_nop in the accumulate kernel's addition.  Two independent
  // products with their links alternating need a sixth of those -- and the group law and the radix-4 butterfly offer their
  // products in pairs (curve29.hpp, ntt.hip).  Measured on one MI355X, same box (tools/ab_pairs.sh, round 3): accumulate
  // kernel -2.0 % at k = 18 and -4.5 % at k = 22, NTT +3 %, proofs 7.9 -> 7.75 / 25.2 -> 24.2 / 91.3 -> 88.2 ms.
  // CQ_MUL_NO_PAIRS falls back to two plain products (the A/B).  Bounds per product exactly as for mul() / sqr(): the
  // columns hold the same sums.
  // (compile-time loops: with the asm statements in the body the unroller gives up on ordinary ones)
  template <int I, int N, class F>
  static __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
      f(std::integral_constant<int, I>{});
      static_for<I + 1, N>(f);
    }
  }
  // the reduction's share of column K and the column's digit, for both chains
  template <int K>
  static __device__ __forceinline__ void pair_digit(uint64_t& a1, uint64_t& a2, uint32_t* m1, uint32_t* m2, Fp29& r1, Fp29& r2) {
    static_for<0, 9>([&](auto I) {
      constexpr int i = decltype(I)::value, j = K - i;
      if constexpr (j >= 1 && j <= 8 && i < K) {
        CQ_CHAIN(a1, m1[i], pl(j));
        CQ_CHAIN(a2, m2[i], pl(j));
      }
    });
    if constexpr (K < 9) {
      m1[K] = ((uint32_t)a1 * NINV) & M29;
      m2[K] = ((uint32_t)a2 * NINV) & M29;
      CQ_CHAIN(a1, m1[K], pl(0));
      CQ_CHAIN(a2, m2[K], pl(0));
    } else {
      r1.a[K - 9] = (uint32_t)a1 & M29;
      r2.a[K - 9] = (uint32_t)a2 & M29;
    }
    a1 >>= 29;
    a2 >>= 29;
  }
  // r1 = x1 y1, r2 = x2 y2
  static __device__ __forceinline__ void mul_pair(const Fp29& x1, const Fp29& y1, const Fp29& x2, const Fp29& y2, Fp29& r1, Fp29& r2) {
#ifdef CQ_MUL_NO_PAIRS
    r1 = mul(x1, y1);
    r2 = mul(x2, y2);
#else
    uint64_t a1 = 0, a2 = 0;
    uint32_t m1[9], m2[9];
    Fp29 o1, o2;  // (the outputs may alias the inputs)
    static_for<0, 17>([&](auto KK) {
      constexpr int K = decltype(KK)::value;
      static_for<0, 9>([&](auto I) {
        constexpr int i = decltype(I)::value, j = K - i;
        if constexpr (j >= 0 && j <= 8) {
          CQ_CHAIN(a1, x1.a[i], y1.a[j]);
          CQ_CHAIN(a2, x2.a[i], y2.a[j]);
        }
      });
      pair_digit<K>(a1, a2, m1, m2, o1, o2);
    });
    o1.a[8] = (uint32_t)a1;
    o2.a[8] = (uint32_t)a2;
    r1 = o1;
    r2 = o2;
#endif
  }
  // r0 = x y + z w (one reduction, as mul2), r1 = a b, r2 = c d: three chains, their links in the order 0 1 0 2 so that no
  // chain follows itself (the group law's last step: Y3 = R (Q - X3) - Y1 PPP next to ZZ PP and ZZZ PPP)
  static __device__ __forceinline__ void mul2_mul_mul(const Fp29& x, const Fp29& y, const Fp29& z, const Fp29& w, const Fp29& a, const Fp29& b,
                                                      const Fp29& c, const Fp29& d, Fp29& r0, Fp29& r1, Fp29& r2) {
#ifdef CQ_MUL_NO_PAIRS
    r0 = mul2(x, y, z, w);
    r1 = mul(a, b);
    r2 = mul(c, d);
#else
    uint64_t a0 = 0, a1 = 0, a2 = 0;
    uint32_t m0[9], m1[9], m2[9];
    Fp29 o0, o1, o2;
    static_for<0, 17>([&](auto KK) {
      constexpr int K = decltype(KK)::value;
      static_for<0, 9>([&](auto I) {
        constexpr int i = decltype(I)::value, j = K - i;
        if constexpr (j >= 0 && j <= 8) {
          CQ_CHAIN(a0, x.a[i], y.a[j]);
          CQ_CHAIN(a1, a.a[i], b.a[j]);
          CQ_CHAIN(a0, z.a[i], w.a[j]);
          CQ_CHAIN(a2, c.a[i], d.a[j]);
        }
      });
      static_for<0, 9>([&](auto I) {
        constexpr int i = decltype(I)::value, j = K - i;
        if constexpr (j >= 1 && j <= 8 && i < K) {
          CQ_CHAIN(a0, m0[i], pl(j));
          CQ_CHAIN(a1, m1[i], pl(j));
          CQ_CHAIN(a2, m2[i], pl(j));
        }
      });
      if constexpr (K < 9) {
        m0[K] = ((uint32_t)a0 * NINV) & M29;
        m1[K] = ((uint32_t)a1 * NINV) & M29;
        m2[K] = ((uint32_t)a2 * NINV) & M29;
        CQ_CHAIN(a0, m0[K], pl(0));
        CQ_CHAIN(a1, m1[K], pl(0));
        CQ_CHAIN(a2, m2[K], pl(0));
      } else {
        o0.a[K - 9] = (uint32_t)a0 & M29;
        o1.a[K - 9] = (uint32_t)a1 & M29;
        o2.a[K - 9] = (uint32_t)a2 & M29;
      }
      a0 >>= 29;
      a1 >>= 29;
      a2 >>= 29;
    });
    o0.a[8] = (uint32_t)a0;
    o1.a[8] = (uint32_t)a1;
    o2.a[8] = (uint32_t)a2;
    r0 = o0;
    r1 = o1;
    r2 = o2;
#endif
  }
  // r1 = x1^2, r2 = x2^2 (cross products once, against doubled limbs, as in sqr())
  static __device__ __forceinline__ void sqr_pair(const Fp29& x1, const Fp29& x2, Fp29& r1, Fp29& r2) {
#if defined(CQ_MUL_NO_PAIRS) || defined(CQ_NO_SQR)
    r1 = x1.sqr();
    r2 = x2.sqr();
#else
    uint64_t a1 = 0, a2 = 0;
    uint32_t m1[9], m2[9], d1[9], d2[9];
    Fp29 o1, o2;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      d1[i] = x1.a[i] << 1;
      d2[i] = x2.a[i] << 1;
    }
    static_for<0, 17>([&](auto KK) {
      constexpr int K = decltype(KK)::value;
      static_for<0, 9>([&](auto I) {
        constexpr int i = decltype(I)::value, j = K - i;
        if constexpr (j > i && j <= 8) {
          CQ_CHAIN(a1, x1.a[i], d1[j]);
          CQ_CHAIN(a2, x2.a[i], d2[j]);
        }
        if constexpr (j == i) {
          CQ_CHAIN(a1, x1.a[i], x1.a[i]);
          CQ_CHAIN(a2, x2.a[i], x2.a[i]);
        }
      });
      pair_digit<K>(a1, a2, m1, m2, o1, o2);
    });
    o1.a[8] = (uint32_t)a1;
    o2.a[8] = (uint32_t)a2;
    r1 = o1;
    r2 = o2;
#endif
  }
  // x^2: the 36 cross products once, against doubled limbs (45 multiplies instead of 81).  Limbs < 2^30 as for mul():
  // a column holds at most 4 cross products < 2^61, a square < 2^60 and the reduction's 9 * 2^58 -- below 2^64.
  __device__ __forceinline__ Fp29 sqr() const {
#ifdef CQ_NO_SQR  // A/B knob (tools/ab_flags_stats.sh): accumulate kernel 1.10 -> 1.07 ms per k = 18 launch with the squaring
    return mul(*this, *this);
#endif
    uint64_t c[18];
    uint32_t d[9];
    CQ_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) d[i] = a[i] << 1;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      c[2 * i] += (uint64_t)a[i] * a[i];
      CQ_UNROLL for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a[i] * d[j];
    }
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      const uint32_t m = ((uint32_t)c[i] * NINV) & M29;
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * pl(j);
      c[i + 1] += c[i] >> 29;
    }
    Fp29 r;
    CQ_UNROLL for (int k = 0; k < 8; k++) {
      r.a[k] = (uint32_t)c[9 + k] & M29;
      c[10 + k] += c[9 + k] >> 29;
    }
    r.a[8] = (uint32_t)c[17];
    return r;
  }

  // limb-wise sum (no carries): limbs < 2^30 for two normalised operands
  __device__ __forceinline__ Fp29 operator+(const Fp29& o) const {
    Fp29 r;
    CQ_UNROLL for (int i = 0; i < 9; i++) r.a[i] = a[i] + o.a[i];
    return r;
  }

  // a - b + K p, normalised.  Needs b < K p, limbs of b < 2^PAD and limbs of a < 2^30 (PAD = 30) / 2^29 (PAD = 31).
  template <uint32_t K, uint32_t PAD = 30>
  static __device__ __forceinline__ Fp29 sub(const Fp29& x, const Fp29& y) {
    Fp29 r;
    CQ_UNROLL for (int i = 0; i < 9; i++) r.a[i] = x.a[i] + (KP29<P, K, PAD>.l[i] - y.a[i]);
    r.normalise();
    return r;
  }
  // K p - b, normalised
  template <uint32_t K>
  static __device__ __forceinline__ Fp29 neg(const Fp29& y) {
    Fp29 r;
    CQ_UNROLL for (int i = 0; i < 9; i++) r.a[i] = KP29<P, K, 30>.l[i] - y.a[i];
    r.normalise();
    return r;
  }

  // ---- packing: value < 2^256 <-> 8 x u32 --------------------------------------------------------------
  static __device__ __forceinline__ Fp29 unpack(const uint32_t* x) {
    Fp29 r;
    Fp<P>::unpack29(x, r.a);
    return r;
  }
  // normalised limbs, value < 2^256
  __device__ __forceinline__ void pack(uint32_t* o) const {
    o[0] = a[0] | (a[1] << 29);
    o[1] = (a[1] >> 3) | (a[2] << 26);
    o[2] = (a[2] >> 6) | (a[3] << 23);
    o[3] = (a[3] >> 9) | (a[4] << 20);
    o[4] = (a[4] >> 12) | (a[5] << 17);
    o[5] = (a[5] >> 15) | (a[6] << 14);
    o[6] = (a[6] >> 18) | (a[7] << 11);
    o[7] = (a[7] >> 21) | (a[8] << 8);
  }

  static __device__ __forceinline__ Fp29 one() {
    Fp29 r;
    CQ_UNROLL for (int i = 0; i < 9; i++) r.a[i] = CONSTS29<P>.one[i];
    return r;
  }
  // R = 2^256 Montgomery value (8 x u32, < p) -> R' limb form, < 2 p
  static __device__ __forceinline__ Fp29 from_mont256(const Fp<P>& y) {
    Fp29 u = unpack(y.v.l), f;
    CQ_UNROLL for (int i = 0; i < 9; i++) f.a[i] = CONSTS29<P>.from256[i];
    return mul(u, f);
  }
  // value < 64 p (normalised limbs) -> canonical R = 2^256 Montgomery value < p
  __device__ __forceinline__ Fp<P> to_mont256() const {
    Fp29 t;
    CQ_UNROLL for (int i = 0; i < 9; i++) t.a[i] = CONSTS29<P>.to256[i];
    Fp29 r = mul(*this, t);  // < 2 p
    Fp<P> o;
    r.pack(o.v.l);
    Fp<P>::cond_sub_p(o.v.l, 0);
    return o;
  }
  // reduce a normalised value < 64 p to < 2 p (one product with the field's 1)
  __device__ __forceinline__ Fp29 reduced() const { return mul(*this, one()); }
};

using Fq29 = Fp29<FqP>;
using Fr29 = Fp29<FrP>;

}  // namespace cq
