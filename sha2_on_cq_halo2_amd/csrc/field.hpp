// BN254 Fr / Fq arithmetic for gfx950 (and the host side of this library).
//
// Values are Montgomery residues a*2^256 mod p kept as eight 32-bit limbs whose byte image is
// identical to the reference's `[u64;4]` little-endian limbs (arithmetic/curves/src/bn256/fr.rs:25,
// fq.rs:25), so buffers cross the C ABI without conversion.  All results are fully reduced
// (< p), like the reference's `add/sub/mul/montgomery_reduce` (derive/field.rs:351-616): the
// byte image of every output is therefore identical to the reference's.
//
// Why 32-bit limbs: CDNA4's integer multiplier is 32x32 (v_mad_u64_u32 / v_mul_hi_u32); a 256-bit
// Montgomery product is 2*64 of those plus carries.  No MFMA path applies to modular integers.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CQ_HD __host__ __device__ __forceinline__
#define CQ_UNROLL _Pragma("unroll")
// First statement of the kernels a proof WAITS for while a throughput kernel of the other stream shares their SIMDs (the
// launch tails, the sort's small kernels, the batch inversion, the main stream's transforms): s_setprio raises the wave's
// issue priority, so their dependent chains run at the speed they have alone and the co-resident NTT waves take the issue
// slots that are left -- stream priorities only order the DISPATCH of workgroups.  -DCQ_CRIT_PRIO=0 is the A/B switch.
#ifndef CQ_CRIT_PRIO
#define CQ_CRIT_PRIO 3
#endif
#define CQ_CRITICAL_WAVES() __builtin_amdgcn_s_setprio(CQ_CRIT_PRIO)
#else
#define CQ_HD inline
#define CQ_UNROLL
#endif

// Hides a value's known bit-width from the optimiser.  hipcc (ROCm 7.2) lowers 64-bit products of two
// operands it can prove to be < 2^24 through its 24-bit multiply patterns and, on gfx950, produced wrong
// high halves for `(x & 0xffffff) * constant` (the 24-bit mask was dropped); values that are 24 bits
// wide by construction are therefore passed through this before being multiplied.
#if defined(__AMDGCN__)
#define CQ_OPAQUE32(x) asm volatile("" : "+v"(x))
#else
#define CQ_OPAQUE32(x) (void)(x)
#endif

namespace cq {

struct alignas(16) U256 {
  uint32_t l[8];
};

// ---- moduli (constants from bn256/fr.rs:29-66 and bn256/fq.rs:29-58, split into 32-bit limbs) ----
struct FrP {
  static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xefffffffu;  // low word of fr.rs:39
  static constexpr uint32_t R[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                    0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t R3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu,
                                     0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};
};

struct FqP {
  static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t INV = 0xe4866389u;  // low word of fq.rs:37
  static constexpr uint32_t R[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                    0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr uint32_t R3[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u,
                                     0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};
};

template <class P>
struct Fp {
  U256 v;

  static CQ_HD Fp zero() {
    Fp r;
    CQ_UNROLL for (int i = 0; i < 8; i++) r.v.l[i] = 0;
    return r;
  }
  static CQ_HD Fp one() {
    Fp r;
    CQ_UNROLL for (int i = 0; i < 8; i++) r.v.l[i] = P::R[i];
    return r;
  }
  static CQ_HD Fp r2() {
    Fp r;
    CQ_UNROLL for (int i = 0; i < 8; i++) r.v.l[i] = P::R2[i];
    return r;
  }
  static CQ_HD Fp r3() {
    Fp r;
    CQ_UNROLL for (int i = 0; i < 8; i++) r.v.l[i] = P::R3[i];
    return r;
  }
  static CQ_HD Fp from_limbs64(const uint64_t* s) {
    Fp r;
    CQ_UNROLL for (int i = 0; i < 4; i++) {
      r.v.l[2 * i] = (uint32_t)s[i];
      r.v.l[2 * i + 1] = (uint32_t)(s[i] >> 32);
    }
    return r;
  }
  CQ_HD void to_limbs64(uint64_t* d) const {
    CQ_UNROLL for (int i = 0; i < 4; i++) d[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
  }

  CQ_HD bool is_zero() const {
    uint32_t o = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) o |= v.l[i];
    return o == 0;
  }
  CQ_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) o |= v.l[i] ^ b.v.l[i];
    return o == 0;
  }
  CQ_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = a - p if a >= p (a < 2p assumed)
  static CQ_HD void cond_sub_p(uint32_t* a, uint32_t top) {
    uint32_t t[8];
    uint64_t br = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)a[i] - P::MOD[i] - br;
      t[i] = (uint32_t)d;
      br = (d >> 32) & 1;
    }
    // take t if (top:a) >= p  <=> top != 0 or no borrow
    bool ge = top || !br;
    CQ_UNROLL for (int i = 0; i < 8; i++) a[i] = ge ? t[i] : a[i];
  }

  CQ_HD Fp operator+(const Fp& b) const {
    Fp r;
    uint64_t c = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      c += (uint64_t)v.l[i] + b.v.l[i];
      r.v.l[i] = (uint32_t)c;
      c >>= 32;
    }
    cond_sub_p(r.v.l, (uint32_t)c);  // p < 2^254 so c is always 0, kept for generality
    return r;
  }
  CQ_HD Fp operator-(const Fp& b) const {
    Fp r;
    uint64_t br = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)v.l[i] - b.v.l[i] - br;
      r.v.l[i] = (uint32_t)d;
      br = (d >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      c += (uint64_t)r.v.l[i] + (P::MOD[i] & mask);
      r.v.l[i] = (uint32_t)c;
      c >>= 32;
    }
    return r;
  }
  CQ_HD Fp neg() const { return zero() - *this; }
  CQ_HD Fp dbl() const { return *this + *this; }

  // limb j of p in base 2^29
  static constexpr uint32_t p29(int j) {
    const int bit = 29 * j, w = bit >> 5, sh = bit & 31;
    const uint64_t lo = P::MOD[w];
    const uint64_t hi = (w + 1 < 8) ? P::MOD[w + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> sh) & 0x1fffffffu);
  }
  // 8 x 32-bit words -> 9 x 29-bit limbs
  static CQ_HD void unpack29(const uint32_t* x, uint32_t* a) {
    a[0] = x[0] & 0x1fffffffu;
    a[1] = ((x[0] >> 29) | (x[1] << 3)) & 0x1fffffffu;
    a[2] = ((x[1] >> 26) | (x[2] << 6)) & 0x1fffffffu;
    a[3] = ((x[2] >> 23) | (x[3] << 9)) & 0x1fffffffu;
    a[4] = ((x[3] >> 20) | (x[4] << 12)) & 0x1fffffffu;
    a[5] = ((x[4] >> 17) | (x[5] << 15)) & 0x1fffffffu;
    a[6] = ((x[5] >> 14) | (x[6] << 18)) & 0x1fffffffu;
    a[7] = ((x[6] >> 11) | (x[7] << 21)) & 0x1fffffffu;
    a[8] = x[7] >> 8;
  }

  // Montgomery product (same function as derive/field.rs:471-564); result < p.
  // Device: unsaturated 29-bit limbs (v_mad_u64_u32, below).  Host: CIOS over 64-bit limbs with a 128-bit
  // accumulator -- identical values, for the host-side glue (window folding, normalisation, transcript scalars).
  CQ_HD Fp operator*(const Fp& b) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // Unsaturated 9 x 29-bit limbs: every partial product lands in a 64-bit column accumulator with a
    // single v_mad_u64_u32 and no carry handling (9 products of < 2^58 plus 9 reduction products stay
    // below 2^63).  Montgomery reduction runs digit-serially in base 2^29 for 8 digits and finishes with
    // one 24-bit digit so that the overall divisor is exactly 2^256 (8*29 + 24): the value in memory
    // keeps the reference's R = 2^256 form.  ~170 multiplies + ~170 simple ops (a CIOS over 8 x 32-bit limbs
    // needed 136 + ~430: its carry plumbing turns into v_mov / 64-bit adds).
    constexpr uint32_t M29 = 0x1fffffffu;
    uint32_t A[9], B[9];
    unpack29(v.l, A);
    unpack29(b.v.l, B);
    CQ_OPAQUE32(A[8]);
    CQ_OPAQUE32(B[8]);
    uint64_t c[18];
    CQ_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)A[i] * B[j];
    }
    constexpr uint32_t INV29 = P::INV & M29;
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      const uint32_t m = ((uint32_t)c[i] * INV29) & M29;
      CQ_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)m * p29(j);
      c[i + 1] += c[i] >> 29;
    }
    {
      uint32_t m = ((uint32_t)c[8] * INV29) & 0x00ffffffu;
      CQ_OPAQUE32(m);
      CQ_UNROLL for (int j = 0; j < 9; j++) c[8 + j] += (uint64_t)m * p29(j);
    }
    // V = sum_{k>=8} c[k] 2^(29(k-8)) has its low 24 bits clear; result = V >> 24 (< 2p)
    uint32_t d[10];
    uint64_t carry = 0;
    CQ_UNROLL for (int k = 0; k < 10; k++) {
      const uint64_t sres = c[8 + k] + carry;
      d[k] = (uint32_t)sres & M29;
      carry = sres >> 29;
    }
    uint32_t r29[9];
    CQ_UNROLL for (int j = 0; j < 9; j++) r29[j] = (d[j] >> 24) | ((d[j + 1] << 5) & M29);
    Fp r;
    r.v.l[0] = r29[0] | (r29[1] << 29);
    r.v.l[1] = (r29[1] >> 3) | (r29[2] << 26);
    r.v.l[2] = (r29[2] >> 6) | (r29[3] << 23);
    r.v.l[3] = (r29[3] >> 9) | (r29[4] << 20);
    r.v.l[4] = (r29[4] >> 12) | (r29[5] << 17);
    r.v.l[5] = (r29[5] >> 15) | (r29[6] << 14);
    r.v.l[6] = (r29[6] >> 18) | (r29[7] << 11);
    r.v.l[7] = (r29[7] >> 21) | (r29[8] << 8);
    cond_sub_p(r.v.l, 0);
    return r;
#else
    typedef unsigned __int128 u128;
    uint64_t a4[4], b4[4], m4[4];
    for (int i = 0; i < 4; i++) {
      a4[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
      b4[i] = (uint64_t)b.v.l[2 * i] | ((uint64_t)b.v.l[2 * i + 1] << 32);
      m4[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    // 64-bit -p^-1 mod 2^64 from the 32-bit one by one Newton step: inv64 = inv32 * (2 + p0 * inv32)
    const uint64_t i32 = P::INV;
    const uint64_t inv64 = i32 * (2 + m4[0] * i32);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a4[j] * b4[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * inv64;
      c = (u128)m * m4[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * m4[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fp r;
    for (int i = 0; i < 4; i++) {
      r.v.l[2 * i] = (uint32_t)t[i];
      r.v.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    cond_sub_p(r.v.l, (uint32_t)t[4]);
    return r;
#endif
  }
  CQ_HD Fp sqr() const { return *this * *this; }

  // canonical (non-Montgomery) value: a*R -> a   (`to_repr`, fr.rs:245-261)
  CQ_HD U256 to_canonical() const {
    Fp o;
    CQ_UNROLL for (int i = 0; i < 8; i++) o.v.l[i] = (i == 0) ? 1u : 0u;
    return (*this * o).v;
  }
  // canonical integer (< p) -> Montgomery   (`from_raw`, derive/field.rs:51-53)
  static CQ_HD Fp from_canonical(const U256& a) {
    Fp x;
    x.v = a;
    return x * r2();
  }
  static CQ_HD Fp from_u64(uint64_t a) {
    U256 u;
    CQ_UNROLL for (int i = 0; i < 8; i++) u.l[i] = 0;
    u.l[0] = (uint32_t)a;
    u.l[1] = (uint32_t)(a >> 32);
    return from_canonical(u);
  }
  // `from_u512` (derive/field.rs:29-47): d0*R2 + d1*R3
  static CQ_HD Fp from_u512(const uint64_t* w8) {
    Fp d0 = from_limbs64(w8), d1 = from_limbs64(w8 + 4);
    return d0 * r2() + d1 * r3();
  }

  // x^e, e given as 8 x 32-bit little-endian words
  CQ_HD Fp pow(const uint32_t* e) const {
    Fp acc = one();
    for (int w = 7; w >= 0; w--) {
      for (int bit = 31; bit >= 0; bit--) {
        acc = acc.sqr();
        if ((e[w] >> bit) & 1) acc = acc * *this;
      }
    }
    return acc;
  }
  CQ_HD Fp pow_u64(uint64_t e) const {
    Fp acc = one();
    Fp base = *this;
    while (e) {
      if (e & 1) acc = acc * base;
      base = base.sqr();
      e >>= 1;
    }
    return acc;
  }
  // Fermat inversion a^(p-2) (fr.rs:200-209); inverse of zero is zero.
  CQ_HD Fp inv_fermat() const {
    uint32_t e[8];
    CQ_UNROLL for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
    e[0] -= 2;  // low limb of both moduli is >= 2
    return pow(e);
  }
  // the inverse every caller uses: the same field element (inverse of zero is zero), a tenth of the work
  CQ_HD Fp inv() const { return inv_safegcd(); }
  // The same inverse by Bernstein-Yang "safegcd" division steps (the 30-bit-limb form used by libsecp256k1's modinv32,
  // restated): 20 batches of 30 branch-free steps on the low words of (f, g) = (p, x) build a 2 x 2 transition matrix
  // with entries below 2^30, which is then applied to the 9-limb f, g (exactly divisible by 2^30) and to d, e modulo p.
  // 600 >= the 590 steps a 256-bit modulus needs.  ~12 000 dependent instructions instead of the ~40 000 of the binary
  // Euclid below: where ONE lane inverts (the total of a batch inversion) that chain is what the workgroup waits for.
  CQ_HD Fp inv_safegcd() const { return inv_safegcd_impl<false>(); }
  // Variable-time form of the same (for the places where ONE lane inverts and a data-dependent loop costs nothing): the
  // division steps with delta starting at 1, in runs -- the trailing zeros of g leave in one shift, and after them g takes
  // the multiple of f (w = -g / f modulo 2^4, or 2^6 right after a swap) that clears its next few bits at once -- and the
  // batches stop when g = 0 (typically 17-19 of them; at most 741 steps for a 256-bit modulus).  Same matrix conventions,
  // same (d, e) / (f, g) updates, same result.
  CQ_HD Fp inv_safegcd_var() const { return inv_safegcd_impl<true>(); }
  template <bool VAR>
  CQ_HD Fp inv_safegcd_impl() const {
    if (is_zero()) return zero();
    constexpr int32_t M30 = 0x3fffffff;
    constexpr uint32_t PINV30 = (0u - P::INV) & 0x3fffffffu;  // p^-1 mod 2^30
    int32_t d[9], e[9], f[9], g[9], pm[9];
    CQ_UNROLL for (int i = 0; i < 9; i++) {
      const int bit = 30 * i, word = bit >> 5, sh = bit & 31;
      uint64_t tp = P::MOD[word], tx = v.l[word];
      if (word + 1 < 8) {
        tp |= (uint64_t)P::MOD[word + 1] << 32;
        tx |= (uint64_t)v.l[word + 1] << 32;
      }
      pm[i] = (int32_t)((tp >> sh) & M30);
      f[i] = pm[i];
      g[i] = (int32_t)((tx >> sh) & M30);
      d[i] = 0;
      e[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;  // -(delta + 1/2); VAR: eta = -delta
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int batch = 0; batch < (VAR ? 26 : 20); batch++) {
      if (VAR) {
        int32_t any = 0;
        CQ_UNROLL for (int i = 0; i < 9; i++) any |= g[i];
        if (any == 0) break;
      }
      // 30 division steps on the low words; (u, v; q, r) = the transition matrix times 2^30
      uint32_t u = 1, vv = 0, q = 0, r = 1, fl = (uint32_t)f[0] | ((uint32_t)f[1] << 30), gl = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
      if (VAR) {
        int left = 30;
        for (;;) {
          const int zeros = __builtin_ctz(gl | (0xffffffffu << left));  // at most `left`
          gl >>= zeros;
          u <<= zeros;
          vv <<= zeros;
          zeta -= zeros;
          left -= zeros;
          if (left == 0) break;
          uint32_t w;  // f and g odd here
          if (zeta < 0) {  // delta > 0: (f, g) <- (g, -f), the matrix rows with them
            zeta = -zeta;
            uint32_t t = fl; fl = gl; gl = 0u - t;
            t = u; u = q; q = 0u - t;
            t = vv; vv = r; r = 0u - t;
            const int limit = zeta + 1 > left ? left : zeta + 1;
            w = (fl * gl * (fl * fl - 2u)) & ((0xffffffffu >> (32 - limit)) & 63u);  // f (f^2 - 2) = -1 / f mod 2^6
          } else {
            const int limit = zeta + 1 > left ? left : zeta + 1;
            w = fl + (((fl + 1u) & 4u) << 1);                                      // 1 / f mod 2^4
            w = (0u - w * gl) & ((0xffffffffu >> (32 - limit)) & 15u);
          }
          gl += fl * w;
          q += u * w;
          r += vv * w;
        }
      } else {
      CQ_UNROLL for (int i = 0; i < 30; i++) {
        uint32_t m1 = (uint32_t)(zeta >> 31);
        const uint32_t m2 = 0u - (gl & 1u);
        const uint32_t x = (fl ^ m1) - m1, y = (u ^ m1) - m1, z = (vv ^ m1) - m1;
        gl += x & m2;
        q += y & m2;
        r += z & m2;
        m1 &= m2;
        zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;
        fl += gl & m1;
        u += q & m1;
        vv += r & m1;
        gl >>= 1;
        u <<= 1;
        vv <<= 1;
      }
      }
      const int64_t tu = (int32_t)u, tv = (int32_t)vv, tq = (int32_t)q, tr = (int32_t)r;
      {  // (d, e) <- (t / 2^30) (d, e) mod p, both kept in (-2p, p)
        const int32_t sd = d[8] >> 31, se = e[8] >> 31;
        int32_t md = ((int32_t)tu & sd) + ((int32_t)tv & se), me = ((int32_t)tq & sd) + ((int32_t)tr & se);
        int64_t cd = tu * d[0] + tv * e[0], ce = tq * d[0] + tr * e[0];
        md -= (int32_t)((PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd += (int64_t)pm[0] * md;
        ce += (int64_t)pm[0] * me;
        cd >>= 30;
        ce >>= 30;
        CQ_UNROLL for (int i = 1; i < 9; i++) {
          cd += tu * d[i] + tv * e[i] + (int64_t)pm[i] * md;
          ce += tq * d[i] + tr * e[i] + (int64_t)pm[i] * me;
          d[i - 1] = (int32_t)cd & M30;
          e[i - 1] = (int32_t)ce & M30;
          cd >>= 30;
          ce >>= 30;
        }
        d[8] = (int32_t)cd;
        e[8] = (int32_t)ce;
      }
      {  // (f, g) <- (t / 2^30) (f, g), exact
        int64_t cf = tu * f[0] + tv * g[0], cg = tq * f[0] + tr * g[0];
        cf >>= 30;
        cg >>= 30;
        CQ_UNROLL for (int i = 1; i < 9; i++) {
          cf += tu * f[i] + tv * g[i];
          cg += tq * f[i] + tr * g[i];
          f[i - 1] = (int32_t)cf & M30;
          g[i - 1] = (int32_t)cg & M30;
          cf >>= 30;
          cg >>= 30;
        }
        f[8] = (int32_t)cf;
        g[8] = (int32_t)cg;
      }
    }
    // now g = 0, f = +-1, d = +-x^-1 in (-2p, p): into [0, p)
    {
      int32_t add = d[8] >> 31;
      const int32_t neg = f[8] >> 31;
      CQ_UNROLL for (int i = 0; i < 9; i++) {
        d[i] += pm[i] & add;
        d[i] = (d[i] ^ neg) - neg;
      }
      CQ_UNROLL for (int i = 0; i < 8; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
      }
      add = d[8] >> 31;
      CQ_UNROLL for (int i = 0; i < 9; i++) d[i] += pm[i] & add;
      CQ_UNROLL for (int i = 0; i < 8; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
      }
    }
    Fp y;
    CQ_UNROLL for (int w = 0; w < 8; w++) {
      const int bit = 32 * w, li = bit / 30, sh = bit - 30 * li;  // word w = bits of limbs li, li + 1 (and li + 2 when sh > 28)
      uint64_t t = (uint64_t)(uint32_t)d[li] >> sh;
      t |= (uint64_t)(uint32_t)d[li + 1] << (30 - sh);
      if (li + 2 < 9) t |= (uint64_t)(uint32_t)d[li + 2] << (60 - sh);
      y.v.l[w] = (uint32_t)t;
    }
    return y * r3();
  }
  // The same inverse by the binary extended Euclid on the 8 x 32-bit integers: ~750 data-dependent steps of shifts
  // and additions instead of ~380 dependent modular products -- about 3x shorter as a chain.  It diverges across
  // lanes, so it is for the places where ONE lane inverts (the total of a batch inversion).  With y = (aR)^-1 as an
  // integer, mont_mul(y, R^3) = a^-1 R.
  CQ_HD Fp inv_euclid() const {
    if (is_zero()) return zero();
    uint32_t u[8], w[8], x1[8], x2[8];
    CQ_UNROLL for (int i = 0; i < 8; i++) {
      u[i] = v.l[i];
      w[i] = P::MOD[i];
      x1[i] = i == 0 ? 1u : 0u;
      x2[i] = 0;
    }
    auto is_one = [](const uint32_t* a) {
      uint32_t o = a[0] ^ 1u;
      CQ_UNROLL for (int i = 1; i < 8; i++) o |= a[i];
      return o == 0;
    };
    auto halve = [](uint32_t* a, uint32_t* x) {  // a even: a /= 2, x = x / 2 mod p
      CQ_UNROLL for (int i = 0; i < 7; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
      a[7] >>= 1;
      uint32_t top = 0;
      if (x[0] & 1u) {  // x + p < 2p < 2^255
        uint64_t c = 0;
        CQ_UNROLL for (int i = 0; i < 8; i++) {
          c += (uint64_t)x[i] + P::MOD[i];
          x[i] = (uint32_t)c;
          c >>= 32;
        }
        top = (uint32_t)c;
      }
      CQ_UNROLL for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
      x[7] = (x[7] >> 1) | (top << 31);
    };
    auto geq = [](const uint32_t* a, const uint32_t* b) {  // borrow-free form: no data-dependent indexing (registers, not scratch)
      uint64_t br = 0;
      CQ_UNROLL for (int i = 0; i < 8; i++) br = (((uint64_t)a[i] - b[i] - br) >> 32) & 1;
      return br == 0;
    };
    auto sub_into = [](uint32_t* a, const uint32_t* b) -> uint32_t {  // a -= b, returns the borrow
      uint64_t br = 0;
      CQ_UNROLL for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a[i] - b[i] - br;
        a[i] = (uint32_t)d;
        br = (d >> 32) & 1;
      }
      return (uint32_t)br;
    };
    auto sub_mod = [&](uint32_t* x, const uint32_t* y) {  // x = x - y mod p, x, y < p
      if (sub_into(x, y)) {
        uint64_t c = 0;
        CQ_UNROLL for (int i = 0; i < 8; i++) {
          c += (uint64_t)x[i] + P::MOD[i];
          x[i] = (uint32_t)c;
          c >>= 32;
        }
      }
    };
    while (!is_one(u) && !is_one(w)) {
      while (!(u[0] & 1u)) halve(u, x1);
      while (!(w[0] & 1u)) halve(w, x2);
      if (geq(u, w)) {
        sub_into(u, w);
        sub_mod(x1, x2);
      } else {
        sub_into(w, u);
        sub_mod(x2, x1);
      }
    }
    Fp y;
    const bool from_u = is_one(u);  // (selected by value: a pointer to one of two local arrays would put both in scratch memory)
    CQ_UNROLL for (int i = 0; i < 8; i++) y.v.l[i] = from_u ? x1[i] : x2[i];
    return y * r3();
  }
};

using Fr = Fp<FrP>;
using Fq = Fp<FqP>;

// field constants the protocol needs (bn256/fr.rs:72-118), canonical 64-bit limbs
static constexpr uint32_t FR_S = 28;
static constexpr uint64_t FR_ROOT_OF_UNITY_RAW[4] = {0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull,
                                                     0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull};
static constexpr uint64_t FR_ZETA_RAW[4] = {0xb8ca0b2d36636f23ull, 0xcc37a73fec2bc5e9ull,
                                            0x048b6e193fd84104ull, 0x30644e72e131a029ull};

// DELTA = 7^(2^28): generator of the t-order multiplicative subgroup (fr.rs:112-118)
static constexpr uint64_t FR_DELTA_RAW[4] = {0x870e56bbe533e9a2ull, 0x5b5f898e5e963f25ull,
                                             0x64ec26aad4c86e71ull, 0x09226b6e22c6f0caull};

// (r - 1) / 2: canonical values above it stand for negative integers where a field element encodes a small signed one
static constexpr uint64_t FR_HALF_RAW[4] = {0xa1f0fac9f8000000ull, 0x9419f4243cdcb848ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull};
inline Fr fr_from_raw(const uint64_t* raw) {
  return Fr::from_limbs64(raw) * Fr::r2();
}

}  // namespace cq
