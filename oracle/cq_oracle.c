/*
 * TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the reference's CQ / KZG proving
 * arithmetic (aleph-zero-foundation/sha2-on-cq-halo2).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product never links it.
 *
 * Each function cites the reference file:line it follows (paths under /root/reference).
 * Parity status: pinned against the reference's KATs through oracle/bn254.py (Python big-int
 * restatement checked by `from_u512` KATs and all field constants) -- see tests/test_oracle_c.py,
 * which requires bit-identical results between this file and the Python oracle.
 * The reference is Rust-only and cannot be built in this environment (no cargo/rustc).
 *
 * Values: 4 x uint64_t little-endian limbs in Montgomery form (a * 2^256 mod p), exactly the
 * reference's in-memory layout (arithmetic/curves/src/bn256/fr.rs:25).
 * Threading: OpenMP stands in for the reference's rayon pool (halo2_proofs/src/multicore.rs).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { uint64_t mod[4]; uint64_t inv; uint64_t r[4]; uint64_t r2[4]; } field_t;

/* bn256/fr.rs:29-66 */
static const field_t FR = {
    {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0xc2e1f593efffffffull,
    {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full},
    {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}};
/* bn256/fq.rs:29-58 */
static const field_t FQ = {
    {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0x87d20782e4866389ull,
    {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full},
    {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}};

/* ---- derive/field.rs helpers: mac / adc / sbb (arithmetic.rs of halo2curves) ---------------- */
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t carry, uint64_t* hi) {
  u128 t = (u128)a + (u128)b * c + carry;
  *hi = (uint64_t)(t >> 64);
  return (uint64_t)t;
}

/* derive/field.rs:397-413 `sub`; also used as the final conditional subtraction of the modulus */
static inline fe f_sub(const field_t* F, fe a, fe b) {
  fe d;
  u128 t;
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    t = (u128)a.l[i] - b.l[i] - borrow;
    d.l[i] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
  uint64_t mask = 0 - borrow;
  uint64_t carry = 0;
  for (int i = 0; i < 4; i++) {
    t = (u128)d.l[i] + (F->mod[i] & mask) + carry;
    d.l[i] = (uint64_t)t;
    carry = (uint64_t)(t >> 64);
  }
  return d;
}
static inline fe f_mod(const field_t* F) { fe m; memcpy(m.l, F->mod, 32); return m; }

/* derive/field.rs:488-501 `add` (sparse): sum then subtract the modulus */
static inline fe f_add(const field_t* F, fe a, fe b) {
  fe d;
  uint64_t carry = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a.l[i] + b.l[i] + carry;
    d.l[i] = (uint64_t)t;
    carry = (uint64_t)(t >> 64);
  }
  return f_sub(F, d, f_mod(F));
}

/* derive/field.rs:415-430 `neg` */
static inline fe f_neg(const field_t* F, fe a) {
  if ((a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0) return a;
  fe z = {{0, 0, 0, 0}};
  return f_sub(F, z, a);
}

/* derive/field.rs:503-562 `mul` (gnark-style interleaved CIOS); result < p */
static inline fe f_mul(const field_t* F, fe a, fe b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t r0, r1, k, lo;
    lo = mac(t0, a.l[i], b.l[0], 0, &r0);
    k = lo * F->inv;
    (void)mac(lo, k, F->mod[0], 0, &r1);
    lo = mac(t1, a.l[i], b.l[1], r0, &r0);
    t0 = mac(lo, k, F->mod[1], r1, &r1);
    lo = mac(t2, a.l[i], b.l[2], r0, &r0);
    t1 = mac(lo, k, F->mod[2], r1, &r1);
    lo = mac(t3, a.l[i], b.l[3], r0, &r0);
    t2 = mac(lo, k, F->mod[3], r1, &r1);
    t3 = r0 + r1;
  }
  fe r = {{t0, t1, t2, t3}};
  return f_sub(F, r, f_mod(F));
}
static inline fe f_sqr(const field_t* F, fe a) { return f_mul(F, a, a); }
static inline int f_is_zero(fe a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline int f_eq(fe a, fe b) { return memcmp(a.l, b.l, 32) == 0; }
static inline fe f_one(const field_t* F) { fe r; memcpy(r.l, F->r, 32); return r; }
static inline fe f_zero(void) { fe r = {{0, 0, 0, 0}}; return r; }
static inline fe f_dbl(const field_t* F, fe a) { return f_add(F, a, a); }

/* `pow` / `pow_vartime` of ff 0.12 (absent from /root/reference): square-and-multiply, MSB first */
static fe f_pow(const field_t* F, fe a, const uint64_t e[4]) {
  fe acc = f_one(F);
  for (int w = 3; w >= 0; w--)
    for (int b = 63; b >= 0; b--) {
      acc = f_sqr(F, acc);
      if ((e[w] >> b) & 1) acc = f_mul(F, acc, a);
    }
  return acc;
}
/* bn256/fr.rs:200-209 `invert` = self^(p-2) */
static fe f_inv(const field_t* F, fe a) {
  uint64_t e[4] = {F->mod[0] - 2, F->mod[1], F->mod[2], F->mod[3]};
  return f_pow(F, a, e);
}
/* fr.rs:245-261 `to_repr`: Montgomery -> canonical limbs */
static fe f_to_canonical(const field_t* F, fe a) {
  fe one = {{1, 0, 0, 0}};
  return f_mul(F, a, one);
}
static fe f_from_u64(const field_t* F, uint64_t v) {
  fe a = {{v, 0, 0, 0}}, r2;
  memcpy(r2.l, F->r2, 32);
  return f_mul(F, a, r2);
}

/* ------------------------------------------------------------------------------------------------
 * G1 (derive/curve.rs).  Jacobian {x,y,z}, identity z = 0; affine {x,y}, identity (0,0).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { fe x, y, z; } jac;
typedef struct { fe x, y; } aff;
#define Q (&FQ)

static inline int jac_is_id(const jac* p) { return f_is_zero(p->z); }
static inline int aff_is_id(const aff* p) { return f_is_zero(p->x) && f_is_zero(p->y); }
static inline jac jac_id(void) { jac r; memset(&r, 0, sizeof r); return r; }

/* derive/curve.rs:422-447 `double` */
static jac jac_double(const jac* p) {
  if (jac_is_id(p)) return jac_id();
  fe a = f_sqr(Q, p->x), b = f_sqr(Q, p->y), c = f_sqr(Q, b);
  fe d = f_add(Q, p->x, b);
  d = f_sqr(Q, d);
  d = f_sub(Q, f_sub(Q, d, a), c);
  d = f_dbl(Q, d);
  fe e = f_add(Q, f_dbl(Q, a), a);
  fe f = f_sqr(Q, e);
  fe z3 = f_dbl(Q, f_mul(Q, p->z, p->y));
  fe x3 = f_sub(Q, f, f_dbl(Q, d));
  c = f_dbl(Q, f_dbl(Q, f_dbl(Q, c)));
  fe y3 = f_sub(Q, f_mul(Q, e, f_sub(Q, d, x3)), c);
  jac r = {x3, y3, z3};
  return r;
}

/* derive/curve.rs:809-851 Jacobian + Jacobian */
static jac jac_add(const jac* p, const jac* q) {
  if (jac_is_id(p)) return *q;
  if (jac_is_id(q)) return *p;
  fe z1z1 = f_sqr(Q, p->z), z2z2 = f_sqr(Q, q->z);
  fe u1 = f_mul(Q, p->x, z2z2), u2 = f_mul(Q, q->x, z1z1);
  fe s1 = f_mul(Q, f_mul(Q, p->y, z2z2), q->z), s2 = f_mul(Q, f_mul(Q, q->y, z1z1), p->z);
  if (f_eq(u1, u2)) {
    if (f_eq(s1, s2)) return jac_double(p);
    return jac_id();
  }
  fe h = f_sub(Q, u2, u1);
  fe i = f_sqr(Q, f_dbl(Q, h));
  fe j = f_mul(Q, h, i);
  fe r = f_dbl(Q, f_sub(Q, s2, s1));
  fe v = f_mul(Q, u1, i);
  fe x3 = f_sub(Q, f_sub(Q, f_sub(Q, f_sqr(Q, r), j), v), v);
  s1 = f_dbl(Q, f_mul(Q, s1, j));
  fe y3 = f_sub(Q, f_mul(Q, r, f_sub(Q, v, x3)), s1);
  fe z3 = f_sub(Q, f_sub(Q, f_sqr(Q, f_add(Q, p->z, q->z)), z1z1), z2z2);
  z3 = f_mul(Q, z3, h);
  jac out = {x3, y3, z3};
  return out;
}

/* derive/curve.rs:853-893 Jacobian + affine */
static jac jac_add_aff(const jac* p, const aff* q) {
  if (jac_is_id(p)) {
    jac r = jac_id();
    if (!aff_is_id(q)) { r.x = q->x; r.y = q->y; r.z = f_one(Q); }
    return r;
  }
  if (aff_is_id(q)) return *p;
  fe z1z1 = f_sqr(Q, p->z);
  fe u2 = f_mul(Q, q->x, z1z1);
  fe s2 = f_mul(Q, f_mul(Q, q->y, z1z1), p->z);
  if (f_eq(p->x, u2)) {
    if (f_eq(p->y, s2)) return jac_double(p);
    return jac_id();
  }
  fe h = f_sub(Q, u2, p->x);
  fe hh = f_sqr(Q, h);
  fe i = f_dbl(Q, f_dbl(Q, hh));
  fe j = f_mul(Q, h, i);
  fe r = f_dbl(Q, f_sub(Q, s2, p->y));
  fe v = f_mul(Q, p->x, i);
  fe x3 = f_sub(Q, f_sub(Q, f_sub(Q, f_sqr(Q, r), j), v), v);
  j = f_dbl(Q, f_mul(Q, p->y, j));
  fe y3 = f_sub(Q, f_mul(Q, r, f_sub(Q, v, x3)), j);
  fe z3 = f_sub(Q, f_sub(Q, f_sqr(Q, f_add(Q, p->z, h)), z1z1), hh);
  jac out = {x3, y3, z3};
  return out;
}

/* derive/curve.rs:914-935 scalar mul, MSB-first double-and-add over the canonical bytes */
static jac jac_mul(const jac* p, fe scalar_mont) {
  fe s = f_to_canonical(&FR, scalar_mont);
  jac acc = jac_id();
  for (int w = 3; w >= 0; w--)
    for (int b = 63; b >= 0; b--) {
      acc = jac_double(&acc);
      if ((s.l[w] >> b) & 1) acc = jac_add(&acc, p);
    }
  return acc;
}

/* derive/curve.rs:399-412 `to_affine` */
static aff jac_to_aff(const jac* p) {
  aff r;
  memset(&r, 0, sizeof r);
  if (jac_is_id(p)) return r;
  fe zi = f_inv(Q, p->z);
  fe zi2 = f_sqr(Q, zi);
  r.x = f_mul(Q, p->x, zi2);
  r.y = f_mul(Q, p->y, f_mul(Q, zi2, zi));
  return r;
}

/* ------------------------------------------------------------------------------------------------
 * halo2_proofs/src/arithmetic.rs
 * ---------------------------------------------------------------------------------------------- */
/* arithmetic.rs:24-42 `get_at` over the canonical little-endian bytes */
static inline size_t get_at(size_t segment, size_t c, const uint8_t* bytes) {
  size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
  if (skip_bytes >= 32) return 0;
  uint8_t v[8] = {0};
  for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
  uint64_t tmp;
  memcpy(&tmp, v, 8);
  tmp >>= skip_bits - skip_bytes * 8;
  tmp %= ((uint64_t)1 << c);
  return (size_t)tmp;
}

/* arithmetic.rs:13-101 `multiexp_serial` (buckets kept Jacobian; None == identity) */
static void multiexp_serial(const fe* coeffs, const aff* bases, size_t len, jac* acc) {
  uint8_t* reprs = (uint8_t*)malloc(len * 32 + 8);
  for (size_t i = 0; i < len; i++) {
    fe c = f_to_canonical(&FR, coeffs[i]);
    memcpy(reprs + i * 32, c.l, 32);
  }
  size_t c;
  if (len < 4) c = 1;
  else if (len < 32) c = 3;
  else c = (size_t)ceil(log((double)(uint32_t)len));
  size_t segments = 256 / c + 1;
  size_t nb = ((size_t)1 << c) - 1;
  jac* buckets = (jac*)malloc(nb * sizeof(jac));
  for (size_t seg = segments; seg-- > 0;) {
    for (size_t k = 0; k < c; k++) *acc = jac_double(acc);
    memset(buckets, 0, nb * sizeof(jac));
    for (size_t i = 0; i < len; i++) {
      size_t d = get_at(seg, c, reprs + i * 32);
      if (d != 0) buckets[d - 1] = jac_add_aff(&buckets[d - 1], &bases[i]);
    }
    jac running = jac_id();
    for (size_t b = nb; b-- > 0;) {
      running = jac_add(&running, &buckets[b]);
      *acc = jac_add(acc, &running);
    }
  }
  free(buckets);
  free(reprs);
}

static int n_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* arithmetic.rs:132-159 `best_multiexp`: chunk = len / threads, partial results folded */
void cqo_best_multiexp(const uint64_t* coeffs, const uint64_t* bases, size_t len, uint64_t out_jac[12]) {
  const fe* cs = (const fe*)coeffs;
  const aff* bs = (const aff*)bases;
  jac acc = jac_id();
  size_t threads = (size_t)n_threads();
  if (len > threads) {
    size_t chunk = len / threads;
    size_t nchunks = (len + chunk - 1) / chunk;
    jac* results = (jac*)calloc(nchunks, sizeof(jac));
#pragma omp parallel for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)nchunks; ci++) {
      size_t start = (size_t)ci * chunk;
      size_t l = start + chunk <= len ? chunk : len - start;
      multiexp_serial(cs + start, bs + start, l, &results[ci]);
    }
    for (size_t ci = 0; ci < nchunks; ci++) acc = jac_add(&acc, &results[ci]);
    free(results);
  } else {
    multiexp_serial(cs, bs, len, &acc);
  }
  memcpy(out_jac, &acc, sizeof acc);
}

/* arithmetic.rs:172-179 */
static inline size_t bitreverse(size_t n, size_t l) {
  size_t r = 0;
  for (size_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
  return r;
}

/* arithmetic.rs:237-274 `recursive_butterfly_arithmetic`; rayon::join -> omp tasks */
static void recursive_butterfly(fe* a, size_t n, size_t twiddle_chunk, const fe* tw, int depth) {
  if (n == 2) {
    fe t = a[1];
    a[1] = f_sub(&FR, a[0], t);
    a[0] = f_add(&FR, a[0], t);
    return;
  }
  fe *left = a, *right = a + n / 2;
  if (depth > 0) {
#pragma omp task
    recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp task
    recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp taskwait
  } else {
    recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, 0);
    recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, 0);
  }
  {
    fe t = right[0];
    right[0] = f_sub(&FR, left[0], t);
    left[0] = f_add(&FR, left[0], t);
  }
  for (size_t i = 1; i < n / 2; i++) {
    fe t = f_mul(&FR, right[i], tw[i * twiddle_chunk]);
    right[i] = f_sub(&FR, left[i], t);
    left[i] = f_add(&FR, left[i], t);
  }
}

/* arithmetic.rs:171-234 `best_fft` */
void cqo_best_fft(uint64_t* a_, const uint64_t omega_[4], uint32_t log_n) {
  fe* a = (fe*)a_;
  fe omega;
  memcpy(omega.l, omega_, 32);
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitreverse(k, log_n);
    if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  size_t ntw = n / 2 ? n / 2 : 1;
  fe* tw = (fe*)malloc(ntw * sizeof(fe));
  fe w = f_one(&FR);
  for (size_t i = 0; i < n / 2; i++) { tw[i] = w; w = f_mul(&FR, w, omega); }
  int threads = n_threads(), log_threads = 0;
  while ((1 << (log_threads + 1)) <= threads) log_threads++;
  if ((int)log_n <= log_threads) {
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t s = 0; s < log_n; s++) {
      for (size_t st = 0; st < n; st += chunk) {
        fe *left = a + st, *right = a + st + chunk / 2;
        fe t = right[0];
        right[0] = f_sub(&FR, left[0], t);
        left[0] = f_add(&FR, left[0], t);
        for (size_t i = 1; i < chunk / 2; i++) {
          fe t2 = f_mul(&FR, right[i], tw[i * twiddle_chunk]);
          right[i] = f_sub(&FR, left[i], t2);
          left[i] = f_add(&FR, left[i], t2);
        }
      }
      chunk *= 2;
      twiddle_chunk /= 2;
    }
  } else {
#pragma omp parallel
#pragma omp single
    recursive_butterfly(a, n, 1, tw, log_threads + 1);
  }
  free(tw);
}

/* arithmetic.rs:304-329 `eval_polynomial` (Horner, chunk per thread with x^start fix-up) */
void cqo_eval_polynomial(const uint64_t* poly_, size_t n, const uint64_t point_[4], uint64_t out[4]) {
  const fe* poly = (const fe*)poly_;
  fe x;
  memcpy(x.l, point_, 32);
  size_t threads = (size_t)n_threads();
  fe res = f_zero();
  if (n * 2 < threads) {
    for (size_t i = n; i-- > 0;) res = f_add(&FR, f_mul(&FR, res, x), poly[i]);
  } else {
    size_t chunk = (n + threads - 1) / threads;
    fe* parts = (fe*)calloc(threads, sizeof(fe));
#pragma omp parallel for
    for (long t = 0; t < (long)threads; t++) {
      size_t start = (size_t)t * chunk;
      if (start >= n) continue;
      size_t end = start + chunk < n ? start + chunk : n;
      fe acc = f_zero();
      for (size_t i = end; i-- > start;) acc = f_add(&FR, f_mul(&FR, acc, x), poly[i]);
      uint64_t e[4] = {start, 0, 0, 0};
      parts[t] = f_mul(&FR, acc, f_pow(&FR, x, e));
    }
    for (size_t t = 0; t < threads; t++) res = f_add(&FR, res, parts[t]);
    free(parts);
  }
  memcpy(out, res.l, 32);
}

/* arithmetic.rs:351-387 `kate_division`: q has n-1 coefficients */
void cqo_kate_division(const uint64_t* a_, size_t n, const uint64_t b_[4], uint64_t* q_) {
  const fe* a = (const fe*)a_;
  fe* q = (fe*)q_;
  fe b;
  memcpy(b.l, b_, 32);
  b = f_neg(&FR, b);
  fe tmp = f_zero();
  for (size_t i = n - 1; i-- > 0;) {
    fe lead = f_sub(&FR, a[i + 1], tmp);
    q[i] = lead;
    tmp = f_mul(&FR, lead, b);
  }
}

/* ff::BatchInvert (ff 0.12): Montgomery's trick; zeros stay zero */
void cqo_batch_invert(uint64_t* v_, size_t n) {
  fe* v = (fe*)v_;
  fe* pref = (fe*)malloc((n ? n : 1) * sizeof(fe));
  fe acc = f_one(&FR);
  for (size_t i = 0; i < n; i++) {
    pref[i] = acc;
    if (!f_is_zero(v[i])) acc = f_mul(&FR, acc, v[i]);
  }
  acc = f_inv(&FR, acc);
  for (size_t i = n; i-- > 0;) {
    if (f_is_zero(v[i])) continue;
    fe t = f_mul(&FR, pref[i], acc);
    acc = f_mul(&FR, acc, v[i]);
    v[i] = t;
  }
  free(pref);
}

/* per-element Fermat inversion, as static_lookup/prover.rs:261-266 does */
void cqo_invert_each(uint64_t* v_, size_t n) {
  fe* v = (fe*)v_;
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) v[i] = f_inv(&FR, v[i]);
}

/* element-wise helpers used by the restated prover */
void cqo_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  fe x, y;
  memcpy(x.l, a, 32);
  memcpy(y.l, b, 32);
  fe r = f_mul(&FR, x, y);
  memcpy(out, r.l, 32);
}
void cqo_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  fe x, y;
  memcpy(x.l, a, 32);
  memcpy(y.l, b, 32);
  fe r = f_mul(&FQ, x, y);
  memcpy(out, r.l, 32);
}
void cqo_fr_inv(const uint64_t a[4], uint64_t out[4]) {
  fe x;
  memcpy(x.l, a, 32);
  fe r = f_inv(&FR, x);
  memcpy(out, r.l, 32);
}
void cqo_g1_mul(const uint64_t p_aff[8], const uint64_t scalar[4], uint64_t out_jac[12]) {
  aff a;
  memcpy(&a, p_aff, sizeof a);
  jac p = jac_id();
  if (!aff_is_id(&a)) { p.x = a.x; p.y = a.y; p.z = f_one(Q); }
  fe s;
  memcpy(s.l, scalar, 32);
  jac r = jac_mul(&p, s);
  memcpy(out_jac, &r, sizeof r);
}
void cqo_g1_to_affine(const uint64_t p_jac[12], uint64_t out_aff[8]) {
  jac p;
  memcpy(&p, p_jac, sizeof p);
  aff a = jac_to_aff(&p);
  memcpy(out_aff, &a, sizeof a);
}
void cqo_g1_add(const uint64_t a_jac[12], const uint64_t b_jac[12], uint64_t out_jac[12]) {
  jac a, b;
  memcpy(&a, a_jac, sizeof a);
  memcpy(&b, b_jac, sizeof b);
  jac r = jac_add(&a, &b);
  memcpy(out_jac, &r, sizeof r);
}

/* ------------------------------------------------------------------------------------------------
 * poly/domain.rs wrappers
 * ---------------------------------------------------------------------------------------------- */
/* domain.rs:366-374 `ifft` */
void cqo_ifft(uint64_t* a_, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor_[4]) {
  cqo_best_fft(a_, omega_inv, log_n);
  fe* a = (fe*)a_;
  fe d;
  memcpy(d.l, divisor_, 32);
  size_t n = (size_t)1 << log_n;
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) a[i] = f_mul(&FR, a[i], d);
}

/* domain.rs:347-363 `distribute_powers_zeta`: a[i] *= c[(i%3)-1] for i%3 != 0 */
void cqo_distribute_powers(uint64_t* a_, size_t n, const uint64_t c1_[4], const uint64_t c2_[4]) {
  fe* a = (fe*)a_;
  fe c[2];
  memcpy(c[0].l, c1_, 32);
  memcpy(c[1].l, c2_, 32);
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) {
    size_t m = (size_t)i % 3;
    if (m) a[i] = f_mul(&FR, a[i], c[m - 1]);
  }
}

/* domain.rs:319-338 `divide_by_vanishing_poly`: a[i] *= t_evaluations[i % t_len] */
void cqo_mul_periodic(uint64_t* a_, size_t n, const uint64_t* t_, size_t t_len) {
  fe* a = (fe*)a_;
  const fe* t = (const fe*)t_;
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) a[i] = f_mul(&FR, a[i], t[(size_t)i % t_len]);
}

/* plonk/evaluation.rs:539-547: h[i] = h[i]*y + (b[i]*(f[i]*l_active[i] + beta) - 1) */
void cqo_cq_quotient_term(uint64_t* h_, const uint64_t* b_, const uint64_t* f_, const uint64_t* la_, size_t n,
                          const uint64_t y_[4], const uint64_t beta_[4]) {
  fe* h = (fe*)h_;
  const fe *b = (const fe*)b_, *f = (const fe*)f_, *la = (const fe*)la_;
  fe y, beta, one = f_one(&FR);
  memcpy(y.l, y_, 32);
  memcpy(beta.l, beta_, 32);
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) {
    fe t = f_add(&FR, f_mul(&FR, f[i], la[i]), beta);
    t = f_sub(&FR, f_mul(&FR, b[i], t), one);
    h[i] = f_add(&FR, f_mul(&FR, h[i], y), t);
  }
}

int cqo_num_threads(void) { return n_threads(); }
void cqo_set_num_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}

/* ================================================================================================
 * create_proof for CQ-only circuits -- restatement of plonk/prover.rs:51-779 with
 * static_lookup/prover.rs, vanishing/prover.rs, evaluation.rs:285-551 (advice cosets + CQ term),
 * gwc/prover.rs and transcript.rs, using the REFERENCE'S OWN algorithm for every step (serial
 * double-and-add for the sparse commitments, one Fermat inversion per row, BTreeMap-style ordered
 * lookups, the always-on kate_division sanity check).  This is the CPU baseline bench.py times.
 * ============================================================================================== */
#include <stdio.h>

/* ---- BLAKE2b (RFC 7693), as blake2b_simd provides it to transcript.rs:179-184 ------------------ */
typedef struct { uint64_t h[8], t[2]; uint8_t buf[128]; size_t buflen; } b2b;
static const uint64_t B2B_IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
static const uint8_t B2B_S[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void b2b_compress(b2b* s, const uint8_t* block, int last) {
  uint64_t m[16], v[16];
  memcpy(m, block, 128);
  for (int i = 0; i < 8; i++) { v[i] = s->h[i]; v[i + 8] = B2B_IV[i]; }
  v[12] ^= s->t[0];
  v[13] ^= s->t[1];
  if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y) \
  v[a] += v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
  v[a] += v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 63);
  for (int r = 0; r < 12; r++) {
    const uint8_t* z = B2B_S[r];
    G(0, 4, 8, 12, m[z[0]], m[z[1]]) G(1, 5, 9, 13, m[z[2]], m[z[3]]) G(2, 6, 10, 14, m[z[4]], m[z[5]])
    G(3, 7, 11, 15, m[z[6]], m[z[7]]) G(0, 5, 10, 15, m[z[8]], m[z[9]]) G(1, 6, 11, 12, m[z[10]], m[z[11]])
    G(2, 7, 8, 13, m[z[12]], m[z[13]]) G(3, 4, 9, 14, m[z[14]], m[z[15]])
  }
#undef G
  for (int i = 0; i < 8; i++) s->h[i] ^= v[i] ^ v[i + 8];
}
static void b2b_init(b2b* s, const char personal[16]) {
  uint8_t p[64] = {0};
  p[0] = 64; p[2] = 1; p[3] = 1;
  memcpy(p + 48, personal, 16);
  for (int i = 0; i < 8; i++) { uint64_t w; memcpy(&w, p + 8 * i, 8); s->h[i] = B2B_IV[i] ^ w; }
  s->t[0] = s->t[1] = 0;
  s->buflen = 0;
}
static void b2b_update(b2b* s, const uint8_t* in, size_t len) {
  while (len) {
    if (s->buflen == 128) {
      s->t[0] += 128;
      if (s->t[0] < 128) s->t[1]++;
      b2b_compress(s, s->buf, 0);
      s->buflen = 0;
    }
    size_t take = 128 - s->buflen;
    if (take > len) take = len;
    memcpy(s->buf + s->buflen, in, take);
    s->buflen += take; in += take; len -= take;
  }
}
static void b2b_final_clone(const b2b* s0, uint8_t out[64]) {
  b2b s = *s0;
  s.t[0] += s.buflen;
  if (s.t[0] < s.buflen) s.t[1]++;
  memset(s.buf + s.buflen, 0, 128 - s.buflen);
  b2b_compress(&s, s.buf, 1);
  memcpy(out, s.h, 64);
}

/* ---- transcript.rs:170-241 (Blake2bWrite + Challenge255) ------------------------------------------ */
typedef struct { b2b st; uint8_t* proof; size_t len; } transcript;
static fe fr_from_u512(const uint64_t w[8]) { /* derive/field.rs:29-47 */
  static const uint64_t R3L[4] = {0x5e94d8e1b4bf0040ull, 0x2a489cbe1cfbb6b8ull, 0x893cc664a19fcfedull, 0x0cf8594b7fcc657cull};
  fe d0, d1, r2, r3;
  memcpy(d0.l, w, 32);
  memcpy(d1.l, w + 4, 32);
  memcpy(r2.l, FR.r2, 32);
  memcpy(r3.l, R3L, 32);
  return f_add(&FR, f_mul(&FR, d0, r2), f_mul(&FR, d1, r3));
}
static void tr_common_scalar(transcript* t, fe s) {
  uint8_t tag = 2;
  b2b_update(&t->st, &tag, 1);
  fe c = f_to_canonical(&FR, s);
  b2b_update(&t->st, (uint8_t*)c.l, 32);
}
static int tr_write_point(transcript* t, const aff* p) {
  if (aff_is_id(p)) return -1;
  uint8_t tag = 1;
  b2b_update(&t->st, &tag, 1);
  fe x = f_to_canonical(Q, p->x), y = f_to_canonical(Q, p->y);
  b2b_update(&t->st, (uint8_t*)x.l, 32);
  b2b_update(&t->st, (uint8_t*)y.l, 32);
  uint8_t b[32];
  memcpy(b, x.l, 32);
  b[31] |= (uint8_t)((y.l[0] & 1) << 7); /* derive/curve.rs:635-646 */
  memcpy(t->proof + t->len, b, 32);
  t->len += 32;
  return 0;
}
static void tr_write_scalar(transcript* t, fe s) {
  tr_common_scalar(t, s);
  fe c = f_to_canonical(&FR, s);
  memcpy(t->proof + t->len, c.l, 32);
  t->len += 32;
}
static fe tr_squeeze(transcript* t) {
  uint8_t tag = 0, out[64];
  b2b_update(&t->st, &tag, 1);
  b2b_final_clone(&t->st, out);
  uint64_t w[8];
  memcpy(w, out, 64);
  return fr_from_u512(w);
}

/* harness RNG: xoshiro256** (same stream as sha2_on_cq_halo2_amd's cq_xoshiro256ss_next_u64) */
static inline uint64_t xs_next(uint64_t s[4]) {
  uint64_t result = ((s[1] * 5) << 7 | (s[1] * 5) >> 57) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
  s[3] = (s[3] << 45) | (s[3] >> 19);
  return result;
}
static fe fr_random(uint64_t s[4]) { /* bn256/fr.rs:159-170 */
  uint64_t w[8];
  for (int i = 0; i < 8; i++) w[i] = xs_next(s);
  return fr_from_u512(w);
}

/* field Ord = numeric order of the canonical value (derive/field.rs:128-141) */
typedef struct { fe canon; uint32_t idx; } keyed;
static int keyed_cmp(const void* a, const void* b) {
  const keyed *x = (const keyed*)a, *y = (const keyed*)b;
  for (int i = 3; i >= 0; i--) {
    if (x->canon.l[i] < y->canon.l[i]) return -1;
    if (x->canon.l[i] > y->canon.l[i]) return 1;
  }
  return 0;
}

static jac commit(const fe* poly, const aff* bases, size_t len) {
  jac r;
  cqo_best_multiexp((const uint64_t*)poly, (const uint64_t*)bases, len, (uint64_t*)&r);
  return r;
}
static jac aff_to_jac(const aff* a) {
  jac r = jac_id();
  if (!aff_is_id(a)) { r.x = a->x; r.y = a->y; r.z = f_one(Q); }
  return r;
}
static void lagrange_to_coeff(fe* a, uint32_t k, fe omega_inv, fe divisor) {
  cqo_ifft((uint64_t*)a, omega_inv.l, k, divisor.l);
}

typedef struct {
  uint32_t k, extended_k;
  fe omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;
  fe* t_evaluations;
  size_t t_len;
} domain_t;

/* poly/domain.rs:39-142 */
static void domain_new(domain_t* d, uint32_t j, uint32_t k) {
  static const uint64_t ROOT[4] = {0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull, 0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull};
  static const uint64_t ZETA[4] = {0xb8ca0b2d36636f23ull, 0xcc37a73fec2bc5e9ull, 0x048b6e193fd84104ull, 0x30644e72e131a029ull};
  fe r2;
  memcpy(r2.l, FR.r2, 32);
  uint64_t n = 1ull << k, qd = j - 1;
  uint32_t ek = k;
  while ((1ull << ek) < n * qd) ek++;
  d->k = k;
  d->extended_k = ek;
  fe w;
  memcpy(w.l, ROOT, 32);
  w = f_mul(&FR, w, r2);
  for (uint32_t i = ek; i < 28; i++) w = f_sqr(&FR, w);
  d->extended_omega = w;
  for (uint32_t i = k; i < ek; i++) w = f_sqr(&FR, w);
  d->omega = w;
  d->omega_inv = f_inv(&FR, d->omega);
  d->extended_omega_inv = f_inv(&FR, d->extended_omega);
  fe z;
  memcpy(z.l, ZETA, 32);
  d->g_coset = f_mul(&FR, z, r2);
  d->g_coset_inv = f_sqr(&FR, d->g_coset);
  d->t_len = (size_t)1 << (ek - k);
  d->t_evaluations = (fe*)malloc(d->t_len * sizeof(fe));
  uint64_t en[4] = {n, 0, 0, 0};
  fe cur = f_pow(&FR, d->g_coset, en), step = f_pow(&FR, d->extended_omega, en);
  for (size_t i = 0; i < d->t_len; i++) {
    d->t_evaluations[i] = f_inv(&FR, f_sub(&FR, cur, f_one(&FR)));
    cur = f_mul(&FR, cur, step);
  }
  d->ifft_divisor = f_inv(&FR, f_from_u64(&FR, n));
  d->extended_ifft_divisor = f_inv(&FR, f_from_u64(&FR, 1ull << ek));
}
/* domain.rs:252-266 */
static fe* coeff_to_extended(const domain_t* d, const fe* a) {
  size_t n = (size_t)1 << d->k, ext = (size_t)1 << d->extended_k;
  fe* v = (fe*)calloc(ext, sizeof(fe));
  memcpy(v, a, n * sizeof(fe));
  cqo_distribute_powers((uint64_t*)v, n, d->g_coset.l, d->g_coset_inv.l);
  cqo_best_fft((uint64_t*)v, d->extended_omega.l, d->extended_k);
  return v;
}

/* keygen.rs:344-373: l_active_row on the extended coset */
void cqo_keygen_l_active(uint32_t k, uint32_t blinding_factors, uint64_t* out_ext) {
  domain_t d;
  domain_new(&d, 3, k);
  size_t n = (size_t)1 << k, ext = (size_t)1 << d.extended_k;
  fe* lb = (fe*)calloc(n, sizeof(fe));
  fe* ll = (fe*)calloc(n, sizeof(fe));
  for (size_t i = n - blinding_factors; i < n; i++) lb[i] = f_one(&FR);
  ll[n - blinding_factors - 1] = f_one(&FR);
  lagrange_to_coeff(lb, k, d.omega_inv, d.ifft_divisor);
  lagrange_to_coeff(ll, k, d.omega_inv, d.ifft_divisor);
  fe *lbe = coeff_to_extended(&d, lb), *lle = coeff_to_extended(&d, ll);
  fe* out = (fe*)out_ext;
  for (size_t i = 0; i < ext; i++) out[i] = f_sub(&FR, f_one(&FR), f_add(&FR, lle[i], lbe[i]));
  free(lb); free(ll); free(lbe); free(lle); free(d.t_evaluations);
}

/* Returns 0 on success; negative on the panics / errors of the reference (lookup failure, identity point). */
int cqo_create_proof(uint32_t k, uint32_t num_advice, uint32_t num_lookups, const uint32_t* widths, const uint32_t* cols,
                     const uint32_t* table_ids, uint32_t num_tables, size_t N, const uint64_t* table_values_,
                     const uint64_t* table_qs_, const uint64_t* g_, const uint64_t* g_lagrange_, const uint64_t* t_g1_lagrange_,
                     const uint64_t* t_open0_, const uint64_t* b0_bound_, const uint64_t* l_active_, const uint64_t vk_repr_[4],
                     const uint64_t* advice_, uint64_t rng[4], uint8_t* proof_out, size_t* proof_len) {
  const size_t n = (size_t)1 << k;
  const aff *g = (const aff*)g_, *g_lagrange = (const aff*)g_lagrange_, *t_lag = (const aff*)t_g1_lagrange_,
            *t_open0 = (const aff*)t_open0_, *b0_bound = (const aff*)b0_bound_, *table_qs = (const aff*)table_qs_;
  const fe *table_values = (const fe*)table_values_, *l_active = (const fe*)l_active_;
  domain_t dom;
  domain_new(&dom, 3, k);
  const size_t ext = (size_t)1 << dom.extended_k;
  /* advice queries (circuit.rs:1619-1633) and blinding_factors (:2022-2047) */
  uint32_t* per_col = (uint32_t*)calloc(num_advice ? num_advice : 1, sizeof(uint32_t));
  uint32_t* aq = (uint32_t*)malloc((num_advice + 1) * sizeof(uint32_t));
  uint32_t naq = 0;
  {
    size_t off = 0;
    for (uint32_t l = 0; l < num_lookups; l++)
      for (uint32_t j = 0; j < widths[l]; j++, off++) {
        uint32_t c = cols[off], seen = 0;
        for (uint32_t q = 0; q < naq; q++) seen |= (aq[q] == c);
        if (!seen) { aq[naq++] = c; per_col[c]++; }
      }
  }
  uint32_t factors = 1;
  for (uint32_t c = 0; c < num_advice; c++) if (per_col[c] > factors) factors = per_col[c];
  if (factors < 3) factors = 3;
  const uint32_t bf = factors + 2;
  const size_t u = n - (bf + 1);
  transcript tr;
  b2b_init(&tr.st, "Halo2-Transcript");
  tr.proof = proof_out;
  tr.len = 0;
  fe vk_repr;
  memcpy(vk_repr.l, vk_repr_, 32);
  tr_common_scalar(&tr, vk_repr);

  /* ---- advice (prover.rs:299-391) ---- */
  fe* advice = (fe*)calloc((size_t)num_advice * n + 1, sizeof(fe));
  for (uint32_t c = 0; c < num_advice; c++) memcpy(advice + (size_t)c * n, (const fe*)advice_ + (size_t)c * n, u * sizeof(fe));
  for (uint32_t c = 0; c < num_advice; c++)
    for (size_t r = u; r < n; r++) advice[(size_t)c * n + r] = fr_random(rng);
  for (uint32_t c = 0; c < num_advice; c++) (void)fr_random(rng);
  for (uint32_t c = 0; c < num_advice; c++) {
    jac cm = commit(advice + (size_t)c * n, g_lagrange, n);
    aff a = jac_to_aff(&cm);
    if (tr_write_point(&tr, &a)) return -6;
  }
  const fe theta = tr_squeeze(&tr);

  /* ---- CQ round 1 (static_lookup/prover.rs:51-183) ---- */
  fe* f_all = (fe*)calloc((size_t)num_lookups * n + 1, sizeof(fe));
  uint32_t* m_all = (uint32_t*)calloc((size_t)num_lookups * N + 1, sizeof(uint32_t));
  /* value -> index maps (BTreeMap<Scalar, usize>, static_lookup.rs:82-83): sorted by canonical value */
  keyed** maps = (keyed**)malloc(num_tables * sizeof(keyed*));
  for (uint32_t t = 0; t < num_tables; t++) {
    maps[t] = (keyed*)malloc(N * sizeof(keyed));
    for (size_t i = 0; i < N; i++) { maps[t][i].canon = f_to_canonical(&FR, table_values[(size_t)t * N + i]); maps[t][i].idx = (uint32_t)i; }
    qsort(maps[t], N, sizeof(keyed), keyed_cmp);
  }
  {
    size_t off = 0;
    for (uint32_t l = 0; l < num_lookups; l++) {
      const uint32_t w = widths[l];
      fe* f = f_all + (size_t)l * n;
      for (uint32_t j = 0; j < w; j++) { /* acc * theta + expression (:108-116) */
        const fe* e = advice + (size_t)cols[off + j] * n;
#pragma omp parallel for
        for (long i = 0; i < (long)n; i++) f[i] = f_add(&FR, f_mul(&FR, f[i], theta), e[i]);
      }
      for (size_t row = 0; row < u; row++) { /* :132-161, serial */
        uint32_t idx = 0xffffffffu;
        for (uint32_t j = 0; j < w; j++) {
          keyed key;
          key.canon = f_to_canonical(&FR, advice[(size_t)cols[off + j] * n + row]);
          keyed* hit = (keyed*)bsearch(&key, maps[table_ids[off + j]], N, sizeof(keyed), keyed_cmp);
          if (!hit) return -4;
          if (j && hit->idx != idx) return -4;
          idx = hit->idx;
        }
        m_all[(size_t)l * N + idx]++;
      }
      jac f_cm = commit(f, g_lagrange, n);
      jac m_cm = jac_id();
      for (size_t i = 0; i < N; i++) /* :167-170 serial double-and-add, ascending index */
        if (m_all[(size_t)l * N + i]) {
          jac base = aff_to_jac(&t_lag[i]);
          jac term = jac_mul(&base, f_from_u64(&FR, m_all[(size_t)l * N + i]));
          m_cm = jac_add(&term, &m_cm);
        }
      aff a1 = jac_to_aff(&f_cm), a2 = jac_to_aff(&m_cm);
      if (tr_write_point(&tr, &a1) || tr_write_point(&tr, &a2)) return -6;
      off += w;
    }
  }
  const fe beta = tr_squeeze(&tr);
  (void)tr_squeeze(&tr); /* gamma */
  const fe beta_inv = f_inv(&FR, beta);

  /* ---- CQ round 2 (static_lookup/prover.rs:187-342) ---- */
  fe* b_all = (fe*)calloc((size_t)num_lookups * n + 1, sizeof(fe));
  fe* fc_all = (fe*)calloc((size_t)num_lookups * n + 1, sizeof(fe));
  fe* a_at_zero = (fe*)calloc(num_lookups + 1, sizeof(fe));
  {
    size_t off = 0;
    for (uint32_t l = 0; l < num_lookups; l++) {
      const uint32_t w = widths[l];
      const fe* f = f_all + (size_t)l * n;
      /* f_set: BTreeSet of all f (:242) -- ordered insert of n elements */
      keyed* fset = (keyed*)malloc(n * sizeof(keyed));
      for (size_t i = 0; i < n; i++) { fset[i].canon = f_to_canonical(&FR, f[i]); fset[i].idx = (uint32_t)i; }
      qsort(fset, n, sizeof(keyed), keyed_cmp);
      jac a_cm = jac_id(), qa_cm = jac_id(), a0_cm = jac_id();
      for (size_t i = 0; i < N; i++) { /* :245-257 */
        const uint32_t mult = m_all[(size_t)l * N + i];
        if (!mult) continue;
        fe tv = f_zero();
        aff tq;
        memset(&tq, 0, sizeof tq);
        for (uint32_t j = 0; j < w; j++) { /* compress_tables :224-240 */
          const uint32_t tid = table_ids[off + j];
          tv = f_add(&FR, f_mul(&FR, tv, theta), table_values[(size_t)tid * N + i]);
          jac tqj = aff_to_jac(&tq);
          jac scaled = jac_mul(&tqj, theta);
          jac sum = jac_add_aff(&scaled, &table_qs[(size_t)tid * N + i]);
          tq = jac_to_aff(&sum);
        }
        fe a_i = f_mul(&FR, f_from_u64(&FR, mult), f_inv(&FR, f_add(&FR, tv, beta)));
        keyed key;
        key.canon = f_to_canonical(&FR, tv);
        if (!bsearch(&key, fset, n, sizeof(keyed), keyed_cmp)) return -5; /* sanity :250 */
        jac b1 = aff_to_jac(&t_lag[i]), b2 = aff_to_jac(&tq), b3 = aff_to_jac(&t_open0[i]);
        jac t1 = jac_mul(&b1, a_i), t2 = jac_mul(&b2, a_i), t3 = jac_mul(&b3, a_i);
        a_cm = jac_add(&t1, &a_cm);
        qa_cm = jac_add(&t2, &qa_cm);
        a0_cm = jac_add(&t3, &a0_cm);
      }
      free(fset);
      fe* bs = b_all + (size_t)l * n;
      for (size_t i = 0; i < u; i++) bs[i] = f_inv(&FR, f_add(&FR, f[i], beta)); /* :261-266: one inversion per row, serial */
      for (size_t i = u; i < n; i++) bs[i] = beta_inv;
      lagrange_to_coeff(bs, k, dom.omega_inv, dom.ifft_divisor);
      jac p_cm = commit(bs + 1, b0_bound, n - 1); /* :299 */
      aff pa = jac_to_aff(&a_cm), pq = jac_to_aff(&qa_cm), p0 = jac_to_aff(&a0_cm);
      if (tr_write_point(&tr, &pa) || tr_write_point(&tr, &pq) || tr_write_point(&tr, &p0)) return -6;
      fe* b0 = (fe*)calloc(n, sizeof(fe));
      memcpy(b0, bs + 1, (n - 1) * sizeof(fe));
      jac b0_cm = commit(b0, g, n); /* :310 */
      free(b0);
      aff pb = jac_to_aff(&b0_cm), pp = jac_to_aff(&p_cm);
      if (tr_write_point(&tr, &pb) || tr_write_point(&tr, &pp)) return -6;
      fe t = f_sub(&FR, f_mul(&FR, bs[0], f_from_u64(&FR, n)), f_mul(&FR, f_from_u64(&FR, bf + 1), beta_inv));
      a_at_zero[l] = f_mul(&FR, t, f_inv(&FR, f_from_u64(&FR, N))); /* :318-324 */
      fe* fc = fc_all + (size_t)l * n;
      memcpy(fc, f, n * sizeof(fe));
      lagrange_to_coeff(fc, k, dom.omega_inv, dom.ifft_divisor); /* :326-334 */
      off += w;
    }
  }

  /* ---- vanishing commit (vanishing/prover.rs:37-65) ---- */
  fe* random_poly = (fe*)malloc(n * sizeof(fe));
  for (size_t i = 0; i < n; i++) random_poly[i] = fr_random(rng);
  (void)fr_random(rng);
  {
    jac c = commit(random_poly, g, n);
    aff a = jac_to_aff(&c);
    if (tr_write_point(&tr, &a)) return -6;
  }
  const fe y = tr_squeeze(&tr);
  for (uint32_t c = 0; c < num_advice; c++) lagrange_to_coeff(advice + (size_t)c * n, k, dom.omega_inv, dom.ifft_divisor); /* prover.rs:587-603 */

  /* ---- evaluate_h (evaluation.rs:285-551) ---- */
  for (uint32_t c = 0; c < num_advice; c++) { /* :317-325: advice cosets are computed although no CQ term reads them */
    fe* cs = coeff_to_extended(&dom, advice + (size_t)c * n);
    free(cs);
  }
  fe* h = (fe*)calloc(ext, sizeof(fe));
  for (uint32_t l = 0; l < num_lookups; l++) { /* :533-548 */
    fe *bc = coeff_to_extended(&dom, b_all + (size_t)l * n), *fcs = coeff_to_extended(&dom, fc_all + (size_t)l * n);
    cqo_cq_quotient_term((uint64_t*)h, (uint64_t*)bc, (uint64_t*)fcs, (const uint64_t*)l_active, ext, y.l, beta.l);
    free(bc);
    free(fcs);
  }
  /* ---- vanishing construct (vanishing/prover.rs:69-120) ---- */
  cqo_mul_periodic((uint64_t*)h, ext, (uint64_t*)dom.t_evaluations, dom.t_len);
  cqo_ifft((uint64_t*)h, dom.extended_omega_inv.l, dom.extended_k, dom.extended_ifft_divisor.l);
  cqo_distribute_powers((uint64_t*)h, ext, dom.g_coset_inv.l, dom.g_coset.l);
  const size_t pieces = 2; /* n * (degree - 1) / n */
  for (size_t i = 0; i < pieces; i++) (void)fr_random(rng);
  for (size_t i = 0; i < pieces; i++) {
    jac c = commit(h + i * n, g, n);
    aff a = jac_to_aff(&c);
    if (tr_write_point(&tr, &a)) return -6;
  }
  const fe x = tr_squeeze(&tr);
  uint64_t en[4] = {n, 0, 0, 0};
  const fe xn = f_pow(&FR, x, en);

  /* ---- evaluations (prover.rs:654-719) ---- */
  fe* adv_evals = (fe*)malloc((naq + 1) * sizeof(fe));
  for (uint32_t q = 0; q < naq; q++) {
    cqo_eval_polynomial((uint64_t*)(advice + (size_t)aq[q] * n), n, x.l, adv_evals[q].l);
    tr_write_scalar(&tr, adv_evals[q]);
  }
  fe* h_poly = (fe*)calloc(n, sizeof(fe)); /* vanishing/prover.rs:131-135 */
  for (size_t i = pieces; i-- > 0;)
    for (size_t r = 0; r < n; r++) h_poly[r] = f_add(&FR, f_mul(&FR, h_poly[r], xn), h[i * n + r]);
  fe random_eval;
  cqo_eval_polynomial((uint64_t*)random_poly, n, x.l, random_eval.l);
  tr_write_scalar(&tr, random_eval);
  fe* b0_polys = (fe*)calloc((size_t)num_lookups * n + 1, sizeof(fe));
  for (uint32_t l = 0; l < num_lookups; l++) { /* static_lookup/prover.rs:360-370 */
    memcpy(b0_polys + (size_t)l * n, b_all + (size_t)l * n + 1, (n - 1) * sizeof(fe));
    fe e1, e2;
    cqo_eval_polynomial((uint64_t*)(b0_polys + (size_t)l * n), n, x.l, e1.l);
    cqo_eval_polynomial((uint64_t*)(fc_all + (size_t)l * n), n, x.l, e2.l);
    tr_write_scalar(&tr, e1);
    tr_write_scalar(&tr, e2);
    tr_write_scalar(&tr, a_at_zero[l]);
  }

  /* ---- multiopen GWC (gwc/prover.rs:42-91): one point group (every query at x) ---- */
  {
    const fe v = tr_squeeze(&tr);
    const size_t nq = naq + 2 * num_lookups + 2;
    const fe** polys = (const fe**)malloc(nq * sizeof(fe*));
    size_t qi = 0;
    for (uint32_t q = 0; q < naq; q++) polys[qi++] = advice + (size_t)aq[q] * n;
    for (uint32_t l = 0; l < num_lookups; l++) { polys[qi++] = b0_polys + (size_t)l * n; polys[qi++] = fc_all + (size_t)l * n; }
    polys[qi++] = h_poly;
    polys[qi++] = random_poly;
    fe* batch = (fe*)calloc(n, sizeof(fe));
    fe eval_batch = f_zero(), pv = f_one(&FR);
    for (size_t q = 0; q < nq; q++) {
      fe ev;
      cqo_eval_polynomial((const uint64_t*)polys[q], n, x.l, ev.l); /* get_eval (query.rs:51-53) */
      const fe* p = polys[q];
#pragma omp parallel for
      for (long r = 0; r < (long)n; r++) batch[r] = f_add(&FR, batch[r], f_mul(&FR, p[r], pv));
      eval_batch = f_add(&FR, eval_batch, f_mul(&FR, ev, pv));
      pv = f_mul(&FR, pv, v);
    }
    batch[0] = f_sub(&FR, batch[0], eval_batch);
    fe* wit = (fe*)malloc(n * sizeof(fe));
    cqo_kate_division((uint64_t*)batch, n, x.l, (uint64_t*)wit);
    { /* KATE SANITY CHECK (arithmetic.rs:370-384) */
      fe ev;
      cqo_eval_polynomial((uint64_t*)batch, n, x.l, ev.l);
      fe nb = f_neg(&FR, x);
      for (size_t r = 0; r < n; r++) {
        fe lhs = f_add(&FR, r < n - 1 ? f_mul(&FR, wit[r], nb) : f_zero(), r ? wit[r - 1] : f_zero());
        fe rhs = r == 0 ? f_sub(&FR, batch[0], ev) : batch[r];
        if (!f_eq(lhs, rhs)) return -5;
      }
    }
    jac c = commit(wit, g, n - 1);
    aff a = jac_to_aff(&c);
    if (tr_write_point(&tr, &a)) return -6;
    free(polys); free(batch); free(wit);
  }
  *proof_len = tr.len;
  for (uint32_t t = 0; t < num_tables; t++) free(maps[t]);
  free(maps); free(per_col); free(aq); free(advice); free(f_all); free(m_all); free(b_all); free(fc_all); free(a_at_zero);
  free(random_poly); free(h); free(adv_evals); free(h_poly); free(b0_polys); free(dom.t_evaluations);
  return 0;
}

/* ================================================================================================
 * BN254 optimal ate pairing -- the acceptance oracle's closing equations with REAL pairings at any k
 * (plonk/verifier.rs, static_lookup/verifier.rs:138-177, gwc/verifier.rs:76-128 end in pairing
 * checks over halo2curves::bn256::Bn256, arithmetic/curves/src/bn256/engine.rs).
 *
 * This restates the PAIRING, in the formulation oracle/pairing.py already pins against the reference's constants
 * and the properties its tests assert (engine.rs:662-762): Fq2 = Fq[i]/(i^2+1); Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)
 * with i = w^6 - 9 (xi = 9 + i = w^6); G2 untwisted into E(Fq12) by (x, y) -> (x w^2, y w^3); Miller loop over
 * 6x+2, x = BN_X (engine.rs:18), plus the two Frobenius lines (engine.rs:430-441); final exponentiation as the plain
 * power (q^12 - 1)/r.  Not the reference's tower arithmetic -- every use is "product of pairings == 1", invariant
 * under the normalisation -- and not fast (about 20 ms per Miller loop), but ~100x the Python restatement, which
 * tests/test_oracle_pairing.py compares it with value for value.
 * ============================================================================================== */
typedef struct { fe c[12]; } fq12;
typedef struct { fe a, b; } fq2; /* a + b i */

static fq12 fq12_zero(void) { fq12 r; memset(&r, 0, sizeof r); return r; }
static fq12 fq12_one(void) { fq12 r = fq12_zero(); r.c[0] = f_one(Q); return r; }
static int fq12_eq(const fq12* x, const fq12* y) { return memcmp(x, y, sizeof *x) == 0; } /* canonical Montgomery limbs */
static int fq12_is_zero(const fq12* x) { fq12 z = fq12_zero(); return fq12_eq(x, &z); }
static fq12 fq12_add(const fq12* x, const fq12* y) { fq12 r; for (int i = 0; i < 12; i++) r.c[i] = f_add(Q, x->c[i], y->c[i]); return r; }
static fq12 fq12_sub(const fq12* x, const fq12* y) { fq12 r; for (int i = 0; i < 12; i++) r.c[i] = f_sub(Q, x->c[i], y->c[i]); return r; }
static fq12 fq12_neg(const fq12* x) { fq12 r; for (int i = 0; i < 12; i++) r.c[i] = f_neg(Q, x->c[i]); return r; }
static fq12 fq12_from_fq(fe v) { fq12 r = fq12_zero(); r.c[0] = v; return r; }
/* schoolbook product, then w^e = 18 w^(e-6) - 82 w^(e-12) from the top down */
static fq12 fq12_mul(const fq12* x, const fq12* y) {
  fe b[23];
  for (int i = 0; i < 23; i++) b[i] = f_zero();
  for (int i = 0; i < 12; i++) {
    if (f_is_zero(x->c[i])) continue;
    for (int j = 0; j < 12; j++) b[i + j] = f_add(Q, b[i + j], f_mul(Q, x->c[i], y->c[j]));
  }
  const fe c18 = f_from_u64(Q, 18), c82 = f_from_u64(Q, 82);
  for (int e = 22; e >= 12; e--) {
    if (f_is_zero(b[e])) continue;
    b[e - 6] = f_add(Q, b[e - 6], f_mul(Q, b[e], c18));
    b[e - 12] = f_sub(Q, b[e - 12], f_mul(Q, b[e], c82));
  }
  fq12 r;
  memcpy(r.c, b, sizeof r.c);
  return r;
}
static fq12 fq12_pow(const fq12* x, const uint64_t* e, int words) {
  fq12 acc = fq12_one();
  int started = 0;
  for (int w = words - 1; w >= 0; w--)
    for (int bit = 63; bit >= 0; bit--) {
      if (started) acc = fq12_mul(&acc, &acc);
      if ((e[w] >> bit) & 1) {
        acc = fq12_mul(&acc, x);
        started = 1;
      }
    }
  return acc;
}
/* inverse: the multiplication-by-x map is linear over Fq; solve M v = 1 by Gauss-Jordan (12 x 12) */
static fq12 fq12_inv(const fq12* x) {
  fe m[12][13];
  fq12 col = *x; /* x * w^j */
  fq12 wj = fq12_zero();
  wj.c[1] = f_one(Q);
  for (int j = 0; j < 12; j++) {
    for (int i = 0; i < 12; i++) m[i][j] = col.c[i];
    col = fq12_mul(&col, &wj);
  }
  for (int i = 0; i < 12; i++) m[i][12] = i == 0 ? f_one(Q) : f_zero();
  for (int p = 0; p < 12; p++) {
    int piv = p;
    while (piv < 12 && f_is_zero(m[piv][p])) piv++;
    if (piv == 12) return fq12_zero(); /* not invertible (x = 0) */
    if (piv != p)
      for (int j = 0; j < 13; j++) { fe t = m[p][j]; m[p][j] = m[piv][j]; m[piv][j] = t; }
    const fe inv = f_inv(Q, m[p][p]);
    for (int j = p; j < 13; j++) m[p][j] = f_mul(Q, m[p][j], inv);
    for (int i = 0; i < 12; i++) {
      if (i == p || f_is_zero(m[i][p])) continue;
      const fe fct = m[i][p];
      for (int j = p; j < 13; j++) m[i][j] = f_sub(Q, m[i][j], f_mul(Q, fct, m[p][j]));
    }
  }
  fq12 r;
  for (int i = 0; i < 12; i++) r.c[i] = m[i][12];
  return r;
}

/* ---- Fq2 and G2 = E'(Fq2): y^2 = x^3 + 3/(9+i) (bn256/curve.rs:85-129) ---- */
static fq2 fq2_add(fq2 x, fq2 y) { fq2 r = {f_add(Q, x.a, y.a), f_add(Q, x.b, y.b)}; return r; }
static fq2 fq2_sub(fq2 x, fq2 y) { fq2 r = {f_sub(Q, x.a, y.a), f_sub(Q, x.b, y.b)}; return r; }
static fq2 fq2_mul(fq2 x, fq2 y) {
  fq2 r = {f_sub(Q, f_mul(Q, x.a, y.a), f_mul(Q, x.b, y.b)), f_add(Q, f_mul(Q, x.a, y.b), f_mul(Q, x.b, y.a))};
  return r;
}
static fq2 fq2_inv(fq2 x) { /* conj / norm */
  const fe nrm = f_inv(Q, f_add(Q, f_sqr(Q, x.a), f_sqr(Q, x.b)));
  fq2 r = {f_mul(Q, x.a, nrm), f_neg(Q, f_mul(Q, x.b, nrm))};
  return r;
}
static int fq2_eq(fq2 x, fq2 y) { return f_eq(x.a, y.a) && f_eq(x.b, y.b); }
typedef struct { fq2 x, y; int inf; } g2aff;
static const uint64_t G2_GEN_RAW[4][4] = { /* curve.rs:100-129, canonical limbs: x.c0, x.c1, y.c0, y.c1 */
    {0x46DEBD5CD992F6EDull, 0x674322D4F75EDADDull, 0x426A00665E5C4479ull, 0x1800DEEF121F1E76ull},
    {0x97E485B7AEF312C2ull, 0xF1AA493335A9E712ull, 0x7260BFB731FB5D25ull, 0x198E9393920D483Aull},
    {0x4CE6CC0166FA7DAAull, 0xE3D1E7690C43D37Bull, 0x4AAB71808DCB408Full, 0x12C85EA5DB8C6DEBull},
    {0x55ACDADCD122975Bull, 0xBC4B313370B38EF3ull, 0xEC9E99AD690C3395ull, 0x090689D0585FF075ull}};
static fe fq_from_canonical(const uint64_t l[4]) { fe a, r2; memcpy(a.l, l, 32); memcpy(r2.l, FQ.r2, 32); return f_mul(Q, a, r2); }
static g2aff g2_generator(void) {
  g2aff g;
  g.x.a = fq_from_canonical(G2_GEN_RAW[0]); g.x.b = fq_from_canonical(G2_GEN_RAW[1]);
  g.y.a = fq_from_canonical(G2_GEN_RAW[2]); g.y.b = fq_from_canonical(G2_GEN_RAW[3]);
  g.inf = 0;
  return g;
}
static g2aff g2_double(g2aff p) {
  if (p.inf) return p;
  const fq2 x2 = fq2_mul(p.x, p.x);
  const fq2 m = fq2_mul(fq2_add(fq2_add(x2, x2), x2), fq2_inv(fq2_add(p.y, p.y)));
  g2aff r;
  r.x = fq2_sub(fq2_mul(m, m), fq2_add(p.x, p.x));
  r.y = fq2_sub(fq2_mul(m, fq2_sub(p.x, r.x)), p.y);
  r.inf = 0;
  return r;
}
static g2aff g2_add(g2aff p, g2aff q) {
  if (p.inf) return q;
  if (q.inf) return p;
  if (fq2_eq(p.x, q.x)) {
    if (fq2_eq(p.y, q.y)) return g2_double(p);
    g2aff o = p; o.inf = 1; return o;
  }
  const fq2 m = fq2_mul(fq2_sub(q.y, p.y), fq2_inv(fq2_sub(q.x, p.x)));
  g2aff r;
  r.x = fq2_sub(fq2_sub(fq2_mul(m, m), p.x), q.x);
  r.y = fq2_sub(fq2_mul(m, fq2_sub(p.x, r.x)), p.y);
  r.inf = 0;
  return r;
}
/* [k]_2, k canonical (kzg/commitment.rs:253-256 computes its G2 elements this way) */
static g2aff g2_mul_gen(const uint64_t k[4]) {
  g2aff acc = g2_generator(), base = acc;
  acc.inf = 1;
  for (int w = 0; w < 4; w++)
    for (int b = 0; b < 64; b++) {
      if ((k[w] >> b) & 1) acc = g2_add(acc, base);
      base = g2_double(base);
    }
  return acc;
}

/* ---- E(Fq12), affine; `inf` marks the identity ---- */
typedef struct { fq12 x, y; int inf; } e12;
/* (x, y) -> (x w^2, y w^3), Fq2 embedded through i = w^6 - 9 */
static e12 twist(g2aff p) {
  e12 r;
  r.inf = p.inf;
  r.x = fq12_zero(); r.y = fq12_zero();
  if (p.inf) return r;
  const fe nine = f_from_u64(Q, 9);
  /* x = (xa - 9 xb) + xb w^6, times w^2 */
  r.x.c[2] = f_sub(Q, p.x.a, f_mul(Q, nine, p.x.b));
  r.x.c[8] = p.x.b;
  r.y.c[3] = f_sub(Q, p.y.a, f_mul(Q, nine, p.y.b));
  r.y.c[9] = p.y.b;
  return r;
}
/* one Miller step: line through r and s (tangent when equal) evaluated at (xt, yt), and r <- r + s; the slope's
 * inversion is shared by the line and the point */
static fq12 line_and_add(e12* r, const e12* s, const fq12* xt, const fq12* yt) {
  fq12 m;
  if (!fq12_eq(&r->x, &s->x)) {
    fq12 dy = fq12_sub(&s->y, &r->y), dx = fq12_sub(&s->x, &r->x), dxi = fq12_inv(&dx);
    m = fq12_mul(&dy, &dxi);
  } else if (fq12_eq(&r->y, &s->y)) {
    fq12 x2 = fq12_mul(&r->x, &r->x), x3 = fq12_add(&x2, &x2);
    x3 = fq12_add(&x3, &x2);
    fq12 y2 = fq12_add(&r->y, &r->y), y2i = fq12_inv(&y2);
    m = fq12_mul(&x3, &y2i);
  } else { /* vertical line: xt - x1; the sum is the identity */
    fq12 l = fq12_sub(xt, &r->x);
    r->inf = 1;
    return l;
  }
  fq12 t1 = fq12_sub(xt, &r->x), t2 = fq12_sub(yt, &r->y);
  fq12 l = fq12_mul(&m, &t1);
  l = fq12_sub(&l, &t2);
  fq12 nx = fq12_mul(&m, &m);
  nx = fq12_sub(&nx, &r->x);
  nx = fq12_sub(&nx, &s->x);
  fq12 t3 = fq12_sub(&r->x, &nx), ny = fq12_mul(&m, &t3);
  ny = fq12_sub(&ny, &r->y);
  r->x = nx;
  r->y = ny;
  return l;
}
static const uint64_t ATE_LOOP[2] = {0x9d797039be763ba8ull, 0x1ull}; /* 6 * 4965661367192848881 + 2 = 29793968203157093288 */
static fq12 miller_loop(g2aff q2, const aff* p1) {
  if (q2.inf || aff_is_id(p1)) return fq12_one();
  const e12 qq = twist(q2);
  const fq12 xt = fq12_from_fq(p1->x), yt = fq12_from_fq(p1->y);
  e12 r = qq;
  fq12 f = fq12_one();
  for (int i = 63; i >= 0; i--) { /* bit 64 is the leading one */
    f = fq12_mul(&f, &f);
    fq12 l = line_and_add(&r, &r, &xt, &yt);
    f = fq12_mul(&f, &l);
    if ((ATE_LOOP[0] >> i) & 1) {
      l = line_and_add(&r, &qq, &xt, &yt);
      f = fq12_mul(&f, &l);
    }
  }
  /* Q1 = frobenius(Q), -Q2 = -frobenius^2(Q) as points of E(Fq12): coordinate-wise q-th powers */
  e12 q1, nq2;
  q1.inf = nq2.inf = 0;
  q1.x = fq12_pow(&qq.x, FQ.mod, 4);
  q1.y = fq12_pow(&qq.y, FQ.mod, 4);
  nq2.x = fq12_pow(&q1.x, FQ.mod, 4);
  nq2.y = fq12_pow(&q1.y, FQ.mod, 4);
  nq2.y = fq12_neg(&nq2.y);
  fq12 l = line_and_add(&r, &q1, &xt, &yt);
  f = fq12_mul(&f, &l);
  e12 r2 = r;
  l = line_and_add(&r2, &nq2, &xt, &yt);
  f = fq12_mul(&f, &l);
  return f;
}
/* (q^12 - 1) / r as 64-bit words, least significant first: computed once by long arithmetic */
static uint64_t FINAL_EXP[48];
static int FINAL_EXP_WORDS = 0;
static void final_exp_init(void) {
  if (FINAL_EXP_WORDS) return;
  /* q^12 by repeated multiplication on a 48-word integer */
  uint64_t acc[49] = {1}, tmp[49];
  for (int it = 0; it < 12; it++) {
    memset(tmp, 0, sizeof tmp);
    for (int i = 0; i < 48; i++) {
      if (!acc[i]) continue;
      uint64_t carry = 0;
      for (int j = 0; j < 4 && i + j < 49; j++) {
        u128 t = (u128)acc[i] * FQ.mod[j] + tmp[i + j] + carry;
        tmp[i + j] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
      for (int j = i + 4; carry && j < 49; j++) {
        u128 t = (u128)tmp[j] + carry;
        tmp[j] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
    }
    memcpy(acc, tmp, sizeof acc);
  }
  /* minus one (q^12 is odd), then long division by r, most significant word first */
  acc[0] -= 1;
  uint64_t quo[48];
  uint64_t rem[5] = {0, 0, 0, 0, 0};
  for (int w = 47; w >= 0; w--)
    for (int b = 63; b >= 0; b--) {
      /* rem = rem * 2 + bit */
      uint64_t c = (acc[w] >> b) & 1;
      for (int i = 0; i < 5; i++) { uint64_t n = (rem[i] << 1) | c; c = rem[i] >> 63; rem[i] = n; }
      /* if rem >= r: rem -= r, quotient bit 1 */
      int ge = rem[4] != 0;
      if (!ge) {
        ge = 1;
        for (int i = 3; i >= 0; i--)
          if (rem[i] != FR.mod[i]) { ge = rem[i] > FR.mod[i]; break; }
      }
      if (ge) {
        uint64_t borrow = 0;
        for (int i = 0; i < 4; i++) {
          u128 t = (u128)rem[i] - FR.mod[i] - borrow;
          rem[i] = (uint64_t)t;
          borrow = (uint64_t)(t >> 64) & 1;
        }
        rem[4] -= borrow;
        quo[w] = (b == 63 ? 0 : quo[w]) | ((uint64_t)1 << b);
      } else if (b == 63) {
        quo[w] = 0;
      }
    }
  memcpy(FINAL_EXP, quo, sizeof quo);
  FINAL_EXP_WORDS = 48;
}
static fq12 final_exponentiation(const fq12* f) {
  final_exp_init();
  return fq12_pow(f, FINAL_EXP, FINAL_EXP_WORDS);
}

/* e(P_i, [k_i]_2) for i < m, multiplied together, after ONE final exponentiation (`multi_miller_loop`,
 * engine.rs:571-640): g1_affine = m x 8 Montgomery limbs ((0,0) = identity), g2_scalars = m x 4 canonical limbs.
 * out12 (optional): the 12 coefficients of the result, Montgomery limbs.  Returns 1 iff the product is one. */
int cqo_pairing_product(const uint64_t* g1_affine, const uint64_t* g2_scalars, size_t m, uint64_t* out12) {
  fq12 f = fq12_one();
  fq12* parts = (fq12*)malloc(sizeof(fq12) * (m ? m : 1));
#pragma omp parallel for schedule(dynamic)
  for (size_t i = 0; i < m; i++) {
    aff p;
    memcpy(&p, g1_affine + 8 * i, sizeof p);
    parts[i] = miller_loop(g2_mul_gen(g2_scalars + 4 * i), &p);
  }
  for (size_t i = 0; i < m; i++) f = fq12_mul(&f, &parts[i]);
  free(parts);
  const fq12 r = final_exponentiation(&f), one = fq12_one();
  if (out12) memcpy(out12, r.c, sizeof r.c);
  return fq12_eq(&r, &one) && !fq12_is_zero(&f);
}
/* [k]_2 in affine coordinates (x.c0, x.c1, y.c0, y.c1: 4 x 4 Montgomery limbs); returns 0 for the identity */
int cqo_g2_mul(const uint64_t k[4], uint64_t out[16]) {
  const g2aff p = g2_mul_gen(k);
  if (p.inf) { memset(out, 0, 128); return 0; }
  memcpy(out, p.x.a.l, 32); memcpy(out + 4, p.x.b.l, 32); memcpy(out + 8, p.y.a.l, 32); memcpy(out + 12, p.y.b.l, 32);
  return 1;
}
