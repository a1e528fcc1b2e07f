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
