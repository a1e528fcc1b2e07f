// Host orchestrator: `create_proof` for CQ-only circuits (halo2_proofs/src/plonk/prover.rs:51-779),
// with the sub-arguments of plonk/static_lookup/prover.rs, plonk/vanishing/prover.rs,
// plonk/evaluation.rs:533-548 and poly/kzg/multiopen/gwc/prover.rs.  Everything O(n) runs on the GPU
// (polynomials never leave HBM); the host keeps the Fiat-Shamir transcript (transcript.rs:170-241),
// draws the blinding scalars from the caller's RNG in the reference's order, folds MSM window sums
// and normalises the handful of commitment points.  Call order, RNG order and transcript order are
// the contract: the proof bytes equal the reference's for
// the same (pk, witness, RNG stream).
//
// Scope: constraint systems with advice columns and static lookups whose inputs are
// `advice[col] @ Rotation::cur()` -- the shape of the reference's one CQ test (tests/my_test.rs).
// Gates, fixed/instance columns, permutations and legacy lookups are "next" rows.
// One deliberate omission: evaluation.rs:317-325 also transforms every advice polynomial to the
// extended coset although no CQ-only term reads them; that dead work is not reproduced.
#include <algorithm>
#include <cstring>
#include <vector>
#include "blake2b.hpp"
#include "cq.hpp"
#include "ctx.hpp"
#include "msm.hpp"
#include "prover.hpp"

using namespace cq;

namespace {

struct Transcript {
  Blake2b st;
  std::vector<uint8_t> proof;
  Transcript() { st.init(64, "Halo2-Transcript"); }
  void common_scalar(const Fr& s) {
    const uint8_t tag = 2;
    st.update(&tag, 1);
    U256 c = s.to_canonical();
    st.update((const uint8_t*)c.l, 32);
  }
  // transcript.rs:221-233; identity is an error
  bool common_point(const G1Affine& p) {
    if (p.is_identity()) return false;
    const uint8_t tag = 1;
    st.update(&tag, 1);
    U256 x = p.x.to_canonical(), y = p.y.to_canonical();
    st.update((const uint8_t*)x.l, 32);
    st.update((const uint8_t*)y.l, 32);
    return true;
  }
  bool write_point(const G1Affine& p) {
    if (!common_point(p)) return false;
    U256 x = p.x.to_canonical(), y = p.y.to_canonical();
    uint8_t b[32];
    memcpy(b, x.l, 32);
    b[31] |= (uint8_t)((y.l[0] & 1) << 7);  // derive/curve.rs:635-646
    proof.insert(proof.end(), b, b + 32);
    return true;
  }
  void write_scalar(const Fr& s) {
    common_scalar(s);
    U256 c = s.to_canonical();
    const uint8_t* b = (const uint8_t*)c.l;
    proof.insert(proof.end(), b, b + 32);
  }
  Fr squeeze() {
    const uint8_t tag = 0;
    st.update(&tag, 1);
    uint8_t out[64];
    st.finalize_clone(out);
    uint64_t w[8];
    memcpy(w, out, 64);
    return Fr::from_u512(w);  // from_bytes_wide (transcript.rs:300-309)
  }
};

// derive/curve.rs:362-397 `batch_normalize`
void batch_normalize(const std::vector<G1Jac>& in, std::vector<G1Affine>& out) {
  out.resize(in.size());
  std::vector<Fq> pref(in.size());
  Fq acc = Fq::one();
  for (size_t i = 0; i < in.size(); i++) {
    pref[i] = acc;
    if (!in[i].is_identity()) acc = acc * in[i].z;
  }
  acc = acc.inv();
  for (size_t i = in.size(); i-- > 0;) {
    if (in[i].is_identity()) {
      out[i] = G1Affine::identity();
      continue;
    }
    Fq zi = pref[i] * acc;
    acc = acc * in[i].z;
    Fq zi2 = zi.sqr();
    out[i] = {in[i].x * zi2, in[i].y * zi2 * zi};
  }
}

G1Jac jac_from_limbs(const uint64_t* l) {
  return {Fq::from_limbs64(l), Fq::from_limbs64(l + 4), Fq::from_limbs64(l + 8)};
}

struct Rng {
  cq_rng_next_u64 next;
  void* state;
  // Fr::random (bn256/fr.rs:159-170): eight next_u64, low limb first
  void words(uint64_t* w8) {
    for (int i = 0; i < 8; i++) w8[i] = next(state);
  }
  Fr fr() {
    uint64_t w[8];
    words(w);
    return Fr::from_u512(w);
  }
};

// contiguous slice of an n-term multiexp owned by `rank` (sizes differ by at most one)
void shard_range(size_t n, uint32_t rank, uint32_t world, size_t& lo, size_t& hi) {
  const size_t base = n / world, rem = n % world;
  lo = rank * base + std::min<size_t>(rank, rem);
  hi = lo + base + (rank < rem ? 1 : 0);
}

// Commitments of one transcript round.  begin() enqueues the (possibly sharded) MSMs, end() waits,
// exchanges partial sums between ranks when sharded, and normalises (batch_normalize).
struct Commit {
  const cq_pk* pk;
  MsmPending pend;
  size_t count = 0;
  int begin(const cq_pk* pk_, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases,
            const std::vector<size_t>& lens) {
    pk = pk_;
    count = scalars.size();
    if (pk->shard_world <= 1) return msm_multi_begin(pk->ctx, scalars.data(), bases.data(), lens.data(), count, pend);
    std::vector<const Fr*> sc(count);
    std::vector<const G1Affine*> bs(count);
    std::vector<size_t> ln(count);
    for (size_t j = 0; j < count; j++) {
      size_t lo, hi;
      shard_range(lens[j], pk->shard_rank, pk->shard_world, lo, hi);
      sc[j] = scalars[j] + lo;
      bs[j] = bases[j] + lo;
      ln[j] = hi - lo;
    }
    return msm_multi_begin(pk->ctx, sc.data(), bs.data(), ln.data(), count, pend);
  }
  int end(std::vector<G1Affine>& out) {
    cq_ctx* c = pk->ctx;
    std::vector<uint64_t> jac(count * 12);
    int rc = msm_multi_end(c, pend, jac.data());
    if (rc != CQ_OK) return rc;
    std::vector<G1Jac> j(count);
    if (pk->shard_world > 1) {
      // all-gather of count x 96 B per rank, then the local EC sum (not an RCCL reduction op)
      std::vector<uint64_t> all((size_t)pk->shard_world * count * 12);
      if (pk->allgather(pk->allgather_user, jac.data(), all.data(), count * 12 * sizeof(uint64_t)) != 0)
        return c->fail(CQ_ERR_INTERNAL, "allgather callback failed");
      for (size_t i = 0; i < count; i++) {
        G1Jac acc = G1Jac::identity();
        for (uint32_t r = 0; r < pk->shard_world; r++) acc = jac_add(acc, jac_from_limbs(all.data() + ((size_t)r * count + i) * 12));
        j[i] = acc;
      }
    } else {
      for (size_t i = 0; i < count; i++) j[i] = jac_from_limbs(jac.data() + 12 * i);
    }
    batch_normalize(j, out);
    return CQ_OK;
  }
};

int commit_batch_v(const cq_pk* pk, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases,
                   const std::vector<size_t>& lens, std::vector<G1Affine>& out) {
  Commit cm;
  int rc = cm.begin(pk, scalars, bases, lens);
  if (rc != CQ_OK) return rc;
  return cm.end(out);
}

int commit_batch(const cq_pk* pk, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases, size_t len,
                 std::vector<G1Affine>& out) {
  std::vector<size_t> lens(scalars.size(), len);
  return commit_batch_v(pk, scalars, bases, lens, out);
}

}  // namespace

#define CQ_TRY(x)               \
  do {                          \
    int _rc = (x);              \
    if (_rc != CQ_OK) return _rc; \
  } while (0)

namespace cq {

size_t prover_arena_elems(const cq_pk* pk) {
  const size_t n = (size_t)1 << pk->k, ext = pk->domain->ext();
  const size_t L = pk->lookups.size(), A = pk->num_advice, N = pk->table_cfg->N;
  size_t wsum = 0;
  for (auto& lk : pk->lookups) wsum += lk.cols.size();
  size_t e = 0;
  e += A * n;          // advice (lagrange -> coeff in place)
  e += 3 * L * n;      // f_lagrange, f_coeff, b
  e += n;              // random poly
  e += 2 * L * ext;    // cosets
  e += ext;            // h on the extended coset
  e += ext;            // h coefficients (n*(j-1) = 2n)
  e += 2 * n;          // gwc batch poly + witness
  e += N + (L * N + 8) + L * N * 2 + wsum * N;  // t, den, a, m_fr, a_scaled
  e += 2 * n;          // rng staging (64 B per element = 2 Fr)
  e += L * N / 8 + 64; // m_counts (u32) + error word
  return e + 1024;
}

int create_proof_dev(cq_pk* pk, const uint64_t* const* advice_dev, cq_rng_next_u64 rng_next, void* rng_state,
                     std::vector<uint8_t>& proof_out) {
  cq_ctx* c = pk->ctx;
  cq_domain* dom = pk->domain;
  const uint32_t k = pk->k;
  const size_t n = (size_t)1 << k, ext = dom->ext();
  const uint32_t bf = pk->bf, u = pk->u;
  const size_t L = pk->lookups.size(), A = pk->num_advice;
  const size_t N = pk->table_cfg->N;
  hipStream_t s = c->stream;
  Rng rng{rng_next, rng_state};
  Transcript tr;

  // ---- carve the arena --------------------------------------------------------------------------
  void* arena_v;
  CQ_TRY(c->ensure_scratch(6, prover_arena_elems(pk) * sizeof(Fr), &arena_v));
  Fr* cur = (Fr*)arena_v;
  auto take = [&](size_t elems) {
    Fr* p = cur;
    cur += elems;
    return p;
  };
  Fr* adv = take(A * n);
  Fr* f_lag = take(L * n);
  Fr* f_coeff = take(L * n);
  Fr* bpoly = take(L * n);
  Fr* random_poly = take(n);
  Fr* cosets = take(2 * L * ext);
  Fr* h_ext = take(ext);
  Fr* h_coeff = take(ext);
  Fr* gwc_batch = take(n);
  Fr* gwc_wit = take(n);
  Fr* t_comp = take(N);
  Fr* den = take(L * N + 8);
  Fr* a_val = take(L * N);
  Fr* m_fr = take(L * N);
  size_t wsum = 0;
  for (auto& lk : pk->lookups) wsum += lk.cols.size();
  Fr* a_scaled = take(wsum * N);
  uint64_t* rng_dev = (uint64_t*)take(2 * n);
  uint32_t* m_counts = (uint32_t*)take(L * N / 8 + 64);
  uint32_t* err_dev = m_counts + L * N;

  // prover.rs:85 -- vk.hash_into(transcript)
  tr.common_scalar(pk->vk_repr);

  // ---- advice: copy in, blind rows u..n (prover.rs:346-350), one unused blind per column (:352-355) ----
  for (size_t a = 0; a < A; a++)
    CQ_HIP(c, hipMemcpyAsync(adv + a * n, advice_dev[a], (size_t)u * sizeof(Fr), hipMemcpyDeviceToDevice, s));
  {
    std::vector<Fr> tails(A * (n - u));
    for (size_t a = 0; a < A; a++)
      for (size_t r = 0; r < n - u; r++) tails[a * (n - u) + r] = rng.fr();
    for (size_t a = 0; a < A; a++) (void)rng.fr();
    void* pin;
    CQ_TRY(c->ensure_pinned(std::max(tails.size() * sizeof(Fr), (size_t)64 * n), &pin));
    memcpy(pin, tails.data(), tails.size() * sizeof(Fr));
    for (size_t a = 0; a < A; a++)
      CQ_HIP(c, hipMemcpyAsync(adv + a * n + u, (Fr*)pin + a * (n - u), (n - u) * sizeof(Fr), hipMemcpyHostToDevice, s));
    CQ_HIP(c, hipStreamSynchronize(s));  // pinned buffer is reused below
    // commit_lagrange per column (:356-360): enqueue now, collect after the host work below
    std::vector<const Fr*> sc(A);
    std::vector<const G1Affine*> bs(A, pk->params->g_lagrange);
    std::vector<size_t> ln(A, n);
    for (size_t a = 0; a < A; a++) sc[a] = adv + a * n;
    Commit adv_cm;
    if (A) CQ_TRY(adv_cm.begin(pk, sc, bs, ln));
    // The next draws from the RNG are the vanishing argument's n coefficients + 1 blind
    // (vanishing/prover.rs:51-55): the CQ rounds in between draw nothing, so taking them now keeps the
    // stream order, overlaps the host-side draws with the advice MSMs, and lets the random
    // polynomial's commitment ride along with round 2's launch.
    uint64_t* w = (uint64_t*)pin;
    for (size_t i = 0; i < n; i++) rng.words(w + 8 * i);
    (void)rng.fr();  // random_blind
    CQ_HIP(c, hipMemcpyAsync(rng_dev, pin, (size_t)64 * n, hipMemcpyHostToDevice, s));
    CQ_TRY(poly_from_u512(c, rng_dev, (uint32_t)n, random_poly));
    // batch_normalize (:363-366), write (:370-374)
    if (A) {
      std::vector<G1Affine> pts;
      CQ_TRY(adv_cm.end(pts));
      for (auto& p : pts)
        if (!tr.write_point(p)) return c->fail(CQ_ERR_TRANSCRIPT, "advice commitment is the identity");
    } else {
      CQ_HIP(c, hipStreamSynchronize(s));
    }
  }
  const Fr theta = tr.squeeze();  // :472

  // ---- CQ round 1 (static_lookup/prover.rs:51-183) ------------------------------------------------
  CQ_HIP(c, hipMemsetAsync(m_counts, 0, (L * N + 16) * sizeof(uint32_t), s));
  for (size_t l = 0; l < L; l++) {
    const cq_lookup_desc& lk = pk->lookups[l];
    const uint32_t w = (uint32_t)lk.cols.size();
    // f = sum_j theta^(w-1-j) * e_j   (:108-116, Horner with the first expression first)
    LincombArgs la;
    la.count = w;
    la.sub_const = Fr::zero();
    Fr p = Fr::one();
    for (int j = (int)w - 1; j >= 0; j--) {
      la.p[j] = adv + (size_t)lk.cols[j] * n;
      la.len[j] = (uint32_t)n;
      la.coeff[j] = p;
      p = p * theta;
    }
    CQ_TRY(poly_lincomb(c, la, (uint32_t)n, f_lag + l * n));
    CqRound1Args ra;
    ra.width = w;
    for (uint32_t j = 0; j < w; j++) {
      ra.cols[j] = adv + (size_t)lk.cols[j] * n;
      ra.values[j] = lk.tables[j]->values;
      ra.slots[j] = lk.tables[j]->slots;
      ra.nslots[j] = lk.tables[j]->nslots;
    }
    CQ_TRY(cq_round1(c, ra, u, m_counts + l * N, err_dev));
    CQ_TRY(cq_m_to_fr(c, m_counts + l * N, (uint32_t)N, m_fr + l * N));
  }
  {
    uint32_t herr = 0;
    CQ_HIP(c, hipMemcpyAsync(&herr, err_dev, 4, hipMemcpyDeviceToHost, s));
    CQ_HIP(c, hipStreamSynchronize(s));
    if (herr == 1) return c->fail(CQ_ERR_LOOKUP, "witness value not in table");
    if (herr == 2) return c->fail(CQ_ERR_LOOKUP, "Vector lookup must be on the same table row");
  }
  if (L) {
    // f_cm (:165) and m_cm (:167-172, as a dense MSM over the table SRS): one launch
    std::vector<const Fr*> sc;
    std::vector<const G1Affine*> bs;
    std::vector<size_t> ln;
    for (size_t l = 0; l < L; l++) { sc.push_back(f_lag + l * n); bs.push_back(pk->params->g_lagrange); ln.push_back(n); }
    for (size_t l = 0; l < L; l++) { sc.push_back(m_fr + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N); }
    std::vector<G1Affine> cm;
    CQ_TRY(commit_batch_v(pk, sc, bs, ln, cm));
    for (size_t l = 0; l < L; l++) {
      if (!tr.write_point(cm[l])) return c->fail(CQ_ERR_TRANSCRIPT, "f commitment is the identity");
      if (!tr.write_point(cm[L + l])) return c->fail(CQ_ERR_TRANSCRIPT, "m commitment is the identity");
    }
  }
  const Fr beta = tr.squeeze();   // prover.rs:529
  (void)tr.squeeze();             // gamma (:532), unused without permutations / legacy lookups
  const Fr beta_inv = beta.inv();

  // ---- CQ round 2 (static_lookup/prover.rs:187-342) -------------------------------------------------
  std::vector<Fr> a_at_zero(L);
  G1Affine random_cm = G1Affine::identity();
  {
    size_t woff = 0;
    for (size_t l = 0; l < L; l++) {
      const cq_lookup_desc& lk = pk->lookups[l];
      const uint32_t w = (uint32_t)lk.cols.size();
      // t_i = sum_j theta^(w-1-j) T_j[i]  (compress_tables :224-240)
      LincombArgs la;
      la.count = w;
      la.sub_const = Fr::zero();
      CqThetaPowers tp;
      tp.width = w;
      Fr p = Fr::one();
      for (int j = (int)w - 1; j >= 0; j--) {
        la.p[j] = lk.tables[j]->values;
        la.len[j] = (uint32_t)N;
        la.coeff[j] = p;
        tp.pow[j] = p;
        p = p * theta;
      }
      CQ_TRY(poly_lincomb(c, la, (uint32_t)N, t_comp));
      CQ_TRY(cq_a_denominators(c, t_comp, m_counts + l * N, (uint32_t)N, beta, den + l * N));
      // B_r = f_r + beta, r < u ; beta on the blinding rows (:261-269)
      CQ_TRY(poly_cq_b_denominators(c, f_lag + l * n, (uint32_t)n, u, beta, bpoly + l * n));
    }
    if (L) {
      // all inversions of the round in two launches (one Fermat inversion per lane dominates the latency)
      CQ_TRY(poly_batch_invert(c, den, (uint32_t)(L * N)));
      CQ_TRY(poly_batch_invert(c, bpoly, (uint32_t)(L * n)));
    }
    woff = 0;
    for (size_t l = 0; l < L; l++) {
      const uint32_t w = (uint32_t)pk->lookups[l].cols.size();
      CqThetaPowers tp;
      tp.width = w;
      Fr p = Fr::one();
      for (int j = (int)w - 1; j >= 0; j--) {
        tp.pow[j] = p;
        p = p * theta;
      }
      CQ_TRY(cq_a_values(c, den + l * N, m_counts + l * N, (uint32_t)N, tp, a_val + l * N, a_scaled + woff * N));
      woff += w;
    }
    if (L) {
      CQ_TRY(domain_lagrange_to_coeff(dom, bpoly, bpoly, (uint32_t)L, n, n));
      CQ_TRY(domain_lagrange_to_coeff(dom, f_lag, f_coeff, (uint32_t)L, n, n));  // :326-334
    }
    // commitments: a, a0 (dense over the table SRS), q_a (over [qs_0|qs_1|...]), p, b0 (n-1 terms of b[1..])
    // and the vanishing argument's random polynomial (vanishing/prover.rs:58) in one batch of launches
    std::vector<G1Affine> r2;
    {
      std::vector<const Fr*> sc;
      std::vector<const G1Affine*> bs;
      std::vector<size_t> ln;
      woff = 0;
      for (size_t l = 0; l < L; l++) {
        const uint32_t w = (uint32_t)pk->lookups[l].cols.size();
        sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N);             // a
        sc.push_back(a_scaled + woff * N); bs.push_back(pk->qs_concat[l]); ln.push_back((size_t)w * N);      // q_a
        sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g_lagrange_opening_at_0); ln.push_back(N);  // a0
        sc.push_back(bpoly + l * n + 1); bs.push_back(pk->params->g); ln.push_back(n - 1);                   // b0 (:310)
        sc.push_back(bpoly + l * n + 1); bs.push_back(pk->b0_g1_bound); ln.push_back(n - 1);                 // p (:299)
        woff += w;
      }
      sc.push_back(random_poly); bs.push_back(pk->params->g); ln.push_back(n);
      CQ_TRY(commit_batch_v(pk, sc, bs, ln, r2));
    }
    for (size_t l = 0; l < L; l++) {
      // write order :306-313: a, q_a, a0, b0, p
      for (size_t q = 0; q < 5; q++)
        if (!tr.write_point(r2[5 * l + q])) return c->fail(CQ_ERR_TRANSCRIPT, "a CQ round-2 commitment is the identity");
    }
    random_cm = r2[5 * L];
    // a(0) = (n*b(0) - (bf+1)/beta) / N   (:318-324)
    if (L) {
      std::vector<Fr> b0(L);
      for (size_t l = 0; l < L; l++)
        CQ_HIP(c, hipMemcpyAsync(&b0[l], bpoly + l * n, sizeof(Fr), hipMemcpyDeviceToHost, s));
      CQ_HIP(c, hipStreamSynchronize(s));
      const Fr n_table_inv = Fr::from_u64(N).inv();
      for (size_t l = 0; l < L; l++)
        a_at_zero[l] = (b0[l] * Fr::from_u64(n) - Fr::from_u64(bf + 1) * beta_inv) * n_table_inv;
    }
  }

  // ---- vanishing::Argument::commit (vanishing/prover.rs:37-65): drawn and committed above ---------
  if (!tr.write_point(random_cm)) return c->fail(CQ_ERR_TRANSCRIPT, "random poly commitment is the identity");
  const Fr y = tr.squeeze();  // prover.rs:584

  // advice polys: lagrange_to_coeff (:587-603), in place
  if (A) CQ_TRY(domain_lagrange_to_coeff(dom, adv, adv, (uint32_t)A, n, n));

  // ---- evaluate_h, CQ terms (evaluation.rs:533-548) + divide by the vanishing polynomial -----------
  {
    if (L) {
      CQ_TRY(domain_coeff_to_extended(dom, bpoly, cosets, (uint32_t)L, n, ext));
      CQ_TRY(domain_coeff_to_extended(dom, f_coeff, cosets + L * ext, (uint32_t)L, n, ext));
    }
    CqQuotientArgs qa;
    qa.count = (uint32_t)L;
    for (size_t l = 0; l < L; l++) {
      qa.b[l] = cosets + l * ext;
      qa.f[l] = cosets + (L + l) * ext;
    }
    qa.l_active = pk->l_active_row;
    qa.t_evals = dom->t_evaluations_dev;
    qa.t_len = (uint32_t)dom->t_evaluations.size();
    qa.y = y;
    qa.beta = beta;
    CQ_TRY(poly_cq_quotient(c, qa, (uint32_t)ext, h_ext));
  }
  // vanishing.construct (vanishing/prover.rs:69-120): coefficients, n-sized pieces, blinds, commitments
  const size_t pieces = dom->quotient_poly_degree;
  CQ_TRY(domain_extended_to_coeff(dom, h_ext, h_coeff));
  for (size_t i = 0; i < pieces; i++) (void)rng.fr();  // h_blinds (:95-98)
  {
    std::vector<const Fr*> sc(pieces);
    std::vector<const G1Affine*> bs(pieces, pk->params->g);
    for (size_t i = 0; i < pieces; i++) sc[i] = h_coeff + i * n;
    std::vector<G1Affine> o;
    CQ_TRY(commit_batch(pk, sc, bs, n, o));
    for (auto& p : o)
      if (!tr.write_point(p)) return c->fail(CQ_ERR_TRANSCRIPT, "h piece commitment is the identity");
  }
  const Fr x = tr.squeeze();  // prover.rs:629
  const Fr xn = x.pow_u64(n);

  // ---- evaluations (prover.rs:654-719): every polynomial is opened at x, one batched launch ---------
  const size_t pieces_n = pieces;
  std::vector<Fr> advice_evals(pk->advice_queries.size()), b0_evals(L), f_evals(L), h_evals(pieces_n);
  Fr random_eval;
  {
    std::vector<const Fr*> ps;
    std::vector<uint32_t> ls;
    for (auto& q : pk->advice_queries) { ps.push_back(adv + (size_t)q.first * n); ls.push_back((uint32_t)n); }
    ps.push_back(random_poly); ls.push_back((uint32_t)n);
    for (size_t l = 0; l < L; l++) {
      ps.push_back(bpoly + l * n + 1); ls.push_back((uint32_t)(n - 1));  // b0 = (b - b(0))/X
      ps.push_back(f_coeff + l * n); ls.push_back((uint32_t)n);
    }
    for (size_t i = 0; i < pieces_n; i++) { ps.push_back(h_coeff + i * n); ls.push_back((uint32_t)n); }
    std::vector<Fr> ev(ps.size());
    for (size_t off = 0; off < ps.size(); off += EVAL_MAX_BATCH) {
      const uint32_t cnt = (uint32_t)std::min((size_t)EVAL_MAX_BATCH, ps.size() - off);
      CQ_TRY(poly_eval_batch(c, ps.data() + off, ls.data() + off, cnt, x, ev.data() + off));
    }
    size_t e = 0;
    for (size_t q = 0; q < advice_evals.size(); q++) advice_evals[q] = ev[e++];
    random_eval = ev[e++];
    for (size_t l = 0; l < L; l++) { b0_evals[l] = ev[e++]; f_evals[l] = ev[e++]; }
    for (size_t i = 0; i < pieces_n; i++) h_evals[i] = ev[e++];
  }
  for (auto& v : advice_evals) tr.write_scalar(v);
  tr.write_scalar(random_eval);  // vanishing/prover.rs:145-146
  for (size_t l = 0; l < L; l++) {  // static_lookup/prover.rs:360-370
    tr.write_scalar(b0_evals[l]);
    tr.write_scalar(f_evals[l]);
    tr.write_scalar(a_at_zero[l]);
  }

  // ---- multiopen, GWC (gwc/prover.rs:42-91): every query is at x => one point group ---------------------
  {
    const Fr v = tr.squeeze();
    // h(X) = sum_i xn^i h_i (vanishing/prover.rs:131-135); get_eval's value follows from the piece evals
    Fr h_eval = Fr::zero();
    for (size_t i = pieces; i-- > 0;) h_eval = h_eval * xn + h_evals[i];
    LincombArgs la;
    la.count = 0;
    Fr pv = Fr::one();
    Fr eval_batch = Fr::zero();
    auto push = [&](const Fr* p, uint32_t len, const Fr& coeff) {
      la.p[la.count] = p;
      la.len[la.count] = len;
      la.coeff[la.count] = coeff;
      la.count++;
    };
    if (pk->advice_queries.size() + 2 * L + pieces + 1 > LINCOMB_MAX) return c->fail(CQ_ERR_ARG, "too many opening queries");
    for (size_t q = 0; q < pk->advice_queries.size(); q++) {
      push(adv + (size_t)pk->advice_queries[q].first * n, (uint32_t)n, pv);
      eval_batch = eval_batch + advice_evals[q] * pv;
      pv = pv * v;
    }
    for (size_t l = 0; l < L; l++) {
      push(bpoly + l * n + 1, (uint32_t)(n - 1), pv);
      eval_batch = eval_batch + b0_evals[l] * pv;
      pv = pv * v;
      push(f_coeff + l * n, (uint32_t)n, pv);
      eval_batch = eval_batch + f_evals[l] * pv;
      pv = pv * v;
    }
    {
      Fr xp = Fr::one();
      for (size_t i = 0; i < pieces; i++) {
        push(h_coeff + i * n, (uint32_t)n, pv * xp);
        xp = xp * xn;
      }
      eval_batch = eval_batch + h_eval * pv;
      pv = pv * v;
    }
    push(random_poly, (uint32_t)n, pv);
    eval_batch = eval_batch + random_eval * pv;
    la.sub_const = eval_batch;
    CQ_TRY(poly_lincomb(c, la, (uint32_t)n, gwc_batch));
    CQ_TRY(poly_kate_division(c, gwc_batch, (uint32_t)n, x, gwc_wit));
    std::vector<const Fr*> sc{gwc_wit};
    std::vector<const G1Affine*> bs{pk->params->g};
    std::vector<G1Affine> o;
    CQ_TRY(commit_batch(pk, sc, bs, n - 1, o));
    if (!tr.write_point(o[0])) return c->fail(CQ_ERR_TRANSCRIPT, "opening witness commitment is the identity");
  }
  proof_out.swap(tr.proof);
  return CQ_OK;
}

}  // namespace cq
