// C ABI: the CQ sub-arguments of create_proof as stand-alone entry points, for a host that keeps
// its own orchestrator (transcript, RNG, the other arguments) and swaps in only these:
//   cq_cq_round1_dev  = static_lookup::Argument::commit               (plonk/static_lookup/prover.rs:51-183)
//   cq_cq_round2_dev  = static_lookup::Committed::commit_log_derivatives (:187-342)
//   cq_quotient_dev   = the static-lookup terms of Evaluator::evaluate_h (plonk/evaluation.rs:533-548), optionally followed
//                       by EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:319-338)
//   cq_permute_expression_pair_dev = permute_expression_pair of the legacy lookup argument (plonk/lookup/prover.rs:400-502)
// They run the same kernels as cq_create_proof (prover.hip), without its cross-round overlap.
#include <algorithm>
#include <cstring>
#include <vector>
#include "cq.hpp"
#include "ctx.hpp"
#include "msm.hpp"
#include "plonk.hpp"
#include "poly.hpp"

using namespace cq;

#define CQ_TRY(x)                 \
  do {                            \
    int _rc = (x);                \
    if (_rc != CQ_OK) return _rc; \
  } while (0)

namespace {

int commit_affine(cq_pk* pk, const std::vector<const Fr*>& sc, const std::vector<const G1Affine*>& bs, const std::vector<size_t>& ln,
                  uint64_t* out_affine) {
  std::vector<uint64_t> jac(sc.size() * 12);
  CQ_TRY(cq_msm_multi_v(pk->ctx, sc.data(), bs.data(), ln.data(), sc.size(), jac.data()));
  for (size_t i = 0; i < sc.size(); i++) CQ_TRY(cq_g1_to_affine(jac.data() + 12 * i, out_affine + 8 * i));
  return CQ_OK;
}

// theta^(w-1-j), j < w: the first expression gets the highest power (Horner, :108-116)
void theta_powers(const Fr& theta, uint32_t w, Fr* pow) {
  Fr p = Fr::one();
  for (int j = (int)w - 1; j >= 0; j--) {
    pow[j] = p;
    p = p * theta;
  }
}

}  // namespace

extern "C" {

int cq_cq_round1_dev(cq_pk* pk, const uint64_t* const* advice_dev, const uint64_t* const* instance_dev, const uint64_t* challenges,
                     const uint64_t theta[4], uint64_t* f_dev, uint32_t* m_dev, uint64_t* commitments) {
  if (!pk || !theta || !f_dev || !m_dev || !commitments || (pk->num_advice && !advice_dev)) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t L = pk->lookups.size(), n = (size_t)1 << pk->k, A = pk->num_advice, I = pk->num_instance;
  if (!L) return CQ_OK;
  if (I && !instance_dev) return c->fail(CQ_ERR_ARG, "cq_round1: instance columns missing");
  const size_t N = pk->table_cfg->N, NC = pk->challenge_phase.size();
  if (NC && !challenges) return c->fail(CQ_ERR_ARG, "cq_round1: challenges missing");
  hipStream_t s = c->stream;
  size_t wsum = 0;
  for (auto& lk : pk->lookups) wsum += lk.cols.size();
  // scratch: contiguous copies of the columns (the expression interpreter addresses column c at base + c * n),
  // evaluated inputs, m as field elements, the error flag
  void* scr;
  const size_t elems = (A + I) * n + (pk->lookup_exprs ? wsum * n : 0) + L * N + NC + 16;
  CQ_TRY(c->ensure_scratch(7, elems * sizeof(Fr), &scr));
  Fr* adv = (Fr*)scr;
  Fr* inst = adv + A * n;
  Fr* inputs = inst + I * n;
  Fr* m_fr = inputs + (pk->lookup_exprs ? wsum * n : 0);
  Fr* chal = m_fr + L * N;
  uint32_t* err_dev = (uint32_t*)(chal + NC + 8);
  for (size_t a = 0; a < A; a++) CQ_HIP(c, hipMemcpyAsync(adv + a * n, advice_dev[a], n * sizeof(Fr), hipMemcpyDeviceToDevice, s));
  for (size_t i = 0; i < I; i++) CQ_HIP(c, hipMemcpyAsync(inst + i * n, instance_dev[i], n * sizeof(Fr), hipMemcpyDeviceToDevice, s));
  if (NC) CQ_HIP(c, hipMemcpyAsync(chal, challenges, NC * sizeof(Fr), hipMemcpyHostToDevice, s));
  CQ_HIP(c, hipMemsetAsync(m_dev, 0, L * N * sizeof(uint32_t), s));
  CQ_HIP(c, hipMemsetAsync(err_dev, 0, 64, s));
  const Fr th = Fr::from_limbs64(theta);
  size_t slot = 0;
  CqRound1Batch r1b;
  r1b.count = 0;
  for (size_t l = 0; l < L; l++) {
    const cq_lookup_desc& lk = pk->lookups[l];
    const uint32_t w = (uint32_t)lk.cols.size();
    const Fr* input[CQ_MAX_WIDTH];
    for (uint32_t j = 0; j < w; j++, slot++) {  // `evaluate(expr, n, 1, fixed, advice, instance)` (:91-107)
      if (lk.prog[j] < 0) {
        input[j] = adv + (size_t)lk.cols[j] * n;
        continue;
      }
      GateEvalArgs ga;
      ga.prog = pk->lookup_prog + lk.prog[j];
      ga.num_polys = 1;
      ga.constants = pk->constants;
      ga.challenges = chal;
      ga.advice = adv;
      ga.fixed = pk->fixed_values;
      ga.instance = inst;
      ga.stride = n;
      ga.size = (uint32_t)n;
      ga.rot_scale = 1;
      ga.y = Fr::zero();
      CQ_TRY(gate_eval(c, ga, inputs + slot * n));
      input[j] = inputs + slot * n;
    }
    LincombArgs la;  // f = sum_j theta^(w-1-j) e_j (:108-116)
    la.count = w;
    la.sub_const = Fr::zero();
    theta_powers(th, w, la.coeff);
    for (uint32_t j = 0; j < w; j++) {
      la.p[j] = input[j];
      la.len[j] = (uint32_t)n;
    }
    CQ_TRY(poly_lincomb(c, la, (uint32_t)n, (Fr*)f_dev + l * n));
    CqRound1Args& ra = r1b.a[r1b.count];  // value -> table index, one index per row, multiplicities (:122-160)
    ra.width = w;
    for (uint32_t j = 0; j < w; j++) {
      ra.cols[j] = input[j];
      ra.values[j] = lk.tables[j]->values;
      ra.slots[j] = lk.tables[j]->slots;
      ra.nslots[j] = lk.tables[j]->nslots;
    }
    r1b.m_counts[r1b.count++] = m_dev + l * N;
    if (r1b.count == CQ_ROUND1_BATCH || l + 1 == L) {
      CQ_TRY(cq_round1(c, r1b, pk->u, err_dev));
      r1b.count = 0;
    }
  }
  CQ_TRY(cq_m_to_fr(c, m_dev, (uint32_t)(L * N), m_fr));
  uint32_t herr = 0;
  CQ_HIP(c, hipMemcpyAsync(&herr, err_dev, 4, hipMemcpyDeviceToHost, s));
  CQ_HIP(c, hipStreamSynchronize(s));
  if (herr == 1) return c->fail(CQ_ERR_LOOKUP, "witness value not in table");
  if (herr == 2) return c->fail(CQ_ERR_LOOKUP, "Vector lookup must be on the same table row");
  // f_cm = commit_lagrange(f) (:165), m_cm = sum m_i [L_i] (:167-172): per lookup (f_cm, m_cm)
  std::vector<const Fr*> sc;
  std::vector<const G1Affine*> bs;
  std::vector<size_t> ln;
  for (size_t l = 0; l < L; l++) {
    sc.push_back((const Fr*)f_dev + l * n); bs.push_back(pk->params->g_lagrange); ln.push_back(n);
    sc.push_back(m_fr + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N);
  }
  return commit_affine(pk, sc, bs, ln, commitments);
}

int cq_cq_round2_dev(cq_pk* pk, const uint64_t* f_dev, const uint32_t* m_dev, const uint64_t theta[4], const uint64_t beta[4],
                     uint64_t* b_coeff_dev, uint64_t* f_coeff_dev, uint64_t* commitments, uint64_t* a_at_zero) {
  if (!pk || !f_dev || !m_dev || !theta || !beta || !b_coeff_dev || !f_coeff_dev || !commitments || !a_at_zero) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t L = pk->lookups.size(), n = (size_t)1 << pk->k;
  if (!L) return CQ_OK;
  const size_t N = pk->table_cfg->N;
  const uint32_t bf = pk->bf, u = pk->u;
  const Fr th = Fr::from_limbs64(theta), be = Fr::from_limbs64(beta);
  if (be.is_zero()) return c->fail(CQ_ERR_ARG, "cq_round2: beta = 0");
  hipStream_t s = c->stream;
  size_t wsum = 0;
  for (auto& lk : pk->lookups) wsum += lk.cols.size();
  void* scr;
  CQ_TRY(c->ensure_scratch(7, (N + 2 * L * N + wsum * N + 64) * sizeof(Fr), &scr));
  Fr* t_comp = (Fr*)scr;
  Fr* den = t_comp + N;
  Fr* a_val = den + L * N;
  Fr* a_scaled = a_val + L * N;
  Fr* bpoly = (Fr*)b_coeff_dev;
  const Fr* f_lag = (const Fr*)f_dev;
  size_t woff = 0;
  for (size_t l = 0; l < L; l++) {
    const cq_lookup_desc& lk = pk->lookups[l];
    const uint32_t w = (uint32_t)lk.cols.size();
    LincombArgs la;  // t_i = sum_j theta^(w-1-j) T_j[i] (compress_tables, :224-240)
    la.count = w;
    la.sub_const = Fr::zero();
    theta_powers(th, w, la.coeff);
    for (uint32_t j = 0; j < w; j++) {
      la.p[j] = lk.tables[j]->values;
      la.len[j] = (uint32_t)N;
    }
    CQ_TRY(poly_lincomb(c, la, (uint32_t)N, t_comp));
    CQ_TRY(cq_a_denominators(c, t_comp, m_dev + l * N, (uint32_t)N, be, den + l * N));  // t_i + beta where m_i != 0
    CQ_TRY(poly_batch_invert(c, den + l * N, (uint32_t)N));
    CqThetaPowers tp;
    tp.width = w;
    theta_powers(th, w, tp.pow);
    CQ_TRY(cq_a_values(c, den + l * N, m_dev + l * N, (uint32_t)N, tp, a_val + l * N, a_scaled + woff * N));  // A_i = m_i / (t_i + beta) (:245-257)
    woff += w;
    // B_r = 1 / (f_r + beta) on the usable rows, 1 / beta on the others (:261-269); b = iNTT(B) (:271-276)
    CQ_TRY(poly_cq_b_denominators(c, f_lag + l * n, (uint32_t)n, u, be, bpoly + l * n));
  }
  CQ_TRY(poly_batch_invert(c, bpoly, (uint32_t)(L * n)));
  CQ_TRY(domain_lagrange_to_coeff(pk->domain, bpoly, bpoly, (uint32_t)L, n, n));
  CQ_TRY(domain_lagrange_to_coeff(pk->domain, f_lag, (Fr*)f_coeff_dev, (uint32_t)L, n, n));  // :326-334
  // a, q_a, a_0 (over the table SRS, the cached quotients), b_0 = (b - b(0)) / X over g, p over b0_g1_bound: per lookup in
  // the order the reference writes them (:306-313)
  std::vector<const Fr*> sc;
  std::vector<const G1Affine*> bs;
  std::vector<size_t> ln;
  woff = 0;
  for (size_t l = 0; l < L; l++) {
    const uint32_t w = (uint32_t)pk->lookups[l].cols.size();
    sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N);
    sc.push_back(a_scaled + woff * N); bs.push_back(pk->qs_concat[l]); ln.push_back((size_t)w * N);
    sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g_lagrange_opening_at_0); ln.push_back(N);
    sc.push_back(bpoly + l * n + 1); bs.push_back(pk->params->g); ln.push_back(n - 1);
    sc.push_back(bpoly + l * n + 1); bs.push_back(pk->b0_g1_bound); ln.push_back(n - 1);
    woff += w;
  }
  CQ_TRY(commit_affine(pk, sc, bs, ln, commitments));
  // a(0) = (n b(0) - (bf + 1) / beta) / N (:318-324)
  std::vector<Fr> b0(L);
  for (size_t l = 0; l < L; l++) CQ_HIP(c, hipMemcpyAsync(&b0[l], bpoly + l * n, sizeof(Fr), hipMemcpyDeviceToHost, s));
  CQ_HIP(c, hipStreamSynchronize(s));
  const Fr n_table_inv = Fr::from_u64(N).inv(), beta_inv = be.inv();
  for (size_t l = 0; l < L; l++) {
    const Fr a0 = (b0[l] * Fr::from_u64(n) - Fr::from_u64(bf + 1) * beta_inv) * n_table_inv;
    a0.to_limbs64(a_at_zero + 4 * l);
  }
  return CQ_OK;
}

int cq_quotient_dev(cq_pk* pk, const uint64_t* b_coeff_dev, const uint64_t* f_coeff_dev, const uint64_t y[4], const uint64_t beta[4],
                    const uint64_t* h_in_dev, int divide_by_vanishing, uint64_t* h_out_dev) {
  if (!pk || !y || !beta || !h_out_dev) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t L = pk->lookups.size(), n = (size_t)1 << pk->k, ext = pk->domain->ext();
  if (L && (!b_coeff_dev || !f_coeff_dev)) return CQ_ERR_ARG;
  void* scr;
  CQ_TRY(c->ensure_scratch(7, (2 * L * ext + 16) * sizeof(Fr), &scr));
  Fr* cosets = (Fr*)scr;  // b, then f, on the extended coset (evaluation.rs:535-536)
  if (L) {
    CQ_TRY(domain_coeff_to_extended(pk->domain, (const Fr*)b_coeff_dev, cosets, (uint32_t)L, n, ext));
    CQ_TRY(domain_coeff_to_extended(pk->domain, (const Fr*)f_coeff_dev, cosets + L * ext, (uint32_t)L, n, ext));
  }
  CqQuotientArgs qa;
  qa.count = (uint32_t)L;
  for (size_t l = 0; l < L; l++) {
    qa.b[l] = cosets + l * ext;
    qa.f[l] = cosets + (L + l) * ext;
  }
  qa.h_in = (const Fr*)h_in_dev;
  qa.l_active = pk->l_active_row;
  qa.t_evals = pk->domain->t_evaluations_dev;
  qa.t_len = divide_by_vanishing ? (uint32_t)pk->domain->t_evaluations.size() : 0u;
  qa.y = Fr::from_limbs64(y);
  qa.beta = Fr::from_limbs64(beta);
  return poly_cq_quotient(c, qa, (uint32_t)ext, (Fr*)h_out_dev);
}

/* permute_expression_pair (plonk/lookup/prover.rs:400-502) without its blinding rows: device sort and matching, lksort.hip */
int cq_permute_expression_pair_dev(cq_ctx* c, uint32_t k, uint32_t usable, const uint64_t* input_dev, const uint64_t* table_dev,
                                   uint64_t* permuted_input_dev, uint64_t* permuted_table_dev) {
  if (!c || k > 28 || usable > (1u << k) || (usable && (!input_dev || !table_dev || !permuted_input_dev || !permuted_table_dev)))
    return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t slot = ((size_t)1 << k) * 4;
  void* scr;
  CQ_TRY(c->ensure_scratch(7, 3 * slot * sizeof(uint64_t) + lookup_permute_scratch_bytes(k) + 64, &scr));
  uint64_t* stage = (uint64_t*)scr;
  void* lk_scratch = stage + 3 * slot;
  uint32_t* status = (uint32_t*)((char*)lk_scratch + lookup_permute_scratch_bytes(k));
  CQ_TRY(fr_to_canonical(c, (const Fr*)input_dev, usable, stage));
  CQ_TRY(fr_to_canonical(c, (const Fr*)table_dev, usable, stage + slot));
  CQ_TRY(lookup_permute_dev(c, stage, stage + slot, usable, k, stage + 2 * slot, lk_scratch, status));
  CQ_TRY(fr_from_canonical(c, stage, usable, (Fr*)permuted_input_dev));
  CQ_TRY(fr_from_canonical(c, stage + 2 * slot, usable, (Fr*)permuted_table_dev));
  uint32_t st[3];
  CQ_HIP(c, hipMemcpyAsync(st, status, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  if (st[0] != 0 || st[1] != st[2]) return c->fail(CQ_ERR_LOOKUP, "lookup input not in table (Error::ConstraintSystemFailure)");
  return CQ_OK;
}

}  // extern "C"
