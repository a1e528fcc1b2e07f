// C ABI: CQ table objects, proving key, create_proof, SHA witness fill, harness RNGs.
#include <algorithm>
#include <cstring>
#include <vector>
#include "cq.hpp"
#include "ctx.hpp"
#include "prover.hpp"
#include "setup.hpp"
#include "msm.hpp"

using namespace cq;

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static uint32_t log2u(size_t x) {
  uint32_t l = 0;
  while (((size_t)1 << (l + 1)) <= x) l++;
  return l;
}

extern "C" {

// ---- StaticTableConfig ---------------------------------------------------------------------------
int cq_table_config_create(cq_ctx* c, size_t size, const uint64_t* g1_lagrange, const uint64_t* opening_at_0,
                           cq_table_config** out) {
  if (!c || !g1_lagrange || !opening_at_0 || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  cq_table_config* t = new cq_table_config();
  t->ctx = c;
  t->N = size;
  t->log_n = log2u(size);
  const size_t bytes = size * sizeof(G1Affine);
  if (hipMalloc(&t->g1_lagrange, bytes) != hipSuccess || hipMalloc(&t->g_lagrange_opening_at_0, bytes) != hipSuccess) {
    delete t;
    return c->fail(CQ_ERR_HIP, "hipMalloc(table config)");
  }
  CQ_HIP(c, hipMemcpyAsync(t->g1_lagrange, g1_lagrange, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipMemcpyAsync(t->g_lagrange_opening_at_0, opening_at_0, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  if (c->msm_precompute) {
    int rc2;
    if ((rc2 = msm_register_tables(c, t->g1_lagrange, size)) != CQ_OK) return rc2;
    if ((rc2 = msm_register_tables(c, t->g_lagrange_opening_at_0, size)) != CQ_OK) return rc2;
  }
  *out = t;
  return CQ_OK;
}

int cq_table_config_setup_from_toxic_waste(cq_ctx* c, size_t size, const uint64_t s[4], cq_table_config** out) {
  if (!c || !s || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  cq_table_config* t = new cq_table_config();
  t->ctx = c;
  t->N = size;
  t->log_n = log2u(size);
  const size_t bytes = size * sizeof(G1Affine);
  if (hipMalloc(&t->g1_lagrange, bytes) != hipSuccess || hipMalloc(&t->g_lagrange_opening_at_0, bytes) != hipSuccess) {
    delete t;
    return c->fail(CQ_ERR_HIP, "hipMalloc(table config)");
  }
  void* tmp;
  int rc;
  if ((rc = c->ensure_scratch(1, 2 * size * sizeof(Fr), &tmp)) != CQ_OK) return rc;
  Fr* lag_sc = (Fr*)tmp;
  Fr* tmp_sc = lag_sc + size;
  const Fr sf = Fr::from_limbs64(s);
  if ((rc = srs_powers_and_lagrange(c, t->log_n, sf, nullptr, t->g1_lagrange, tmp_sc, lag_sc)) != CQ_OK) return rc;
  if ((rc = srs_opening_at_zero(c, t->log_n, sf, lag_sc, tmp_sc, t->g_lagrange_opening_at_0)) != CQ_OK) return rc;
  if (c->msm_precompute) {
    if ((rc = msm_register_tables(c, t->g1_lagrange, size)) != CQ_OK) return rc;
    if ((rc = msm_register_tables(c, t->g_lagrange_opening_at_0, size)) != CQ_OK) return rc;
  }
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = t;
  return CQ_OK;
}

void cq_table_config_destroy(cq_table_config* t) {
  if (!t) return;
  hipStreamSynchronize(t->ctx->stream);
  msm_unregister_tables(t->ctx, t->g1_lagrange);
  msm_unregister_tables(t->ctx, t->g_lagrange_opening_at_0);
  if (t->g1_lagrange) hipFree(t->g1_lagrange);
  if (t->g_lagrange_opening_at_0) hipFree(t->g_lagrange_opening_at_0);
  delete t;
}

int cq_table_config_download(cq_table_config* t, uint64_t* g1_lagrange, uint64_t* opening_at_0) {
  if (!t) return CQ_ERR_ARG;
  cq_ctx* c = t->ctx;
  const size_t bytes = t->N * sizeof(G1Affine);
  if (g1_lagrange) CQ_HIP(c, hipMemcpyAsync(g1_lagrange, t->g1_lagrange, bytes, hipMemcpyDeviceToHost, c->stream));
  if (opening_at_0) CQ_HIP(c, hipMemcpyAsync(opening_at_0, t->g_lagrange_opening_at_0, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

// ---- StaticTableValues -----------------------------------------------------------------------------
static int table_alloc(cq_ctx* c, size_t size, const uint64_t* values, cq_static_table** out) {
  cq_static_table* t = new cq_static_table();
  t->ctx = c;
  t->N = size;
  if (hipMalloc(&t->values, size * sizeof(Fr)) != hipSuccess || hipMalloc(&t->qs, size * sizeof(G1Affine)) != hipSuccess) {
    delete t;
    return c->fail(CQ_ERR_HIP, "hipMalloc(static table)");
  }
  if (hipMemcpyAsync(t->values, values, size * sizeof(Fr), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
    delete t;
    return c->fail(CQ_ERR_HIP, "upload(table values)");
  }
  int rc = cq_table_build_index(c, t->values, (uint32_t)size, &t->slots, &t->nslots);
  if (rc != CQ_OK) {
    hipFree(t->values);
    hipFree(t->qs);
    delete t;
    return rc;
  }
  *out = t;
  return CQ_OK;
}

int cq_static_table_create(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t* qs_affine, cq_static_table** out) {
  if (!c || !values || !qs_affine || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;  // static_lookup.rs:80
  CQ_HIP(c, hipSetDevice(c->device));
  int rc = table_alloc(c, size, values, out);
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync((*out)->qs, qs_affine, size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

int cq_static_table_setup_from_toxic_waste(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t s[4],
                                           cq_static_table** out) {
  if (!c || !values || !s || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  int rc = table_alloc(c, size, values, out);
  if (rc != CQ_OK) return rc;
  cq_static_table* t = *out;
  // T(s): interpolate the values over the size-N domain, evaluate at s
  cq_domain* dom = nullptr;
  if ((rc = domain_create(c, 2, log2u(size), &dom)) != CQ_OK) return rc;
  void* tmp;
  if ((rc = c->ensure_scratch(1, 2 * size * sizeof(Fr), &tmp)) != CQ_OK) return rc;
  Fr* coeffs = (Fr*)tmp;
  Fr* sc = coeffs + size;
  const Fr sf = Fr::from_limbs64(s);
  Fr ts;
  if ((rc = domain_lagrange_to_coeff(dom, t->values, coeffs, 1, size, size)) != CQ_OK) return rc;
  if ((rc = poly_eval(c, coeffs, (uint32_t)size, sf, &ts)) != CQ_OK) return rc;
  if ((rc = cq_qs_scalars(c, t->values, (uint32_t)size, ts, sf, dom->omega, dom->ifft_divisor, sc)) != CQ_OK) return rc;
  if ((rc = fixed_base_mul(c, sc, (uint32_t)size, t->qs)) != CQ_OK) return rc;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  domain_destroy(dom);
  return CQ_OK;
}

void cq_static_table_destroy(cq_static_table* t) {
  if (!t) return;
  hipStreamSynchronize(t->ctx->stream);
  if (t->values) hipFree(t->values);
  if (t->qs) hipFree(t->qs);
  if (t->slots) hipFree(t->slots);
  delete t;
}

int cq_static_table_download_qs(cq_static_table* t, uint64_t* qs_affine) {
  if (!t || !qs_affine) return CQ_ERR_ARG;
  cq_ctx* c = t->ctx;
  CQ_HIP(c, hipMemcpyAsync(qs_affine, t->qs, t->N * sizeof(G1Affine), hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

// ---- proving key -------------------------------------------------------------------------------------
int cq_pk_create(cq_ctx* c, cq_params* params, const cq_circuit* cs, cq_table_config* cfg, const uint64_t* b0_g1_bound,
                 int b0_on_device, cq_pk** out) {
  if (!c || !params || !cs || !cfg || !b0_g1_bound || !out) return CQ_ERR_ARG;
  if (cs->k != params->k) return c->fail(CQ_ERR_ARG, "pk: circuit k differs from params k");
  if (cs->num_lookups > CQ_MAX_LOOKUPS) return c->fail(CQ_ERR_ARG, "pk: too many lookups");
  CQ_HIP(c, hipSetDevice(c->device));
  cq_pk* pk = new cq_pk();
  pk->ctx = c;
  pk->params = params;
  pk->k = cs->k;
  pk->num_advice = cs->num_advice;
  pk->table_cfg = cfg;
  pk->vk_repr = Fr::from_limbs64(cs->vk_repr);
  size_t off = 0;
  std::vector<uint32_t> per_col(cs->num_advice, 0);
  for (uint32_t l = 0; l < cs->num_lookups; l++) {
    cq_lookup_desc d;
    const uint32_t w = cs->lookup_widths[l];
    if (w == 0 || w > CQ_MAX_WIDTH) {
      delete pk;
      return c->fail(CQ_ERR_ARG, "pk: lookup width out of range");
    }
    for (uint32_t j = 0; j < w; j++) {
      const uint32_t col = cs->lookup_columns[off + j];
      cq_static_table* t = cs->lookup_tables[off + j];
      if (col >= cs->num_advice || !t) {
        delete pk;
        return c->fail(CQ_ERR_ARG, "pk: bad lookup column / table");
      }
      if (t->N != cfg->N) {  // "Tables should all be of the same size" (static_lookup/prover.rs:81-83)
        delete pk;
        return c->fail(CQ_ERR_ARG, "pk: table size differs from the table config");
      }
      d.cols.push_back(col);
      d.tables.push_back(t);
      // query_advice_index (plonk/circuit.rs:1619-1633)
      if (std::find(pk->advice_queries.begin(), pk->advice_queries.end(), std::make_pair(col, 0u)) == pk->advice_queries.end()) {
        pk->advice_queries.push_back({col, 0u});
        per_col[col]++;
      }
    }
    off += w;
    pk->lookups.push_back(d);
  }
  // blinding_factors (plonk/circuit.rs:2022-2047)
  uint32_t factors = 1;
  for (uint32_t v : per_col) factors = std::max(factors, v);
  if (per_col.empty()) factors = 1;
  factors = std::max(3u, factors);
  pk->bf = factors + 2;
  const size_t n = (size_t)1 << pk->k;
  if (n < (size_t)pk->bf + 3) {  // minimum_rows (circuit.rs:2051-2059)
    delete pk;
    return c->fail(CQ_ERR_ARG, "pk: not enough rows available");
  }
  pk->u = (uint32_t)(n - (pk->bf + 1));
  int rc;
  // degree 3 = max(3, 2 + input degree 1) (static_lookup.rs:181-190)
  if ((rc = domain_create(c, 3, pk->k, &pk->domain)) != CQ_OK) {
    delete pk;
    return rc;
  }
  // l_active_row = 1 - (l_last + l_blind) on the extended coset (keygen.rs:344-373); by linearity it is
  // the coset extension of the indicator of the usable rows
  const size_t ext = pk->domain->ext();
  if (hipMalloc(&pk->l_active_row, ext * sizeof(Fr)) != hipSuccess) return c->fail(CQ_ERR_HIP, "hipMalloc(l_active_row)");
  void* tmp;
  if ((rc = c->ensure_scratch(1, n * sizeof(Fr), &tmp)) != CQ_OK) return rc;
  if ((rc = poly_fill_usable_rows(c, (Fr*)tmp, (uint32_t)n, pk->u)) != CQ_OK) return rc;
  if ((rc = domain_lagrange_to_coeff(pk->domain, (Fr*)tmp, (Fr*)tmp, 1, n, n)) != CQ_OK) return rc;
  if ((rc = domain_coeff_to_extended(pk->domain, (Fr*)tmp, pk->l_active_row, 1, n, ext)) != CQ_OK) return rc;
  // b0_g1_bound: n-1 points (best_multiexp asserts equal lengths, arithmetic.rs:133)
  if (b0_on_device) {
    pk->b0_g1_bound = (G1Affine*)b0_g1_bound;
  } else {
    if (hipMalloc(&pk->b0_g1_bound, (n - 1) * sizeof(G1Affine)) != hipSuccess) return c->fail(CQ_ERR_HIP, "hipMalloc(b0 bound)");
    pk->own_b0 = true;
    CQ_HIP(c, hipMemcpyAsync(pk->b0_g1_bound, b0_g1_bound, (n - 1) * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  }
  // per lookup: [qs_0 | qs_1 | ...] so that q_a is one MSM
  for (auto& lk : pk->lookups) {
    G1Affine* cat = nullptr;
    const size_t N = cfg->N;
    if (hipMalloc(&cat, lk.tables.size() * N * sizeof(G1Affine)) != hipSuccess) return c->fail(CQ_ERR_HIP, "hipMalloc(qs)");
    for (size_t j = 0; j < lk.tables.size(); j++)
      CQ_HIP(c, hipMemcpyAsync(cat + j * N, lk.tables[j]->qs, N * sizeof(G1Affine), hipMemcpyDeviceToDevice, c->stream));
    pk->qs_concat.push_back(cat);
    if (c->msm_precompute && (rc = msm_register_tables(c, cat, lk.tables.size() * N)) != CQ_OK) return rc;
  }
  if (c->msm_precompute && (rc = msm_register_tables(c, pk->b0_g1_bound, n - 1)) != CQ_OK) return rc;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = pk;
  return CQ_OK;
}

int cq_pk_set_sharding(cq_pk* pk, uint32_t rank, uint32_t world, cq_allgather_fn fn, void* user) {
  if (!pk || world == 0 || rank >= world || (world > 1 && !fn)) return CQ_ERR_ARG;
  pk->shard_rank = rank;
  pk->shard_world = world;
  pk->allgather = fn;
  pk->allgather_user = user;
  return CQ_OK;
}

void cq_pk_destroy(cq_pk* pk) {
  if (!pk) return;
  hipStreamSynchronize(pk->ctx->stream);
  if (pk->domain) domain_destroy(pk->domain);
  if (pk->l_active_row) hipFree(pk->l_active_row);
  msm_unregister_tables(pk->ctx, pk->b0_g1_bound);
  if (pk->own_b0 && pk->b0_g1_bound) hipFree(pk->b0_g1_bound);
  for (auto p : pk->qs_concat) {
    msm_unregister_tables(pk->ctx, p);
    hipFree(p);
  }
  delete pk;
}

uint32_t cq_pk_usable_rows(const cq_pk* pk) { return pk ? pk->u : 0; }

size_t cq_pk_proof_size(const cq_pk* pk) {
  if (!pk) return 0;
  const size_t L = pk->lookups.size();
  const size_t points = pk->num_advice + 2 * L + 5 * L + 1 + pk->domain->quotient_poly_degree + 1;
  const size_t scalars = pk->advice_queries.size() + 1 + 3 * L;
  return 32 * (points + scalars);
}

int cq_create_proof(cq_pk* pk, const uint64_t* const* advice_dev, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                    size_t proof_cap, size_t* proof_len) {
  if (!pk || (!advice_dev && pk->num_advice) || !rng || !proof || !proof_len) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  std::vector<uint8_t> out;
  int rc = create_proof_dev(pk, advice_dev, rng, rng_state, out);
  if (rc != CQ_OK) return rc;
  if (out.size() > proof_cap) return c->fail(CQ_ERR_ARG, "proof buffer too small");
  memcpy(proof, out.data(), out.size());
  *proof_len = out.size();
  return CQ_OK;
}

int cq_create_proof_host(cq_pk* pk, const uint64_t* const* advice, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                         size_t proof_cap, size_t* proof_len) {
  if (!pk || (!advice && pk->num_advice)) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t n = (size_t)1 << pk->k;
  void* stage;
  int rc;
  if ((rc = c->ensure_scratch(7, (size_t)pk->num_advice * n * sizeof(Fr) + 64, &stage)) != CQ_OK) return rc;
  std::vector<const uint64_t*> ptrs(pk->num_advice);
  for (uint32_t a = 0; a < pk->num_advice; a++) {
    Fr* d = (Fr*)stage + (size_t)a * n;
    CQ_HIP(c, hipMemcpyAsync(d, advice[a], (size_t)pk->u * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
    ptrs[a] = (const uint64_t*)d;
  }
  return cq_create_proof(pk, ptrs.data(), rng, rng_state, proof, proof_cap, proof_len);
}

// ---- SHA witness fill --------------------------------------------------------------------------------
int cq_sha_witness_fill_dev(cq_ctx* c, const uint32_t* words_dev, size_t nwords, uint32_t pairs, size_t n,
                            uint64_t* const* cols_dev) {
  if (!c || !words_dev || !cols_dev || pairs == 0 || pairs > 8 || nwords > 0x3fffffffull || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  ShaCols sc;
  for (uint32_t i = 0; i < 16; i++) sc.p[i] = i < 2 * pairs ? (Fr*)cols_dev[i] : nullptr;
  return sha_witness_fill(c, words_dev, (uint32_t)nwords, pairs, (uint32_t)n, sc);
}

int cq_sha_spread_table_dev(cq_ctx* c, size_t size, uint64_t* dense_dev, uint64_t* spread_dev) {
  if (!c || !dense_dev || !spread_dev || size == 0 || size > 65536) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return sha_spread_table(c, (uint32_t)size, (Fr*)dense_dev, (Fr*)spread_dev);
}

// ---- sha/src/tables.rs generators ----------------------------------------------------------------------
int cq_sha_synthesis_table_dev(cq_ctx* c, int kind, uint32_t first_limb_len, uint32_t second_limb_len, uint64_t* out_dev) {
  if (!c || !out_dev || kind < 0 || kind > 3 || first_limb_len == 0 || second_limb_len == 0 ||
      first_limb_len + 2 * second_limb_len > 28)
    return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return sha_synthesis_table(c, (uint32_t)kind, first_limb_len, second_limb_len, out_dev);
}
int cq_sha_decomposition_table_dev(cq_ctx* c, uint32_t first_limb_len, uint32_t second_limb_len, uint32_t k_bits,
                                   uint64_t* out_dev) {
  if (!c || !out_dev || k_bits > 28 || first_limb_len == 0 || second_limb_len == 0) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return sha_decomposition_table(c, first_limb_len, second_limb_len, k_bits, out_dev);
}

// ---- StaticTableValues::new (static_lookup.rs:78-126): the reference's O(N^2) construction on the GPU ----
int cq_static_table_new(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t* srs_g1, cq_static_table** out) {
  if (!c || !values || !srs_g1 || !out || !is_pow2(size) || size > (1u << 20)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  int rc = table_alloc(c, size, values, out);
  if (rc != CQ_OK) return rc;
  cq_static_table* t = *out;
  const uint32_t N = (uint32_t)size;
  cq_domain* dom = nullptr;
  if ((rc = domain_create(c, 2, log2u(size), &dom)) != CQ_OK) return rc;
  const uint32_t group = 16;  // roots per MSM launch
  G1Affine* srs = nullptr;
  Fr *coeffs = nullptr, *quot = nullptr;
  if (hipMalloc(&srs, size * sizeof(G1Affine)) != hipSuccess || hipMalloc(&coeffs, size * sizeof(Fr)) != hipSuccess ||
      hipMalloc(&quot, (size_t)group * size * sizeof(Fr)) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(static_table_new)");
  CQ_HIP(c, hipMemcpyAsync(srs, srs_g1, size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  if (c->msm_precompute && (rc = msm_register_tables(c, srs, size)) != CQ_OK) return rc;
  if ((rc = domain_lagrange_to_coeff(dom, t->values, coeffs, 1, size, size)) != CQ_OK) return rc;  // :99-105
  std::vector<uint64_t> jac(group * 12);
  std::vector<G1Affine> host_qs(size);
  for (uint32_t first = 0; first < N; first += group) {
    const uint32_t cnt = std::min(group, N - first);
    if ((rc = cq_table_quotients(c, coeffs, N, dom->omega, dom->ifft_divisor, first, cnt, quot)) != CQ_OK) return rc;  // :111-115
    std::vector<const Fr*> sc(cnt);
    std::vector<const G1Affine*> bs(cnt, srs);
    for (uint32_t r = 0; r < cnt; r++) sc[r] = quot + (size_t)r * N;
    if ((rc = cq_msm_multi(c, sc.data(), bs.data(), N - 1, cnt, jac.data())) != CQ_OK) return rc;  // :117
    for (uint32_t r = 0; r < cnt; r++) {
      G1Jac q = {Fq::from_limbs64(jac.data() + 12 * r), Fq::from_limbs64(jac.data() + 12 * r + 4),
                 Fq::from_limbs64(jac.data() + 12 * r + 8)};
      host_qs[first + r] = jac_to_affine(q);
    }
  }
  CQ_HIP(c, hipMemcpyAsync(t->qs, host_qs.data(), size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  msm_unregister_tables(c, srs);
  hipFree(srs);
  hipFree(coeffs);
  hipFree(quot);
  domain_destroy(dom);
  return CQ_OK;
}

// ---- harness RNGs ----------------------------------------------------------------------------------------
void cq_xoshiro256ss_seed(uint64_t seed, uint64_t state[4]) {
  uint64_t z = seed;
  for (int i = 0; i < 4; i++) {  // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    state[i] = x ^ (x >> 31);
  }
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t cq_xoshiro256ss_next_u64(void* st) {
  uint64_t* s = (uint64_t*)st;
  const uint64_t result = rotl64(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}
uint64_t cq_buffer_rng_next_u64(void* st) {
  cq_buffer_rng* b = (cq_buffer_rng*)st;
  if (b->pos >= b->len) return 0;
  return b->words[b->pos++];
}

}  // extern "C"
