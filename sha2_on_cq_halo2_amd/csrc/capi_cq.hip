// C ABI: CQ table objects, proving key, create_proof, SHA witness fill, harness RNGs.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "cq.hpp"
#include "xoshiro.hpp"
#include "ctx.hpp"
#include "g1fft.hpp"
#include "plonk.hpp"
#include "prover.hpp"
#include "setup.hpp"
#include "msm.hpp"
#include "comm.hpp"

using namespace cq;

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static uint32_t log2u(size_t x) {
  uint32_t l = 0;
  while (((size_t)1 << (l + 1)) <= x) l++;
  return l;
}

// Owns a half-built object (and plain device allocations) until release(): every early return -- CQ_HIP included --
// destroys it, so a failing constructor leaks nothing and never hands back a live handle next to an error code.
namespace {
template <class T>
struct Building {
  T* p;
  void (*destroy)(T*);
  std::vector<void*> dev;       // temporaries, hipFree'd on every path
  std::vector<const void*> reg; // temporary MSM window tables, unregistered on every path
  cq_ctx* c = nullptr;
  cq_domain* dom = nullptr;
  Building(T* p_, void (*d)(T*), cq_ctx* c_) : p(p_), destroy(d), c(c_) {}
  ~Building() {
    if (c) hipStreamSynchronize(c->stream);
    for (const void* r : reg) msm_unregister_tables(c, r);
    for (void* d : dev) hipFree(d);
    if (dom) domain_destroy(dom);
    if (p) destroy(p);
  }
  T* release() {
    T* q = p;
    p = nullptr;
    return q;
  }
};
}  // namespace

extern "C" {

// ---- StaticTableConfig ---------------------------------------------------------------------------
int cq_table_config_create(cq_ctx* c, size_t size, const uint64_t* g1_lagrange, const uint64_t* opening_at_0,
                           cq_table_config** out) {
  if (!c || !g1_lagrange || !opening_at_0 || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_table_config* t = new cq_table_config();
  Building<cq_table_config> guard(t, cq_table_config_destroy, c);
  t->ctx = c;
  t->N = size;
  t->log_n = log2u(size);
  const size_t bytes = size * sizeof(G1Affine);
  if (hipMalloc(&t->g1_lagrange, bytes) != hipSuccess || hipMalloc(&t->g_lagrange_opening_at_0, bytes) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(table config)");
  CQ_HIP(c, hipMemcpyAsync(t->g1_lagrange, g1_lagrange, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipMemcpyAsync(t->g_lagrange_opening_at_0, opening_at_0, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  if (c->msm_precompute) {
    int rc2;
    if ((rc2 = msm_register_tables(c, t->g1_lagrange, size)) != CQ_OK) return rc2;
    if ((rc2 = msm_register_tables(c, t->g_lagrange_opening_at_0, size)) != CQ_OK) return rc2;
  }
  *out = guard.release();
  return CQ_OK;
}

int cq_table_config_setup_from_toxic_waste(cq_ctx* c, size_t size, const uint64_t s[4], cq_table_config** out) {
  if (!c || !s || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_table_config* t = new cq_table_config();
  Building<cq_table_config> guard(t, cq_table_config_destroy, c);
  t->ctx = c;
  t->N = size;
  t->log_n = log2u(size);
  const size_t bytes = size * sizeof(G1Affine);
  if (hipMalloc(&t->g1_lagrange, bytes) != hipSuccess || hipMalloc(&t->g_lagrange_opening_at_0, bytes) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(table config)");
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
  *out = guard.release();
  return CQ_OK;
}

void cq_table_config_destroy(cq_table_config* t) {
  if (!t) return;
  if (t->key_users > 0) {  // a key still uses it: the last one frees it (see cq_params_destroy)
    t->owner_released = true;
    return;
  }
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
// fills a fresh object the caller already guards (Building): nothing to undo here on failure
static int table_alloc(cq_ctx* c, size_t size, const uint64_t* values, cq_static_table* t) {
  t->ctx = c;
  t->N = size;
  if (hipMalloc(&t->values, size * sizeof(Fr)) != hipSuccess || hipMalloc(&t->qs, size * sizeof(G1Affine)) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(static table)");
  CQ_HIP(c, hipMemcpyAsync(t->values, values, size * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  return cq_table_build_index(c, t->values, (uint32_t)size, &t->slots, &t->nslots);
}

int cq_static_table_create(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t* qs_affine, cq_static_table** out) {
  if (!c || !values || !qs_affine || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;  // static_lookup.rs:80
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_static_table* t = new cq_static_table();
  Building<cq_static_table> guard(t, cq_static_table_destroy, c);
  int rc = table_alloc(c, size, values, t);
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(t->qs, qs_affine, size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = guard.release();
  return CQ_OK;
}

int cq_static_table_setup_from_toxic_waste(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t s[4],
                                           cq_static_table** out) {
  if (!c || !values || !s || !out || !is_pow2(size) || size > (1u << 28)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_static_table* t = new cq_static_table();
  Building<cq_static_table> guard(t, cq_static_table_destroy, c);
  int rc = table_alloc(c, size, values, t);
  if (rc != CQ_OK) return rc;
  // T(s): interpolate the values over the size-N domain, evaluate at s
  if ((rc = domain_create(c, 2, log2u(size), &guard.dom)) != CQ_OK) return rc;
  cq_domain* dom = guard.dom;
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
  *out = guard.release();
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
// `rc` is passed through
static int pk_abort(cq_pk*, int rc) { return rc; }  // the Building guard of pk_create_impl destroys the key

// Cursor over a ProvingKey::write byte stream (plonk.rs:349-362): big-endian u32 counts, raw 32-byte elements
namespace {
struct RawReader {
  const uint8_t* p = nullptr;
  size_t left = 0;
  bool ok = true;
  uint32_t be32() {
    if (left < 4) { ok = false; return 0; }
    const uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    p += 4;
    left -= 4;
    return v;
  }
  const uint8_t* take(size_t bytes) {
    if (left < bytes) { ok = false; return nullptr; }
    const uint8_t* q = p;
    p += bytes;
    left -= bytes;
    return q;
  }
};
void put_be32(uint8_t*& o, uint32_t v) {
  o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v;
  o += 4;
}
// unit vector at `row` on the extended coset (l0 / l_last, keygen.rs:340-363); tmp: n elements of scratch
int unit_coset(cq_pk* pk, size_t row, Fr* tmp, Fr* dst) {
  cq_ctx* c = pk->ctx;
  const size_t n = (size_t)1 << pk->k, ext = pk->domain->ext();
  const Fr one = Fr::one();
  CQ_HIP(c, hipMemsetAsync(tmp, 0, n * sizeof(Fr), c->stream));
  CQ_HIP(c, hipMemcpyAsync(tmp + row, &one, sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = domain_lagrange_to_coeff(pk->domain, tmp, tmp, 1, n, n)) != CQ_OK) return rc;
  return domain_coeff_to_extended(pk->domain, tmp, dst, 1, n, ext);
}
}  // namespace

// Window tables of the arrays a key multiplies over besides its SRS, at the width of the SRS tables (so that all MSMs of
// a round share one launch): the cached quotients and an own b0 bound are the key's; the table SRS may be shared with
// keys of other sizes and then carries a table per width (2^12 points: 4 MiB each).  The width follows the length of
// the point range a rank multiplies: 2^k / world.
static int pk_small_tables(cq_pk* pk) {
  cq_ctx* c = pk->ctx;
  const size_t n = (size_t)1 << pk->k;
  const uint32_t width = c->msm_table_c ? c->msm_table_c : msm_table_window_bits(n / std::max<uint32_t>(pk->shard_world, 1));
  if (width == pk->table_c && !pk->held_tables.empty()) return CQ_OK;
  pk->table_c = width;
  if (!c->msm_precompute) return CQ_OK;
  std::vector<cq_pk::TableRef> old;
  old.swap(pk->held_tables);
  int rc = CQ_OK;
  auto want = [&](const G1Affine* b, size_t len) {
    bool held = false;
    if (rc == CQ_OK) rc = msm_register_tables(c, b, len, pk->table_c, &held);
    if (held) pk->held_tables.push_back({b, len, pk->table_c});
  };
  const cq_table_config* cfg = pk->table_cfg;
  for (size_t l = 0; l < pk->qs_concat.size(); l++) want(pk->qs_concat[l], pk->lookups[l].tables.size() * cfg->N);
  if (cfg && !pk->lookups.empty()) {
    want(cfg->g1_lagrange, cfg->N);
    want(cfg->g_lagrange_opening_at_0, cfg->N);
  }
  if (pk->b0_g1_bound) want(pk->b0_g1_bound, n - 1);
  for (auto& t : old) msm_release_table(c, t.bases, t.n, t.c);  // after the new ones: a shared entry is never rebuilt in between
  return rc;
}

static int pk_create_impl(cq_ctx* c, cq_params* params, const cq_circuit* cs, cq_table_config* cfg, const uint64_t* b0_g1_bound,
                          int b0_on_device, const uint8_t* raw, size_t raw_len, uint32_t num_selectors, int checked, cq_pk** out) {
  if (!c || !params || !cs || !out) return CQ_ERR_ARG;
  if (cs->num_lookups && (!cfg || !b0_g1_bound)) return c->fail(CQ_ERR_ARG, "pk: static lookups need a table config and b0_g1_bound");
  if (cs->k != params->k) return c->fail(CQ_ERR_ARG, "pk: circuit k differs from params k");
  if (cs->num_lookups > CQ_MAX_LOOKUPS) return c->fail(CQ_ERR_ARG, "pk: too many lookups");
  CQ_HIP(c, hipSetDevice(c->device));
  const cq_plonk* pl = cs->plonk;
  *out = nullptr;
  cq_pk* pk = new cq_pk();
  Building<cq_pk> guard(pk, cq_pk_destroy, c);  // every failing return below (CQ_HIP included) destroys the key
  pk->ctx = c;
  pk->params = params;
  pk->k = cs->k;
  pk->num_advice = cs->num_advice;
  pk->table_cfg = cfg;
  pk->vk_repr = Fr::from_limbs64(cs->vk_repr);
  params->key_users++;
  if (cfg) cfg->key_users++;
  pk->counted_users = true;
  const size_t n = (size_t)1 << pk->k;
  size_t off = 0;
  for (uint32_t l = 0; l < cs->num_lookups; l++) {
    cq_lookup_desc d;
    const uint32_t w = cs->lookup_widths[l];
    if (w == 0 || w > CQ_MAX_WIDTH) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: lookup width out of range"));
    for (uint32_t j = 0; j < w; j++) {
      const uint32_t col = cs->lookup_columns[off + j];
      cq_static_table* t = cs->lookup_tables[off + j];
      const bool is_expr = pl && pl->lookup_input_program_lens;
      if ((!is_expr && col >= cs->num_advice) || !t) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: bad lookup column / table"));
      // "Tables should all be of the same size" (static_lookup/prover.rs:81-83)
      if (t->N != cfg->N) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: table size differs from the table config"));
      d.cols.push_back(col);
      d.prog.push_back(-1);
      d.tables.push_back(t);
      // query_advice_index (plonk/circuit.rs:1619-1633)
      if (!(pl && pl->num_advice_queries) && !is_expr &&
          std::find(pk->advice_queries.begin(), pk->advice_queries.end(), std::make_pair(col, (int32_t)0)) == pk->advice_queries.end())
        pk->advice_queries.push_back({col, 0});
    }
    off += w;
    pk->lookups.push_back(d);
  }
  if (pl) {
    pk->num_fixed = pl->num_fixed;
    pk->num_instance = pl->num_instance;
    pk->cs_degree = pl->cs_degree ? pl->cs_degree : 3;
    if (pk->cs_degree < 3 || pk->cs_degree - 2 > PERM_MAX_CHUNK) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: cs_degree out of range"));
    if (pl->num_perm_columns > PERM_MAX_COLUMNS) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: too many permutation columns"));
    if ((pl->num_fixed && !pl->fixed && !raw) || (pl->num_gate_polys && (!pl->gate_program_lens || !pl->gate_programs)) ||
        (pl->num_constants && !pl->constants) || (pl->num_perm_columns && (!pl->perm_column_kinds || !pl->perm_column_indices)))
      return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: null pointer in cq_plonk"));
    for (uint32_t q = 0; q < pl->num_advice_queries; q++) {
      if (pl->advice_query_columns[q] >= cs->num_advice) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: advice query column out of range"));
      pk->advice_queries.push_back({pl->advice_query_columns[q], pl->advice_query_rotations[q]});
    }
    for (uint32_t q = 0; q < pl->num_fixed_queries; q++) {
      if (pl->fixed_query_columns[q] >= pl->num_fixed) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: fixed query column out of range"));
      pk->fixed_queries.push_back({pl->fixed_query_columns[q], pl->fixed_query_rotations[q]});
    }
    // phases (circuit.rs advice_column_phase / challenge_phase)
    pk->advice_phase.assign(cs->num_advice, 0);
    for (uint32_t a = 0; a < cs->num_advice && pl->advice_column_phases; a++) pk->advice_phase[a] = pl->advice_column_phases[a];
    if (pl->num_challenges && !pl->challenge_phases) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: null pointer in cq_plonk"));
    pk->challenge_phase.assign(pl->challenge_phases, pl->challenge_phases + pl->num_challenges);
    for (uint8_t ph : pk->advice_phase) pk->num_phases = std::max<uint32_t>(pk->num_phases, ph + 1u);
    for (uint8_t ph : pk->challenge_phase) pk->num_phases = std::max<uint32_t>(pk->num_phases, ph + 1u);
    if (pk->num_phases > 3) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: the API only supports 3 phases (prover.rs:337)"));
    for (uint32_t q = 0; q < pl->num_perm_columns; q++) {
      const uint32_t kind = pl->perm_column_kinds[q], idx = pl->perm_column_indices[q];
      const uint32_t lim = kind == CQ_COL_ADVICE ? cs->num_advice : kind == CQ_COL_FIXED ? pl->num_fixed : kind == CQ_COL_INSTANCE ? pl->num_instance : 0;
      if (idx >= lim) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: permutation column out of range"));
      pk->perm_columns.push_back({kind, idx});
    }
  }
  // ---- expression programs (gates, legacy lookups, static lookup inputs): validated and uploaded ----
  int rc;
  if (pl && pl->num_gate_polys) {
    const char* why = nullptr;
    size_t words = 0;
    if (!gate_program_check(pl->gate_program_lens, pl->gate_programs, pl->num_gate_polys, pl->num_constants, cs->num_advice,
                            pl->num_fixed, pl->num_instance, pl->num_challenges, &why, &words))
      return pk_abort(pk, c->fail(CQ_ERR_ARG, why));
    std::vector<uint32_t> blob;
    size_t o = 0;
    for (uint32_t g = 0; g < pl->num_gate_polys; g++) {
      blob.push_back(pl->gate_program_lens[g]);
      blob.insert(blob.end(), pl->gate_programs + o, pl->gate_programs + o + pl->gate_program_lens[g]);
      o += pl->gate_program_lens[g];
    }
    pk->num_gate_polys = pl->num_gate_polys;
    if (hipMalloc(&pk->gate_prog, blob.size() * sizeof(uint32_t)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(gate program)"));
    CQ_HIP(c, hipMemcpy(pk->gate_prog, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (pl && pl->num_constants) {
    if (hipMalloc(&pk->constants, pl->num_constants * sizeof(Fr)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(gate constants)"));
    CQ_HIP(c, hipMemcpy(pk->constants, pl->constants, pl->num_constants * sizeof(Fr), hipMemcpyHostToDevice));
  }
  if (pl && pl->num_legacy_lookups) {
    if (!pl->legacy_lookup_widths || !pl->legacy_program_lens || !pl->legacy_programs) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: null pointer in cq_plonk"));
    size_t nprog = 0;
    for (uint32_t l = 0; l < pl->num_legacy_lookups; l++) {
      if (pl->legacy_lookup_widths[l] == 0 || pl->legacy_lookup_widths[l] > 64) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: legacy lookup width out of range"));
      nprog += 2 * (size_t)pl->legacy_lookup_widths[l];
    }
    const char* why = nullptr;
    size_t words = 0;
    if (!gate_program_check(pl->legacy_program_lens, pl->legacy_programs, (uint32_t)nprog, pl->num_constants, cs->num_advice, pl->num_fixed,
                            pl->num_instance, pl->num_challenges, &why, &words))
      return pk_abort(pk, c->fail(CQ_ERR_ARG, why));
    std::vector<uint32_t> blob;
    size_t o = 0, idx = 0;
    for (uint32_t l = 0; l < pl->num_legacy_lookups; l++) {
      cq_pk::LegacyLookup lk;
      lk.width = pl->legacy_lookup_widths[l];
      for (int side = 0; side < 2; side++) {
        (side ? lk.tab_off : lk.in_off) = blob.size();
        for (uint32_t j = 0; j < lk.width; j++, idx++) {
          blob.push_back(pl->legacy_program_lens[idx]);
          blob.insert(blob.end(), pl->legacy_programs + o, pl->legacy_programs + o + pl->legacy_program_lens[idx]);
          o += pl->legacy_program_lens[idx];
        }
      }
      pk->legacy.push_back(lk);
    }
    if (hipMalloc(&pk->legacy_prog, blob.size() * sizeof(uint32_t)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(legacy lookup programs)"));
    CQ_HIP(c, hipMemcpy(pk->legacy_prog, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (pl && pl->lookup_input_program_lens && off) {
    // static lookup inputs given as expressions: one single-polynomial program each
    if (!pl->lookup_input_programs) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: null pointer in cq_plonk"));
    const char* why = nullptr;
    size_t words = 0;
    if (!gate_program_check(pl->lookup_input_program_lens, pl->lookup_input_programs, (uint32_t)off, pl->num_constants,
                            cs->num_advice, pl->num_fixed, pl->num_instance, pl->num_challenges, &why, &words))
      return pk_abort(pk, c->fail(CQ_ERR_ARG, why));
    std::vector<uint32_t> blob;
    size_t o = 0, idx = 0;
    for (auto& lk : pk->lookups)
      for (size_t j = 0; j < lk.cols.size(); j++, idx++) {
        lk.prog[j] = (int64_t)blob.size();
        blob.push_back(pl->lookup_input_program_lens[idx]);
        blob.insert(blob.end(), pl->lookup_input_programs + o, pl->lookup_input_programs + o + pl->lookup_input_program_lens[idx]);
        o += pl->lookup_input_program_lens[idx];
      }
    pk->lookup_exprs = true;
    if (hipMalloc(&pk->lookup_prog, blob.size() * sizeof(uint32_t)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(lookup programs)"));
    CQ_HIP(c, hipMemcpy(pk->lookup_prog, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  // blinding_factors (plonk/circuit.rs:2022-2047)
  if (pl && pl->blinding_factors) {
    pk->bf = pl->blinding_factors;
  } else {
    std::vector<uint32_t> per_col(cs->num_advice, 0);
    for (auto& q : pk->advice_queries) per_col[q.first]++;
    uint32_t factors = 1;
    for (uint32_t v : per_col) factors = std::max(factors, v);
    pk->bf = std::max(3u, factors) + 2;
  }
  if (n < (size_t)pk->bf + 3)  // minimum_rows (circuit.rs:2051-2059)
    return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: not enough rows available"));
  pk->u = (uint32_t)(n - (pk->bf + 1));
  // extended domain for cs.degree(): 3 for permutation / static lookups (static_lookup.rs:181-190), more with gates
  if ((rc = domain_create(c, pk->cs_degree, pk->k, &pk->domain)) != CQ_OK) return pk_abort(pk, rc);
  const size_t ext = pk->domain->ext();
  void* tmp;
  if ((rc = c->ensure_scratch(1, n * sizeof(Fr), &tmp)) != CQ_OK) return pk_abort(pk, rc);
  // l_active_row = 1 - (l_last + l_blind) on the extended coset (keygen.rs:344-373); by linearity it is
  // the coset extension of the indicator of the usable rows
  if (hipMalloc(&pk->l_active_row, ext * sizeof(Fr)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(l_active_row)"));
  // ---- ProvingKey::read (plonk.rs:379-403): with a serialized key the polynomial data below is uploaded straight
  // into HBM instead of being recomputed.  The VerifyingKey part (k, fixed / permutation commitments, selector
  // bits; :92-113) is skipped: the prover only needs its shape, which the circuit description gives.
  RawReader rd;
  std::vector<std::pair<Fr*, size_t>> to_check;  // uploaded vectors to validate under SerdeFormat::RawBytes
  auto read_poly = [&](Fr* dst, size_t expect) -> int {
    const uint32_t len = rd.be32();
    const uint8_t* src = rd.take((size_t)len * sizeof(Fr));
    if (!rd.ok || len != expect) return c->fail(CQ_ERR_ARG, "pk: serialized polynomial has the wrong length");
    if (dst) {
      CQ_HIP(c, hipMemcpyAsync(dst, src, (size_t)len * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
      to_check.push_back({dst, len});
    }
    return CQ_OK;
  };
  auto read_slice = [&](Fr* dst, size_t count, size_t each) -> int {
    if (rd.be32() != count || !rd.ok) return c->fail(CQ_ERR_ARG, "pk: serialized key has a different number of polynomials");
    for (size_t i = 0; i < count; i++) {
      int r2 = read_poly(dst ? dst + i * each : nullptr, each);
      if (r2 != CQ_OK) return r2;
    }
    return CQ_OK;
  };
  if (raw) {
    rd.p = raw;
    rd.left = raw_len;
    const uint32_t rk = rd.be32(), nfix = rd.be32();
    const size_t nperm = pk->perm_columns.size();
    rd.take((size_t)nfix * sizeof(G1Affine) + nperm * sizeof(G1Affine) + (size_t)num_selectors * ((n + 7) / 8));
    if (!rd.ok || rk != pk->k || nfix != pk->num_fixed) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: serialized key does not match the circuit"));
  }
  if (raw) {
    // l0, l_last, l_active_row (keygen.rs:338-373) come first; a CQ-only prover reads only the last
    for (int which = 0; which < 2; which++) {
      Fr** dst = which ? &pk->l_last : &pk->l0;
      if (pk->general() && hipMalloc(dst, ext * sizeof(Fr)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(l0/l_last)"));
      if ((rc = read_poly(pk->general() ? *dst : nullptr, ext)) != CQ_OK) return pk_abort(pk, rc);
    }
    if ((rc = read_poly(pk->l_active_row, ext)) != CQ_OK) return pk_abort(pk, rc);
  } else {
    if ((rc = poly_fill_usable_rows(c, (Fr*)tmp, (uint32_t)n, pk->u)) != CQ_OK) return pk_abort(pk, rc);
    if ((rc = domain_lagrange_to_coeff(pk->domain, (Fr*)tmp, (Fr*)tmp, 1, n, n)) != CQ_OK) return pk_abort(pk, rc);
    if ((rc = domain_coeff_to_extended(pk->domain, (Fr*)tmp, pk->l_active_row, 1, n, ext)) != CQ_OK) return pk_abort(pk, rc);
    if (pk->general()) {
      // l0 (keygen.rs:340-345) and l_last (:357-363): unit vectors at rows 0 and n - bf - 1
      for (int which = 0; which < 2; which++) {
        Fr** dst = which ? &pk->l_last : &pk->l0;
        if (hipMalloc(dst, ext * sizeof(Fr)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(l0/l_last)"));
        if ((rc = unit_coset(pk, which ? n - pk->bf - 1 : 0, (Fr*)tmp, *dst)) != CQ_OK) return pk_abort(pk, rc);
      }
    }
  }
  if (pk->num_fixed) {
    const size_t F = pk->num_fixed;
    if (hipMalloc(&pk->fixed_values, F * n * sizeof(Fr)) != hipSuccess || hipMalloc(&pk->fixed_polys, F * n * sizeof(Fr)) != hipSuccess ||
        hipMalloc(&pk->fixed_cosets, F * ext * sizeof(Fr)) != hipSuccess)
      return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(fixed columns)"));
    if (raw) {
      if ((rc = read_slice(pk->fixed_values, F, n)) != CQ_OK || (rc = read_slice(pk->fixed_polys, F, n)) != CQ_OK ||
          (rc = read_slice(pk->fixed_cosets, F, ext)) != CQ_OK)
        return pk_abort(pk, rc);
    } else {
      for (size_t f = 0; f < F; f++)
        CQ_HIP(c, hipMemcpyAsync(pk->fixed_values + f * n, pl->fixed[f], n * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
      if ((rc = domain_lagrange_to_coeff(pk->domain, pk->fixed_values, pk->fixed_polys, (uint32_t)F, n, n)) != CQ_OK) return pk_abort(pk, rc);
      if ((rc = domain_coeff_to_extended(pk->domain, pk->fixed_polys, pk->fixed_cosets, (uint32_t)F, n, ext)) != CQ_OK) return pk_abort(pk, rc);
    }
  } else if (raw) {
    if ((rc = read_slice(nullptr, 0, n)) != CQ_OK || (rc = read_slice(nullptr, 0, n)) != CQ_OK || (rc = read_slice(nullptr, 0, ext)) != CQ_OK)
      return pk_abort(pk, rc);
  }
  if (!pk->perm_columns.empty()) {
    // permutation::keygen::Assembly::build_pk (permutation/keygen.rs:151-208)
    const size_t PC = pk->perm_columns.size();
    if (hipMalloc(&pk->omega_powers, n * sizeof(Fr)) != hipSuccess || hipMalloc(&pk->perm_values, PC * n * sizeof(Fr)) != hipSuccess ||
        hipMalloc(&pk->perm_polys, PC * n * sizeof(Fr)) != hipSuccess || hipMalloc(&pk->perm_cosets, PC * ext * sizeof(Fr)) != hipSuccess)
      return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(permutation key)"));
    if ((rc = fr_powers(c, pk->domain->omega, (uint32_t)n, pk->omega_powers)) != CQ_OK) return pk_abort(pk, rc);
    {
      const size_t hi = std::max<size_t>(ext / 256, 1);
      if (hipMalloc(&pk->ext_pow_lo, 256 * sizeof(Fr)) != hipSuccess || hipMalloc(&pk->ext_pow_hi, hi * sizeof(Fr)) != hipSuccess)
        return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(extended omega powers)"));
      if ((rc = fr_powers(c, pk->domain->extended_omega, 256, pk->ext_pow_lo)) != CQ_OK ||
          (rc = fr_powers(c, pk->domain->extended_omega.pow_u64(256), (uint32_t)hi, pk->ext_pow_hi)) != CQ_OK)
        return pk_abort(pk, rc);
    }
    if (raw) {  // permutation::ProvingKey::read (permutation.rs:124-134)
      if ((rc = read_slice(pk->perm_values, PC, n)) != CQ_OK || (rc = read_slice(pk->perm_polys, PC, n)) != CQ_OK ||
          (rc = read_slice(pk->perm_cosets, PC, ext)) != CQ_OK)
        return pk_abort(pk, rc);
    } else {
    std::vector<uint32_t> ident;
    const uint32_t* mapping = pl->perm_mapping;
    if (!mapping) {
      ident.resize(PC * n * 2);
      for (size_t col = 0; col < PC; col++)
        for (size_t r = 0; r < n; r++) {
          ident[(col * n + r) * 2] = (uint32_t)col;
          ident[(col * n + r) * 2 + 1] = (uint32_t)r;
        }
      mapping = ident.data();
    } else {
      for (size_t cell = 0; cell < PC * n; cell++)
        if (mapping[2 * cell] >= PC || mapping[2 * cell + 1] >= n) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: permutation mapping out of range"));
    }
    std::vector<Fr> dp(PC);
    const Fr delta = fr_from_raw(FR_DELTA_RAW);
    Fr cur = Fr::one();
    for (size_t col = 0; col < PC; col++) {
      dp[col] = cur;
      cur = cur * delta;
    }
    void* stage;
    if ((rc = c->ensure_scratch(2, PC * n * 2 * sizeof(uint32_t) + PC * sizeof(Fr), &stage)) != CQ_OK) return pk_abort(pk, rc);
    Fr* dp_dev = (Fr*)stage;
    uint32_t* map_dev = (uint32_t*)(dp_dev + PC);
    CQ_HIP(c, hipMemcpy(dp_dev, dp.data(), PC * sizeof(Fr), hipMemcpyHostToDevice));
    CQ_HIP(c, hipMemcpy(map_dev, mapping, PC * n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
    if ((rc = perm_sigma(c, map_dev, (uint32_t)PC, (uint32_t)n, pk->omega_powers, dp_dev, pk->perm_values)) != CQ_OK) return pk_abort(pk, rc);
    if ((rc = domain_lagrange_to_coeff(pk->domain, pk->perm_values, pk->perm_polys, (uint32_t)PC, n, n)) != CQ_OK) return pk_abort(pk, rc);
    if ((rc = domain_coeff_to_extended(pk->domain, pk->perm_polys, pk->perm_cosets, (uint32_t)PC, n, ext)) != CQ_OK) return pk_abort(pk, rc);
    }
  } else if (raw) {
    if ((rc = read_slice(nullptr, 0, n)) != CQ_OK || (rc = read_slice(nullptr, 0, n)) != CQ_OK || (rc = read_slice(nullptr, 0, ext)) != CQ_OK)
      return pk_abort(pk, rc);
  }
  if (raw && rd.left != 0) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: trailing bytes after the serialized key"));
  if (raw && checked) {  // SerdeFormat::RawBytes: every field element below the modulus (helpers.rs:62-79)
    void* flag;
    if ((rc = c->ensure_scratch(1, 64, &flag)) != CQ_OK) return pk_abort(pk, rc);
    CQ_HIP(c, hipMemsetAsync(flag, 0, 4, c->stream));
    for (auto& v : to_check)
      if ((rc = fr_validate(c, v.first, v.second, (uint32_t*)flag)) != CQ_OK) return pk_abort(pk, rc);
    uint32_t bad = 0;
    CQ_HIP(c, hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, c->stream));
    CQ_HIP(c, hipStreamSynchronize(c->stream));
    if (bad) return pk_abort(pk, c->fail(CQ_ERR_ARG, "pk: field element not below the modulus"));
  }
  // b0_g1_bound: n-1 points (best_multiexp asserts equal lengths, arithmetic.rs:133)
  if (b0_g1_bound) {
    if (b0_on_device) {
      pk->b0_g1_bound = (G1Affine*)b0_g1_bound;
    } else {
      if (hipMalloc(&pk->b0_g1_bound, (n - 1) * sizeof(G1Affine)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(b0 bound)"));
      pk->own_b0 = true;
      CQ_HIP(c, hipMemcpyAsync(pk->b0_g1_bound, b0_g1_bound, (n - 1) * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
    }
  }
  // per lookup: [qs_0 | qs_1 | ...] so that q_a is one MSM
  for (auto& lk : pk->lookups) {
    G1Affine* cat = nullptr;
    const size_t N = cfg->N;
    if (hipMalloc(&cat, lk.tables.size() * N * sizeof(G1Affine)) != hipSuccess) return pk_abort(pk, c->fail(CQ_ERR_HIP, "hipMalloc(qs)"));
    pk->qs_concat.push_back(cat);
    for (size_t j = 0; j < lk.tables.size(); j++)
      CQ_HIP(c, hipMemcpyAsync(cat + j * N, lk.tables[j]->qs, N * sizeof(G1Affine), hipMemcpyDeviceToDevice, c->stream));
  }
  if ((rc = pk_small_tables(pk)) != CQ_OK) return pk_abort(pk, rc);
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = guard.release();
  return CQ_OK;
}

int cq_pk_create(cq_ctx* c, cq_params* params, const cq_circuit* cs, cq_table_config* cfg, const uint64_t* b0_g1_bound,
                 int b0_on_device, cq_pk** out) {
  return pk_create_impl(c, params, cs, cfg, b0_g1_bound, b0_on_device, nullptr, 0, 0, 0, out);
}

int cq_pk_read_raw(cq_ctx* c, cq_params* params, const cq_circuit* cs, cq_table_config* cfg, const uint64_t* b0_g1_bound,
                   int b0_on_device, const uint8_t* buf, size_t len, uint32_t num_selectors, int checked, cq_pk** out) {
  if (!buf) return CQ_ERR_ARG;
  return pk_create_impl(c, params, cs, cfg, b0_g1_bound, b0_on_device, buf, len, num_selectors, checked, out);
}

size_t cq_pk_raw_size(const cq_pk* pk, uint32_t num_selectors) {
  if (!pk) return 0;
  const size_t n = (size_t)1 << pk->k, ext = pk->domain->ext(), F = pk->num_fixed, PC = pk->perm_columns.size();
  const size_t poly_n = 4 + n * sizeof(Fr), poly_e = 4 + ext * sizeof(Fr);
  return 8 + (F + PC) * sizeof(G1Affine) + (size_t)num_selectors * ((n + 7) / 8) + 3 * poly_e + 3 * 4 + F * (2 * poly_n + poly_e) + 3 * 4 +
         PC * (2 * poly_n + poly_e);
}

// ProvingKey::write, SerdeFormat::RawBytes (plonk.rs:349-362)
int cq_pk_write_raw(cq_pk* pk, const uint8_t* selector_bits, uint32_t num_selectors, uint8_t* buf, size_t cap, size_t* written) {
  if (!pk || !buf || !written || (num_selectors && !selector_bits)) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t n = (size_t)1 << pk->k, ext = pk->domain->ext(), F = pk->num_fixed, PC = pk->perm_columns.size();
  const size_t total = cq_pk_raw_size(pk, num_selectors);
  if (cap < total) return c->fail(CQ_ERR_ARG, "pk: output buffer too small");
  uint8_t* o = buf;
  // VerifyingKey::write (:92-113): k, fixed commitments, permutation commitments, packed selector bits
  put_be32(o, pk->k);
  put_be32(o, (uint32_t)F);
  std::vector<uint64_t> cm((F + PC) * 8 + 8);
  int rc = cq_pk_vk_commitments(pk, cm.data(), cm.data() + F * 8);
  if (rc != CQ_OK) return rc;
  memcpy(o, cm.data(), (F + PC) * sizeof(G1Affine));
  o += (F + PC) * sizeof(G1Affine);
  const size_t sel_bytes = (size_t)num_selectors * ((n + 7) / 8);
  if (sel_bytes) memcpy(o, selector_bits, sel_bytes);
  o += sel_bytes;
  auto write_poly = [&](const Fr* src, size_t len) -> int {
    put_be32(o, (uint32_t)len);
    CQ_HIP(c, hipMemcpyAsync(o, src, len * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    o += len * sizeof(Fr);
    return CQ_OK;
  };
  auto write_slice = [&](const Fr* src, size_t count, size_t each) -> int {
    put_be32(o, (uint32_t)count);
    for (size_t i = 0; i < count; i++) {
      int r2 = write_poly(src + i * each, each);
      if (r2 != CQ_OK) return r2;
    }
    return CQ_OK;
  };
  // l0, l_last (a CQ-only key does not keep them: computed here), l_active_row
  void* tmp;
  if ((rc = c->ensure_scratch(1, (n + 2 * ext) * sizeof(Fr), &tmp)) != CQ_OK) return rc;
  Fr* l0 = pk->l0;
  Fr* l_last = pk->l_last;
  if (!l0) {
    l0 = (Fr*)tmp + n;
    l_last = l0 + ext;
    if ((rc = unit_coset(pk, 0, (Fr*)tmp, l0)) != CQ_OK || (rc = unit_coset(pk, n - pk->bf - 1, (Fr*)tmp, l_last)) != CQ_OK) return rc;
  }
  if ((rc = write_poly(l0, ext)) != CQ_OK || (rc = write_poly(l_last, ext)) != CQ_OK || (rc = write_poly(pk->l_active_row, ext)) != CQ_OK) return rc;
  if ((rc = write_slice(pk->fixed_values, F, n)) != CQ_OK || (rc = write_slice(pk->fixed_polys, F, n)) != CQ_OK ||
      (rc = write_slice(pk->fixed_cosets, F, ext)) != CQ_OK)
    return rc;
  if ((rc = write_slice(pk->perm_values, PC, n)) != CQ_OK || (rc = write_slice(pk->perm_polys, PC, n)) != CQ_OK ||
      (rc = write_slice(pk->perm_cosets, PC, ext)) != CQ_OK)
    return rc;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *written = (size_t)(o - buf);
  return *written == total ? CQ_OK : c->fail(CQ_ERR_INTERNAL, "pk: serialized size mismatch");
}

int cq_pk_set_opener(cq_pk* pk, int opener) {
  if (!pk || (opener != CQ_OPENER_GWC && opener != CQ_OPENER_SHPLONK)) return CQ_ERR_ARG;
  pk->opener = opener;
  return CQ_OK;
}

int cq_pk_set_rng_fill(cq_pk* pk, cq_rng_fill_fn fill) {
  if (!pk) return CQ_ERR_ARG;
  pk->rng_fill = fill;
  return CQ_OK;
}

// MSM window tables of a sharded key: a rank only ever multiplies its slice of every (scalars, bases) range, so the
// 17 x SRS tables are built for those slices alone (the union of the slices over the MSMs that share an array: lengths
// n, n - 1, N, w N).  The whole-array tables of the SRS objects the key is built on are given up for that only while this
// key is their one user -- another key on the same params / table config keeps finding them, and the slices then resolve
// into them -- and are rebuilt when the key returns to world = 1 or is destroyed.
static int pk_shard_tables(cq_pk* pk) {
  cq_ctx* c = pk->ctx;
  for (auto& t : pk->shard_tables) msm_release_table(c, t.bases, t.n, t.c);
  pk->shard_tables.clear();
  if (!c->msm_precompute) return CQ_OK;
  const size_t n = (size_t)1 << pk->k, N = pk->table_cfg ? pk->table_cfg->N : 0;
  const bool has_cfg = pk->table_cfg && !pk->lookups.empty();
  int rc;
  if ((rc = pk_small_tables(pk)) != CQ_OK) return rc;  // the window width follows the length of a rank's point range
  if (pk->shard_world <= 1) {  // back to whole arrays
    if (pk->dropped_params_tables) {
      pk->dropped_params_tables = false;
      if ((rc = msm_register_tables(c, pk->params->g, n)) != CQ_OK || (rc = msm_register_tables(c, pk->params->g_lagrange, n)) != CQ_OK) return rc;
    }
    if (pk->dropped_cfg_tables) {
      pk->dropped_cfg_tables = false;
      if ((rc = msm_register_tables(c, pk->table_cfg->g1_lagrange, N)) != CQ_OK ||
          (rc = msm_register_tables(c, pk->table_cfg->g_lagrange_opening_at_0, N)) != CQ_OK)
        return rc;
    }
    return CQ_OK;
  }
  if (pk->params->key_users <= 1 && !pk->dropped_params_tables) {
    msm_unregister_tables(c, pk->params->g);
    msm_unregister_tables(c, pk->params->g_lagrange);
    pk->dropped_params_tables = true;
  }
  if (has_cfg && pk->table_cfg->key_users <= 1 && !pk->dropped_cfg_tables) {
    msm_unregister_tables(c, pk->table_cfg->g1_lagrange);
    msm_unregister_tables(c, pk->table_cfg->g_lagrange_opening_at_0);
    pk->dropped_cfg_tables = true;
  }
  struct Use { const G1Affine* b; size_t len; };
  std::vector<Use> uses{{pk->params->g_lagrange, n}, {pk->params->g, n}, {pk->params->g, n - 1}};
  if (has_cfg) {
    uses.push_back({pk->table_cfg->g1_lagrange, N});
    uses.push_back({pk->table_cfg->g_lagrange_opening_at_0, N});
  }
  if (pk->b0_g1_bound) uses.push_back({pk->b0_g1_bound, n - 1});
  // (the cached quotients [qs_0 | qs_1 | ...] are the key's own small arrays: their whole-array tables stay)
  std::vector<std::pair<const G1Affine*, const G1Affine*>> iv;
  for (auto& u : uses) {
    size_t lo, hi;
    shard_range(u.len, pk->shard_rank, pk->shard_world, lo, hi);
    if (hi > lo) iv.push_back({u.b + lo, u.b + hi});
  }
  std::sort(iv.begin(), iv.end());
  std::vector<std::pair<const G1Affine*, const G1Affine*>> merged;
  for (auto& x : iv) {
    if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
    else merged.push_back(x);
  }
  for (auto& m : merged) {
    bool held = false;
    const size_t len = (size_t)(m.second - m.first);
    if ((rc = msm_register_tables(c, m.first, len, pk->table_c, &held)) != CQ_OK) return rc;
    if (held) pk->shard_tables.push_back({m.first, len, pk->table_c});  // not held: served by a whole-array table that stayed
  }
  return CQ_OK;
}

int cq_pk_set_sharding(cq_pk* pk, uint32_t rank, uint32_t world, cq_allgather_fn fn, void* user) {
  if (!pk || world == 0 || rank >= world) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  if (world > 1 && !fn && (!c->rccl_comm || c->rccl_world != world || c->rccl_rank != rank))
    return c->fail(CQ_ERR_ARG, "set_sharding: no all-gather callback and no matching RCCL communicator on the context");
  CQ_HIP(c, hipSetDevice(c->device));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  pk->shard_rank = rank;
  pk->shard_world = world;
  // world = 1 over a one-rank RCCL communicator: still "sharded" -- every collective of the prover runs, over one rank
  pk->shard_single = world == 1 && !fn && c->rccl_comm && c->rccl_world == 1;
  pk->allgather = world > 1 ? fn : nullptr;
  pk->allgather_user = user;
  int rc = pk_shard_tables(pk);
  if (rc != CQ_OK) return rc;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

int cq_pk_set_column_sharding(cq_pk* pk, int on, cq_bcast_fn fn, void* user) {
  if (!pk) return CQ_ERR_ARG;
  pk->shard_columns = on != 0;
  pk->bcast = fn;
  pk->bcast_user = user;
  return CQ_OK;
}

int cq_pk_set_resident_sharding(cq_pk* pk, int on, cq_exchange_fn fn, void* user) {
  if (!pk) return CQ_ERR_ARG;
  pk->shard_resident = on != 0;
  pk->exchange = fn;
  pk->exchange_user = user;
  return CQ_OK;
}

void cq_pk_destroy(cq_pk* pk) {
  if (!pk) return;
  hipStreamSynchronize(pk->ctx->stream);
  if (pk->domain) domain_destroy(pk->domain);
  if (pk->l_active_row) hipFree(pk->l_active_row);
  for (void* p : {(void*)pk->fixed_values, (void*)pk->fixed_polys, (void*)pk->fixed_cosets, (void*)pk->l0, (void*)pk->l_last,
                  (void*)pk->gate_prog, (void*)pk->constants, (void*)pk->lookup_prog, (void*)pk->legacy_prog, (void*)pk->perm_values, (void*)pk->perm_polys, (void*)pk->perm_cosets,
                  (void*)pk->omega_powers, (void*)pk->ext_pow_lo, (void*)pk->ext_pow_hi})
    if (p) hipFree(p);
  // window tables: the slices of a sharded key go, the whole-array tables it had given up come back (their owners --
  // params, table config -- are still alive: a key never outlives them), its references are returned
  if (pk->shard_world > 1 || pk->dropped_params_tables || pk->dropped_cfg_tables) {
    pk->shard_world = 1;
    (void)pk_shard_tables(pk);
  }
  for (auto& t : pk->held_tables) msm_release_table(pk->ctx, t.bases, t.n, t.c);
  if (pk->own_b0 && pk->b0_g1_bound) {
    msm_unregister_tables(pk->ctx, pk->b0_g1_bound);
    hipFree(pk->b0_g1_bound);
  }
  for (auto p : pk->qs_concat) {
    msm_unregister_tables(pk->ctx, p);
    hipFree(p);
  }
  if (pk->counted_users) {
    cq_params* params = pk->params;
    cq_table_config* cfg = pk->table_cfg;
    if (--params->key_users == 0 && params->owner_released) cq_params_destroy(params);
    if (cfg && --cfg->key_users == 0 && cfg->owner_released) cq_table_config_destroy(cfg);
  }
  delete pk;
}

uint32_t cq_pk_usable_rows(const cq_pk* pk) { return pk ? pk->u : 0; }

size_t cq_pk_proof_size(const cq_pk* pk) {
  if (!pk) return 0;
  const size_t L = pk->lookups.size(), S = pk->perm_sets();
  // one opening witness per distinct evaluation point (gwc/prover.rs:42-91)
  std::vector<int32_t> rots{0};
  auto seen = [&](int32_t r) {
    if (std::find(rots.begin(), rots.end(), r) == rots.end()) rots.push_back(r);
  };
  for (auto& q : pk->advice_queries) seen(q.second);
  for (auto& q : pk->fixed_queries) seen(q.second);
  const size_t PL = pk->legacy.size();
  if (S || PL) seen(1);
  if (PL) seen(-1);
  if (S > 1) seen(-(int32_t)(pk->bf + 1));
  const size_t openings = pk->opener == CQ_OPENER_SHPLONK ? 2 : rots.size();
  const size_t points = pk->num_advice + 2 * L + 3 * PL + S + 5 * L + 1 + pk->domain->quotient_poly_degree + openings;
  const size_t scalars = pk->advice_queries.size() + pk->fixed_queries.size() + 1 + pk->perm_columns.size() + (S ? 3 * S - 1 : 0) + 3 * L + 5 * PL;
  return 32 * (points + scalars);
}

static int create_proof_any(cq_pk* pk, const uint64_t* const* advice_dev, const uint64_t* const* instances,
                            const size_t* instance_lens, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                            size_t proof_cap, size_t* proof_len, cq_phase_fn phase_fn = nullptr, void* phase_user = nullptr) {
  if (!pk || (!advice_dev && pk->num_advice) || !rng || !proof || !proof_len) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  // "InvalidInstances" (prover.rs:73-82)
  if (pk->num_instance && (!instances || !instance_lens)) return c->fail(CQ_ERR_ARG, "create_proof: instance columns missing");
  CQ_HIP(c, hipSetDevice(c->device));
  std::vector<uint8_t> out;
  if (pk->num_phases > 1 && !phase_fn) return c->fail(CQ_ERR_ARG, "create_proof: a multi-phase circuit needs cq_create_proof_phases");
  // `overrun` counts the words THIS proof asked for beyond the stream (a caller's struct may hold anything there)
  if (rng == cq_buffer_rng_next_u64) ((cq_buffer_rng*)rng_state)->overrun = 0;
  int rc = create_proof_dev(pk, advice_dev, instances, instance_lens, phase_fn, phase_user, rng, rng_state, out);
  // A sharded proof that fails on this rank (over RCCL) leaves its peers in, or heading for, a collective this rank will not
  // join: give up the communicator, so that they fail too (asynchronous error or their wait's time-out) instead of
  // hanging.  Not for the failures every rank runs into alike, at the same point of the same program, with nothing pending
  // between them -- a witness value missing from a table, an identity commitment, a bad argument: those just return.
  const bool rank_local = rc == CQ_ERR_HIP || rc == CQ_ERR_INTERNAL || rc == CQ_ERR_NO_DEVICE;
  if (rank_local && pk->sharded() && !pk->allgather && c->rccl_comm && !pk->shard_single) {
    const std::string keep = c->err;
    cq::comm_rccl_abort(c);
    c->err = keep + " (sharded proof: this rank's RCCL communicator was aborted so that its peers do not hang)";
  }
  // An exhausted stream reads as zeros: the proof would not be zero-knowledge, and an all-zero random polynomial also
  // trips the transcript's identity-commitment check -- both are reported as what they are.  Any other error keeps its code.
  if (rng == cq_buffer_rng_next_u64 && ((cq_buffer_rng*)rng_state)->overrun && (rc == CQ_OK || rc == CQ_ERR_TRANSCRIPT))
    return c->fail(CQ_ERR_ARG, "create_proof: the pre-drawn RNG stream ran out (blinding would be zero)");
  if (rc != CQ_OK) return rc;
  if (out.size() > proof_cap) return c->fail(CQ_ERR_ARG, "proof buffer too small");
  memcpy(proof, out.data(), out.size());
  *proof_len = out.size();
  return CQ_OK;
}

int cq_create_proof(cq_pk* pk, const uint64_t* const* advice_dev, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                    size_t proof_cap, size_t* proof_len) {
  return create_proof_any(pk, advice_dev, nullptr, nullptr, rng, rng_state, proof, proof_cap, proof_len);
}

static int create_proof_host_any(cq_pk* pk, const uint64_t* const* advice, const uint64_t* const* instances,
                                 const size_t* instance_lens, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
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
  return create_proof_any(pk, ptrs.data(), instances, instance_lens, rng, rng_state, proof, proof_cap, proof_len);
}

int cq_create_proof_host(cq_pk* pk, const uint64_t* const* advice, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                         size_t proof_cap, size_t* proof_len) {
  return create_proof_host_any(pk, advice, nullptr, nullptr, rng, rng_state, proof, proof_cap, proof_len);
}

int cq_create_proof_phases(cq_pk* pk, uint64_t* const* advice_dev, const uint64_t* const* instances, const size_t* instance_lens,
                           cq_phase_fn phase_fn, void* phase_user, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                           size_t proof_cap, size_t* proof_len) {
  return create_proof_any(pk, advice_dev, instances, instance_lens, rng, rng_state, proof, proof_cap, proof_len, phase_fn, phase_user);
}

int cq_create_proof_instances(cq_pk* pk, const uint64_t* const* advice, int advice_on_device, const uint64_t* const* instances,
                              const size_t* instance_lens, cq_rng_next_u64 rng, void* rng_state, uint8_t* proof,
                              size_t proof_cap, size_t* proof_len) {
  return advice_on_device ? create_proof_any(pk, advice, instances, instance_lens, rng, rng_state, proof, proof_cap, proof_len)
                          : create_proof_host_any(pk, advice, instances, instance_lens, rng, rng_state, proof, proof_cap, proof_len);
}

// Several instances of one circuit (independent witnesses, independent transcripts and RNG streams; the reference's
// `circuits: &[ConcreteCircuit]`, plonk/prover.rs:419-463, proves them under ONE transcript and asserts a single circuit,
// :413-417, so a batch here means independent proofs).  The proofs run on `lanes` contexts of their own on the same GPU,
// one host thread each: while one proof sits in the latency-bound tail of an MSM launch or waits for its host (transcript,
// normalisation), the others' kernels fill the chip -- BASELINE configs[4].  Byte for byte the proofs cq_create_proof
// gives one at a time.
int cq_create_proof_batch(cq_pk* pk, size_t count, const uint64_t* const* const* advice_dev, cq_rng_next_u64 rng,
                          void* const* rng_states, uint8_t* const* proofs, size_t proof_cap, size_t* proof_lens, uint32_t lanes) {
  if (!pk || (count && (!advice_dev || !rng || !rng_states || !proofs || !proof_lens))) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  if (pk->num_instance || pk->num_phases > 1) return c->fail(CQ_ERR_ARG, "create_proof_batch: circuits with instance columns or several phases go through cq_create_proof_instances / _phases");
  if (pk->sharded()) return c->fail(CQ_ERR_ARG, "create_proof_batch: the key is sharded across ranks");
  if (!count) return CQ_OK;
  CQ_HIP(c, hipSetDevice(c->device));
  if (lanes == 0) lanes = 3;
  lanes = (uint32_t)std::min<size_t>(lanes, count);
  while (c->lanes.size() < lanes) {
    cq_ctx* lane = nullptr;
    int rc = cq_ctx_create(c->device, nullptr, &lane);
    if (rc != CQ_OK) return c->fail(rc, "create_proof_batch: lane context");
    lane->parent = c;
    c->lanes.push_back(lane);
  }
  CQ_HIP(c, hipStreamSynchronize(c->stream));  // whatever filled the witnesses on the caller's stream is done
  (void)c->pool();  // the lanes share the parent's worker threads: created here, before several threads would race to
  std::atomic<size_t> next{0};
  std::atomic<int> first_rc{CQ_OK};
  std::mutex err_mu;
  std::string err;
  auto worker = [&](uint32_t li) {
    cq_ctx* lc = c->lanes[li];
    if (hipSetDevice(lc->device) != hipSuccess) {
      first_rc.store(CQ_ERR_HIP);
      return;
    }
    // the key as this lane sees it: same device data, this lane's context (stream, scratch, twiddle cache)
    cq_pk lane_pk = *pk;
    cq_domain lane_dom = *pk->domain;
    lane_dom.ctx = lc;
    lane_pk.ctx = lc;
    lane_pk.domain = &lane_dom;
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= count || first_rc.load() != CQ_OK) break;
      std::vector<uint8_t> out;
      if (rng == cq_buffer_rng_next_u64) ((cq_buffer_rng*)rng_states[i])->overrun = 0;
      int rc = create_proof_dev(&lane_pk, advice_dev[i], nullptr, nullptr, nullptr, nullptr, rng, rng_states[i], out);
      if ((rc == CQ_OK || rc == CQ_ERR_TRANSCRIPT) && rng == cq_buffer_rng_next_u64 && ((cq_buffer_rng*)rng_states[i])->overrun) {
        rc = CQ_ERR_ARG;
        lc->err = "create_proof: the pre-drawn RNG stream ran out (blinding would be zero)";
      }
      if (rc == CQ_OK && out.size() > proof_cap) {
        rc = CQ_ERR_ARG;
        lc->err = "proof buffer too small";
      }
      if (rc != CQ_OK) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (first_rc.load() == CQ_OK) {
          first_rc.store(rc);
          err = "proof " + std::to_string(i) + ": " + lc->err;
        }
        break;
      }
      memcpy(proofs[i], out.data(), out.size());
      proof_lens[i] = out.size();
    }
    hipStreamSynchronize(lc->stream);
  };
  std::vector<std::thread> th;
  for (uint32_t li = 1; li < lanes; li++) {
    try {
      th.emplace_back(worker, li);
    } catch (...) {  // no thread to be had: the remaining lanes' share falls to the ones that run
      break;
    }
  }
  worker(0);
  for (auto& t : th) t.join();
  if (first_rc.load() != CQ_OK) return c->fail(first_rc.load(), err);
  return CQ_OK;
}

// fixed_commitments (keygen.rs:247-250) and permutation::VerifyingKey::commitments (permutation/keygen.rs:115-149)
int cq_pk_vk_commitments(cq_pk* pk, uint64_t* fixed_commitments, uint64_t* permutation_commitments) {
  if (!pk) return CQ_ERR_ARG;
  cq_ctx* c = pk->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t n = (size_t)1 << pk->k;
  for (int which = 0; which < 2; which++) {
    const size_t cnt = which ? pk->perm_columns.size() : pk->num_fixed;
    uint64_t* dst = which ? permutation_commitments : fixed_commitments;
    const Fr* src = which ? pk->perm_values : pk->fixed_values;
    if (!cnt) continue;
    if (!dst) return c->fail(CQ_ERR_ARG, "vk commitments: null output");
    for (size_t i = 0; i < cnt; i++) {
      uint64_t jac[12];
      int rc = cq_commit_lagrange_dev(pk->params, (const uint64_t*)(src + i * n), n, jac);
      if (rc != CQ_OK) return rc;
      if ((rc = cq_g1_to_affine(jac, dst + 8 * i)) != CQ_OK) return rc;
    }
  }
  return CQ_OK;
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
  *out = nullptr;
  cq_static_table* t = new cq_static_table();
  Building<cq_static_table> guard(t, cq_static_table_destroy, c);
  int rc = table_alloc(c, size, values, t);
  if (rc != CQ_OK) return rc;
  const uint32_t N = (uint32_t)size;
  if ((rc = domain_create(c, 2, log2u(size), &guard.dom)) != CQ_OK) return rc;
  cq_domain* dom = guard.dom;
  const uint32_t group = 16;  // roots per MSM launch
  G1Affine* srs = nullptr;
  Fr *coeffs = nullptr, *quot = nullptr;
  const bool alloc_ok = hipMalloc(&srs, size * sizeof(G1Affine)) == hipSuccess && hipMalloc(&coeffs, size * sizeof(Fr)) == hipSuccess &&
                        hipMalloc(&quot, (size_t)group * size * sizeof(Fr)) == hipSuccess;
  for (void* d : {(void*)srs, (void*)coeffs, (void*)quot})
    if (d) guard.dev.push_back(d);
  if (!alloc_ok) return c->fail(CQ_ERR_HIP, "hipMalloc(static_table_new)");
  CQ_HIP(c, hipMemcpyAsync(srs, srs_g1, size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  guard.reg.push_back(srs);  // unregistering an array that never got tables is a no-op
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
  *out = guard.release();
  return CQ_OK;
}

int cq_static_table_new_fk(cq_ctx* c, size_t size, const uint64_t* values, const uint64_t* srs_g1, cq_static_table** out) {
  if (!c || !values || !srs_g1 || !out || !is_pow2(size) || size < 2 || size > (1u << 22)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_static_table* t = new cq_static_table();
  Building<cq_static_table> guard(t, cq_static_table_destroy, c);
  int rc = table_alloc(c, size, values, t);
  if (rc != CQ_OK) return rc;
  if ((rc = domain_create(c, 2, log2u(size), &guard.dom)) != CQ_OK) return rc;
  G1Affine* srs = nullptr;
  Fr* coeffs = nullptr;
  const bool alloc_ok = hipMalloc(&srs, size * sizeof(G1Affine)) == hipSuccess && hipMalloc(&coeffs, size * sizeof(Fr)) == hipSuccess;
  for (void* d : {(void*)srs, (void*)coeffs})
    if (d) guard.dev.push_back(d);
  if (!alloc_ok) return c->fail(CQ_ERR_HIP, "hipMalloc(static_table_new_fk)");
  CQ_HIP(c, hipMemcpyAsync(srs, srs_g1, size * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  if ((rc = domain_lagrange_to_coeff(guard.dom, t->values, coeffs, 1, size, size)) != CQ_OK) return rc;  // :99-105
  if ((rc = fk_table_quotients(c, coeffs, srs, log2u(size), t->qs)) != CQ_OK) return rc;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = guard.release();
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
void cq_xoshiro256ss_fill(uint64_t state[4], uint64_t* dst, size_t count, uint32_t threads) {
  cq::xoshiro_fill(state, dst, count, threads);
}
uint64_t cq_buffer_rng_next_u64(void* st) {
  cq_buffer_rng* b = (cq_buffer_rng*)st;
  if (b->pos >= b->len) {
    b->overrun++;
    return 0;
  }
  return b->words[b->pos++];
}
// not compared against by the prover: an "unknown" generator as a caller's RngCore is
uint64_t cq_opaque_rng_next_u64(void* st) { return cq_xoshiro256ss_next_u64(st); }
void cq_opaque_rng_fill(void* st, uint64_t* dst, size_t count) { cq::xoshiro_fill((uint64_t*)st, dst, count, 8); }

}  // extern "C"

// ---- permutation::keygen::Assembly (permutation/keygen.rs:14-113), host bookkeeping ---------------------------
void cq_permutation_assembly_init(uint32_t columns, uint32_t n, uint32_t* mapping, uint32_t* aux, uint32_t* sizes) {
  // every cell starts as its own 1-cycle: mapping == aux == identity (:22-41)
  for (uint32_t col = 0; col < columns; col++)
    for (uint32_t r = 0; r < n; r++) {
      const size_t cell = (size_t)col * n + r;
      mapping[2 * cell] = aux[2 * cell] = col;
      mapping[2 * cell + 1] = aux[2 * cell + 1] = r;
      sizes[cell] = 1;
    }
}

int cq_permutation_assembly_copy(uint32_t columns, uint32_t n, uint32_t* mapping, uint32_t* aux, uint32_t* sizes,
                                 uint32_t left_column, uint32_t left_row, uint32_t right_column, uint32_t right_row) {
  if (!mapping || !aux || !sizes) return CQ_ERR_ARG;
  if (left_column >= columns || right_column >= columns) return CQ_ERR_ARG;  // Error::ColumnNotInPermutation
  if (left_row >= n || right_row >= n) return CQ_ERR_ARG;                    // Error::BoundsFailure (:62-66)
  auto cell = [&](uint32_t col, uint32_t row) { return (size_t)col * n + row; };
  size_t left = cell(left_column, left_row), right = cell(right_column, right_row);
  size_t left_cycle = cell(aux[2 * left], aux[2 * left + 1]);
  size_t right_cycle = cell(aux[2 * right], aux[2 * right + 1]);
  if (left_cycle == right_cycle) return CQ_OK;  // already in the same cycle (:75-77)
  if (sizes[left_cycle] < sizes[right_cycle]) std::swap(left_cycle, right_cycle);
  // merge the right cycle into the left one (:83-93)
  sizes[left_cycle] += sizes[right_cycle];
  const uint32_t lc_col = (uint32_t)(left_cycle / n), lc_row = (uint32_t)(left_cycle % n);
  size_t i = right_cycle;
  do {
    aux[2 * i] = lc_col;
    aux[2 * i + 1] = lc_row;
    i = cell(mapping[2 * i], mapping[2 * i + 1]);
  } while (i != right_cycle);
  std::swap(mapping[2 * left], mapping[2 * right]);  // :95-97
  std::swap(mapping[2 * left + 1], mapping[2 * right + 1]);
  return CQ_OK;
}
