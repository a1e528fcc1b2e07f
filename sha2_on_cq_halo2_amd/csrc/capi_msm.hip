// C ABI: best_multiexp and ParamsKZG::commit / commit_lagrange.  See include/cq_halo2.h.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "msm.hpp"
#include "g1fft.hpp"
#include "setup.hpp"
#include "cq.hpp"

using namespace cq;

uint32_t msm_table_window_bits(size_t n) {
  if (const char* e = getenv("CQ_TABLE_C")) {
    const long v = atol(e);
    if (v >= (long)MSM_TABLE_C_MIN && v <= (long)MSM_TABLE_C_MAX) return (uint32_t)v;
  }
  // Measured on MI355X, whole proofs of the SHA-shaped circuit (tools/table_c_sweep.sh; ms at c = 15 / 16 / 17 / 18 / 20):
  //   k = 18:  8.2 /  8.6 /  9.4 /  14.0 / 16.3      k = 20: 26.5 / 26.0 / 25.4 / 32.4 / 33.9      k = 22: 97.2 / 94.0 / 92.3 / 115 / 99
  // 17 bits: 15 instead of 17 additions per scalar, 2^16 buckets per MSM that the two-pass sort still reaches with a byte
  // per entry; beyond, the third sort pass and the 2^17..2^19-bucket reductions cost more than the additions saved.
  return n >= ((size_t)1 << 20) ? 17u : MSM_TABLE_C;
}

static uint32_t pick_c(cq_ctx* c, uint32_t n) {
  if (c->msm_c) return c->msm_c;
  return msm_window_bits(n);
}

// Runs `count` MSMs (each with its own base array and length); results to host Jacobians.
// Base arrays registered with cq_msm_precompute use their per-window tables (one bucket set per MSM,
// no window folding on the host) and may share a launch whatever their lengths; plain MSMs are
// grouped by equal length.
int msm_multi_begin(cq_ctx* c, const Fr* const* scalars, const G1Affine* const* bases, const size_t* lens, size_t count,
                    MsmPending& pend) {
  pend.launches.clear();
  pend.count = count;
  for (size_t j = 0; j < count; j++)
    if (lens[j] > 0x7fffffffull) return c->fail(CQ_ERR_ARG, "msm: len too large");
  // Tables of one width share a launch: where an array has several (a small table SRS next to SRS arrays of different
  // sizes), take the width of the longest MSM's table.
  uint32_t want_c = 0;
  {
    size_t longest = 0;
    for (size_t j = 0; j < count; j++)
      if (lens[j] > longest)
        if (const cq_ctx::MsmTable* t = c->find_msm_table(bases[j], lens[j])) { longest = lens[j]; want_c = t->c; }
  }
  // first pass: plan the launches and the result slots
  size_t done = 0, slots = 0;
  while (done < count) {
    MsmPending::Launch ln;
    ln.first = done;
    if (lens[done] == 0) {
      ln.batch = 1;
      ln.empty = true;
      pend.launches.push_back(ln);
      done++;
      continue;
    }
    const cq_ctx::MsmTable* t0 = c->find_msm_table(bases[done], lens[done], nullptr, want_c);
    const bool pre = t0 != nullptr;
    uint32_t nmax = (uint32_t)lens[done];
    uint32_t batch = 1;
    while (done + batch < count && batch < MSM_MAX_BATCH) {
      const size_t l = lens[done + batch];
      if (l == 0) break;
      const cq_ctx::MsmTable* t = c->find_msm_table(bases[done + batch], l, nullptr, want_c);
      if ((t != nullptr) != pre) break;
      if (pre && t->c != t0->c) break;
      if (!pre && l != lens[done]) break;
      nmax = std::max(nmax, (uint32_t)l);
      batch++;
    }
    const uint32_t cb = pre ? t0->c : pick_c(c, nmax);
    // keep the workspace below ~32 GiB (of 288)
    while (batch > 1 && MsmLayout(nmax, cb, batch, pre).total > ((size_t)32 << 30)) batch = (batch + 1) / 2;
    ln.batch = batch;
    ln.pre = pre;
    ln.c = cb;
    ln.nmax = nmax;
    MsmLayout L(nmax, cb, batch, pre);
    ln.W = L.W;
    ln.Wb = L.Wb;
    ln.M = L.M;
    ln.cols = L.cols;
    ln.slot = slots;
    slots += (size_t)MSM_SET_POINTS * batch * L.Wb;  // bit-plane sums of every bucket set
    pend.launches.push_back(ln);
    done += batch;
  }
  void *wsums = nullptr, *host = nullptr;
  int rc;
  if (slots) {
    if ((rc = c->ensure_scratch(4, slots * sizeof(G1Jac), &wsums)) != CQ_OK) return rc;
    if ((rc = c->ensure_pinned_msm(slots * sizeof(G1Jac), &host)) != CQ_OK) return rc;
  }
  pend.host = host;
  pend.slots = slots;
  for (auto& ln : pend.launches) {
    if (ln.empty) continue;
    MsmLayout L(ln.nmax, ln.c, ln.batch, ln.pre);
    std::vector<const G1Affine*> bp(ln.batch);
    std::vector<size_t> strides(ln.batch, 0);
    for (uint32_t j = 0; j < ln.batch; j++) {
      size_t toff = 0;
      const cq_ctx::MsmTable* t = ln.pre ? c->find_msm_table(bases[ln.first + j], lens[ln.first + j], &toff, ln.c) : nullptr;
      bp[j] = ln.pre ? (const G1Affine*)t->table + toff : bases[ln.first + j];
      strides[j] = ln.pre ? t->n : 0;
    }
    void* ws;
    if ((rc = c->ensure_scratch(3, L.total, &ws)) != CQ_OK) return rc;
    int r = msm_run(c, scalars + ln.first, bp.data(), lens + ln.first, ln.nmax, ln.c, ln.batch, ln.pre, strides.data(), ws,
                    (G1Jac*)wsums + ln.slot);
    if (r != 0) return c->fail(CQ_ERR_HIP, "msm launch failed");
  }
  if (slots)
    CQ_HIP(c, hipMemcpyAsync(host, wsums, slots * sizeof(G1Jac), hipMemcpyDeviceToHost, c->stream));
  return CQ_OK;
}

int msm_multi_end(cq_ctx* c, MsmPending& pend, uint64_t* out_jac) {
  if (int wrc = c->wait(c->stream, "msm: waiting for the launch")) return wrc;
  // the host's share of the reduction (msm_set_value: ~33 group operations per bucket set), spread over the context's
  // worker threads when a launch has many sets
  struct Item { const MsmPending::Launch* ln; uint32_t j; };
  std::vector<Item> items;
  size_t sets = 0;
  for (auto& ln : pend.launches)
    for (uint32_t j = 0; j < ln.batch; j++) {
      items.push_back({&ln, j});
      if (!ln.empty) sets += ln.Wb;
    }
  const G1Jac* host = (const G1Jac*)pend.host;
  auto fold = [&](size_t i) {
    const MsmPending::Launch& ln = *items[i].ln;
    const uint32_t j = items[i].j;
    uint64_t* o = out_jac + (ln.first + j) * 12;
    if (ln.empty) {
      memset(o, 0, 12 * sizeof(uint64_t));
      return;
    }
    const G1Jac* res = host + ln.slot;
    G1Jac r = ln.pre ? msm_fold_sets(res + (size_t)MSM_SET_POINTS * j * ln.Wb, ln.Wb, ln.M, ln.cols)
                     : msm_fold_windows(res + (size_t)MSM_SET_POINTS * j * ln.W, ln.W, ln.c, ln.cols);
    r.x.to_limbs64(o);
    r.y.to_limbs64(o + 4);
    r.z.to_limbs64(o + 8);
  };
  if (sets > 6) {
    c->pool().parallel_for(items.size(), fold);
  } else {
    for (size_t i = 0; i < items.size(); i++) fold(i);
  }
  return CQ_OK;
}

int cq_msm_multi_v(cq_ctx* c, const Fr* const* scalars, const G1Affine* const* bases, const size_t* lens, size_t count,
                   uint64_t* out_jac) {
  MsmPending pend;
  int rc = msm_multi_begin(c, scalars, bases, lens, count, pend);
  if (rc != CQ_OK) return rc;
  return msm_multi_end(c, pend, out_jac);
}

int cq_msm_multi(cq_ctx* c, const Fr* const* scalars, const G1Affine* const* bases, size_t len, size_t count,
                 uint64_t* out_jac) {
  std::vector<size_t> lens(count, len);
  return cq_msm_multi_v(c, scalars, bases, lens.data(), count, out_jac);
}

void msm_unregister_tables(cq_ctx* c, const void* bases) {
  for (size_t i = 0; i < c->msm_tables.size();) {
    if (c->msm_tables[i].bases == bases) {
      hipFree(c->msm_tables[i].table);
      c->msm_tables.erase(c->msm_tables.begin() + i);
    } else {
      i++;
    }
  }
}

void msm_release_table(cq_ctx* c, const void* bases, size_t n, uint32_t c_bits) {
  for (size_t i = 0; i < c->msm_tables.size(); i++) {
    cq_ctx::MsmTable& t = c->msm_tables[i];
    if (t.bases != bases || t.n != n || t.c != c_bits) continue;
    if (--t.refs == 0) {
      hipFree(t.table);
      c->msm_tables.erase(c->msm_tables.begin() + i);
    }
    return;
  }
}

// Builds and registers per-window tables T[w][i] = 2^(c*w) * bases[i] for a device-resident base array.
// Memory: ceil(255/c) x n x 64 B (17 x the SRS for n >= 2^15) -- sized for 288 GB of HBM.
int msm_register_tables(cq_ctx* c, const G1Affine* bases, size_t n, uint32_t want_c, bool* held) {
  if (held) *held = false;
  if (!bases || n == 0 || n > (1u << 26)) return CQ_OK;  // nothing to do / unsupported: plain mode
  // The width comes from the array's own length (15 bits up to 2^19 points, 17 from 2^20 on) unless the caller of the
  // library (cq_msm_set_table_window) or of this function (a proving key bringing its small arrays to the width of its
  // SRS, so that its MSMs share launches) says otherwise -- never from what happened to be registered first.
  const uint32_t cb = want_c ? want_c : (c->msm_table_c ? c->msm_table_c : msm_table_window_bits(n));
  for (auto& t : c->msm_tables)
    if (t.bases == bases && t.n == n && t.c == cb) {  // the very entry: one more holder
      t.refs++;
      if (held) *held = true;
      return CQ_OK;
    }
  {
    size_t off = 0;
    const cq_ctx::MsmTable* t = c->find_msm_table(bases, n, &off, cb);
    if (t && (t->c == cb || !want_c)) return CQ_OK;  // served by a registered (larger) array's table
  }
  const uint32_t W = (255 + cb - 1) / cb;
  void* table = nullptr;
  if (hipMalloc(&table, (size_t)W * n * sizeof(G1Affine)) != hipSuccess) {
    (void)hipGetLastError();
    return CQ_OK;  // not enough memory: stay in plain mode
  }
  if (msm_precompute_tables(c, bases, (uint32_t)n, cb, (G1Affine*)table) != 0) {
    hipFree(table);
    return c->fail(CQ_ERR_HIP, "msm precompute failed");
  }
  c->msm_tables.push_back({bases, n, cb, table, 1});
  if (held) *held = true;
  return CQ_OK;
}

static int msm_batch(cq_ctx* c, const Fr* const* scalars, const G1Affine* bases, size_t len, size_t count,
                     uint64_t* out_jac) {
  std::vector<const G1Affine*> bp(count, bases);
  return cq_msm_multi(c, scalars, bp.data(), len, count, out_jac);
}

namespace {
// destroys a half-built cq_params on every failing return (CQ_HIP included); release() on success
struct ParamsGuard {
  cq_params* p;
  explicit ParamsGuard(cq_params* p_) : p(p_) {}
  ~ParamsGuard() {
    if (p) cq_params_destroy(p);
  }
  cq_params* release() {
    cq_params* q = p;
    p = nullptr;
    return q;
  }
};
cq_params* params_new(cq_ctx* c, uint32_t k) {
  cq_params* p = new cq_params();
  p->ctx = c;
  p->k = k;
  p->n = (size_t)1 << k;
  p->g = nullptr;
  p->g_lagrange = nullptr;
  return p;
}
}  // namespace

extern "C" {

int cq_msm_precompute_dev(cq_ctx* c, const uint64_t* bases_dev, size_t n) {
  if (!c || !bases_dev) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return msm_register_tables(c, (const G1Affine*)bases_dev, n);
}

int cq_msm_forget_dev(cq_ctx* c, const uint64_t* bases_dev) {
  if (!c || !bases_dev) return CQ_ERR_ARG;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  msm_unregister_tables(c, bases_dev);
  return CQ_OK;
}

int cq_msm_set_precompute(cq_ctx* c, int on) {
  if (!c) return CQ_ERR_ARG;
  c->msm_precompute = on != 0;
  return CQ_OK;
}

int cq_msm_set_table_window(cq_ctx* c, uint32_t bits) {
  if (!c || (bits != 0 && (bits < MSM_TABLE_C_MIN || bits > MSM_TABLE_C_MAX))) return CQ_ERR_ARG;
  c->msm_table_c = bits;
  return CQ_OK;
}

int cq_msm_table_width_dev(cq_ctx* c, const uint64_t* bases_dev, size_t n, uint32_t preferred_bits, uint32_t* bits) {
  if (!c || !bases_dev || !bits) return CQ_ERR_ARG;
  const cq_ctx::MsmTable* t = c->find_msm_table(bases_dev, n, nullptr, preferred_bits);
  *bits = t ? t->c : 0;
  return CQ_OK;
}

int cq_msm_set_window(cq_ctx* c, uint32_t bits) {
  if (!c || (bits != 0 && (bits < 2 || bits > 15))) return CQ_ERR_ARG;
  c->msm_c = bits;
  return CQ_OK;
}

int cq_best_multiexp_dev(cq_ctx* c, const uint64_t* coeffs_dev, const uint64_t* bases_dev, size_t len,
                         uint64_t out_jac[12]) {
  if (!c || !out_jac || (len && (!coeffs_dev || !bases_dev))) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  const Fr* sc = (const Fr*)coeffs_dev;
  return msm_batch(c, &sc, (const G1Affine*)bases_dev, len, 1, out_jac);
}

int cq_msm_batch_dev(cq_ctx* c, const uint64_t* const* coeffs_dev, const uint64_t* bases_dev, size_t len,
                     size_t count, uint64_t* out_jac) {
  if (!c || !out_jac || !coeffs_dev || (len && !bases_dev)) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return msm_batch(c, (const Fr* const*)coeffs_dev, (const G1Affine*)bases_dev, len, count, out_jac);
}

int cq_best_multiexp(cq_ctx* c, const uint64_t* coeffs, const uint64_t* bases, size_t len, uint64_t out_jac[12]) {
  if (!c || !out_jac || (len && (!coeffs || !bases))) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  if (len == 0) {
    memset(out_jac, 0, 12 * sizeof(uint64_t));
    return CQ_OK;
  }
  void *ds, *db;
  int rc;
  if ((rc = c->ensure_scratch(1, len * sizeof(Fr), &ds)) != CQ_OK) return rc;
  if ((rc = c->ensure_scratch(2, len * sizeof(G1Affine), &db)) != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(ds, coeffs, len * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipMemcpyAsync(db, bases, len * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream));
  const Fr* sc = (const Fr*)ds;
  return msm_batch(c, &sc, (const G1Affine*)db, len, 1, out_jac);
}

// ---- ParamsKZG ---------------------------------------------------------------------------------
int cq_params_create(cq_ctx* c, uint32_t k, const uint64_t* g, const uint64_t* g_lagrange, cq_params** out) {
  if (!c || !g || !g_lagrange || !out || k > FR_S) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_params* p = params_new(c, k);
  ParamsGuard guard(p);
  const size_t bytes = p->n * sizeof(G1Affine);
  hipError_t e;
  if ((e = hipMalloc(&p->g, bytes)) != hipSuccess || (e = hipMalloc(&p->g_lagrange, bytes)) != hipSuccess)
    return c->hip_fail(e, "hipMalloc(params)");
  CQ_HIP(c, hipMemcpyAsync(p->g, g, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipMemcpyAsync(p->g_lagrange, g_lagrange, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  if (c->msm_precompute) {
    int rc2;
    if ((rc2 = msm_register_tables(c, p->g, p->n)) != CQ_OK) return rc2;
    if ((rc2 = msm_register_tables(c, p->g_lagrange, p->n)) != CQ_OK) return rc2;
  }
  *out = guard.release();
  return CQ_OK;
}

/* ParamsKZG::setup_from_toxic_waste (kzg/commitment.rs:209-276), computed on the GPU */
int cq_params_setup_from_toxic_waste(cq_ctx* c, uint32_t k, const uint64_t s[4], cq_params** out) {
  if (!c || !s || !out || k > FR_S) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_params* p = params_new(c, k);
  ParamsGuard guard(p);
  const size_t bytes = p->n * sizeof(G1Affine);
  hipError_t e;
  if ((e = hipMalloc(&p->g, bytes)) != hipSuccess || (e = hipMalloc(&p->g_lagrange, bytes)) != hipSuccess)
    return c->hip_fail(e, "hipMalloc(params)");
  void* tmp;
  int rc;
  if ((rc = c->ensure_scratch(1, p->n * sizeof(Fr), &tmp)) != CQ_OK) return rc;
  rc = srs_powers_and_lagrange(c, k, Fr::from_limbs64(s), p->g, p->g_lagrange, (Fr*)tmp, nullptr);
  if (rc != CQ_OK) return rc;
  if (c->msm_precompute) {
    if ((rc = msm_register_tables(c, p->g, p->n)) != CQ_OK) return rc;
    if ((rc = msm_register_tables(c, p->g_lagrange, p->n)) != CQ_OK) return rc;
  }
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *out = guard.release();
  return CQ_OK;
}

/* scalars[i] * G for device-resident scalars (every SRS element has this form) */
int cq_fixed_base_mul_dev(cq_ctx* c, const uint64_t* scalars_dev, size_t n, uint64_t* out_affine_dev) {
  if (!c || (n && (!scalars_dev || !out_affine_dev)) || n > 0x7fffffffull) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return fixed_base_mul(c, (const Fr*)scalars_dev, (uint32_t)n, (G1Affine*)out_affine_dev);
}

/* ParamsKZG::read_custom with RawBytes / RawBytesUnchecked (kzg/commitment.rs:383-459):
 * k:u32 LE | n x 64 B g | n x 64 B g_lagrange | [128 B g2 | 128 B s_g2, ignored] -- straight into HBM */
int cq_params_read_raw(cq_ctx* c, const uint8_t* buf, size_t len, int checked, cq_params** out) {
  if (!c || !buf || !out || len < 4) return CQ_ERR_ARG;
  uint32_t k;
  memcpy(&k, buf, 4);
  if (k > FR_S) return c->fail(CQ_ERR_ARG, "params: k out of range");
  const size_t n = (size_t)1 << k;
  if (len < 4 + 2 * n * sizeof(G1Affine)) return c->fail(CQ_ERR_ARG, "params: buffer too short");
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_params* p = params_new(c, k);
  ParamsGuard guard(p);
  const size_t bytes = n * sizeof(G1Affine);
  if (hipMalloc(&p->g, bytes) != hipSuccess || hipMalloc(&p->g_lagrange, bytes) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(params)");
  CQ_HIP(c, hipMemcpyAsync(p->g, buf + 4, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipMemcpyAsync(p->g_lagrange, buf + 4 + bytes, bytes, hipMemcpyHostToDevice, c->stream));
  if (checked) {  // SerdeFormat::RawBytes: coordinates < q and on the curve
    void* tmp;
    int rc;
    if ((rc = c->ensure_scratch(1, 64, &tmp)) != CQ_OK) return rc;
    CQ_HIP(c, hipMemsetAsync(tmp, 0, 4, c->stream));
    if ((rc = g1_validate(c, p->g, (uint32_t)n, (uint32_t*)tmp)) != CQ_OK) return rc;
    if ((rc = g1_validate(c, p->g_lagrange, (uint32_t)n, (uint32_t*)tmp)) != CQ_OK) return rc;
    uint32_t bad = 0;
    CQ_HIP(c, hipMemcpyAsync(&bad, tmp, 4, hipMemcpyDeviceToHost, c->stream));
    CQ_HIP(c, hipStreamSynchronize(c->stream));
    if (bad) return c->fail(CQ_ERR_ARG, "params: invalid point encoding");
  }
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  if (c->msm_precompute) {
    int rc2;
    if ((rc2 = msm_register_tables(c, p->g, p->n)) != CQ_OK) return rc2;
    if ((rc2 = msm_register_tables(c, p->g_lagrange, p->n)) != CQ_OK) return rc2;
  }
  *out = guard.release();
  return CQ_OK;
}

int cq_g_to_lagrange_dev(cq_ctx* c, const uint64_t* g_dev, uint32_t k, uint64_t* g_lagrange_dev) {
  if (!c || !g_dev || !g_lagrange_dev || k > 26) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return g1_to_lagrange(c, (const G1Affine*)g_dev, k, (G1Affine*)g_lagrange_dev);
}

// ParamsKZG::downsize (kzg/commitment.rs:480-492)
int cq_params_downsize(cq_params* p, uint32_t k, cq_params** out) {
  if (!p || !out || k > p->k) return CQ_ERR_ARG;  // `assert!(k <= self.k)`
  cq_ctx* c = p->ctx;
  CQ_HIP(c, hipSetDevice(c->device));
  *out = nullptr;
  cq_params* q = params_new(c, k);
  ParamsGuard guard(q);
  const size_t bytes = q->n * sizeof(G1Affine);
  if (hipMalloc(&q->g, bytes) != hipSuccess || hipMalloc(&q->g_lagrange, bytes) != hipSuccess)
    return c->fail(CQ_ERR_HIP, "hipMalloc(params)");
  CQ_HIP(c, hipMemcpyAsync(q->g, p->g, bytes, hipMemcpyDeviceToDevice, c->stream));
  int rc = g1_to_lagrange(c, q->g, k, q->g_lagrange);
  if (rc != CQ_OK) return rc;
  if (c->msm_precompute) {
    if ((rc = msm_register_tables(c, q->g, q->n)) != CQ_OK) return rc;
    if ((rc = msm_register_tables(c, q->g_lagrange, q->n)) != CQ_OK) return rc;
  }
  *out = guard.release();
  return CQ_OK;
}

/* G1 part of ParamsKZG::write_custom(RawBytes) (commitment.rs:366-379): 4 + 128 n bytes; the caller
 * appends its g2 / s_g2. */
int cq_params_write_raw(cq_params* p, uint8_t* buf, size_t cap, size_t* written) {
  if (!p || !buf || !written) return CQ_ERR_ARG;
  cq_ctx* c = p->ctx;
  const size_t bytes = p->n * sizeof(G1Affine);
  if (cap < 4 + 2 * bytes) return c->fail(CQ_ERR_ARG, "params: output buffer too small");
  memcpy(buf, &p->k, 4);
  CQ_HIP(c, hipMemcpyAsync(buf + 4, p->g, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipMemcpyAsync(buf + 4 + bytes, p->g_lagrange, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  *written = 4 + 2 * bytes;
  return CQ_OK;
}

// A proving key refers to the params (and table config) it was built on for as long as it lives -- it counts itself as a
// user, shares their window tables, restores them when it stops sharding.  The documented order is keys first; a caller
// (or a garbage collector) that destroys the params first only marks them released, and the last key frees them.
void cq_params_destroy(cq_params* p) {
  if (!p) return;
  if (p->key_users > 0) {
    p->owner_released = true;
    return;
  }
  hipStreamSynchronize(p->ctx->stream);
  if (p->g) msm_unregister_tables(p->ctx, p->g);
  if (p->g_lagrange) msm_unregister_tables(p->ctx, p->g_lagrange);
  if (p->g) hipFree(p->g);
  if (p->g_lagrange) hipFree(p->g_lagrange);
  delete p;
}

const uint64_t* cq_params_g_dev(const cq_params* p) { return p ? (const uint64_t*)p->g : nullptr; }
const uint64_t* cq_params_g_lagrange_dev(const cq_params* p) { return p ? (const uint64_t*)p->g_lagrange : nullptr; }

static int commit_host(cq_params* p, const G1Affine* bases, const uint64_t* poly, size_t len, uint64_t out_jac[12]) {
  if (!p || !out_jac || (len && !poly)) return CQ_ERR_ARG;
  cq_ctx* c = p->ctx;
  if (len > p->n) return c->fail(CQ_ERR_ARG, "commit: polynomial longer than the SRS");
  CQ_HIP(c, hipSetDevice(c->device));
  if (len == 0) {
    memset(out_jac, 0, 12 * sizeof(uint64_t));
    return CQ_OK;
  }
  void* ds;
  int rc;
  if ((rc = c->ensure_scratch(1, len * sizeof(Fr), &ds)) != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(ds, poly, len * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  const Fr* sc = (const Fr*)ds;
  return msm_batch(c, &sc, bases, len, 1, out_jac);
}

int cq_commit(cq_params* p, const uint64_t* poly, size_t len, uint64_t out_jac[12]) {
  return commit_host(p, p ? p->g : nullptr, poly, len, out_jac);
}
int cq_commit_lagrange(cq_params* p, const uint64_t* poly, size_t len, uint64_t out_jac[12]) {
  return commit_host(p, p ? p->g_lagrange : nullptr, poly, len, out_jac);
}
int cq_commit_dev(cq_params* p, const uint64_t* poly_dev, size_t len, uint64_t out_jac[12]) {
  if (!p || len > p->n) return CQ_ERR_ARG;
  return cq_best_multiexp_dev(p->ctx, poly_dev, (const uint64_t*)p->g, len, out_jac);
}
int cq_commit_lagrange_dev(cq_params* p, const uint64_t* poly_dev, size_t len, uint64_t out_jac[12]) {
  if (!p || len > p->n) return CQ_ERR_ARG;
  return cq_best_multiexp_dev(p->ctx, poly_dev, (const uint64_t*)p->g_lagrange, len, out_jac);
}

}  // extern "C"

extern "C" {
/* sum of `count` Jacobian points (host memory): the local EC sum after an all-gather of per-rank MSM
 * partials (EC addition is not an RCCL reduction op). */
int cq_g1_sum(const uint64_t* jac_points, size_t count, uint64_t out_jac[12]) {
  if ((!jac_points && count) || !out_jac) return CQ_ERR_ARG;
  G1Jac acc = G1Jac::identity();
  for (size_t i = 0; i < count; i++) {
    const uint64_t* p = jac_points + 12 * i;
    G1Jac q = {Fq::from_limbs64(p), Fq::from_limbs64(p + 4), Fq::from_limbs64(p + 8)};
    acc = jac_add(acc, q);
  }
  acc.x.to_limbs64(out_jac);
  acc.y.to_limbs64(out_jac + 4);
  acc.z.to_limbs64(out_jac + 8);
  return CQ_OK;
}
/* `to_affine` (derive/curve.rs:399-412) on the host */
int cq_g1_to_affine(const uint64_t jac[12], uint64_t out_affine[8]) {
  if (!jac || !out_affine) return CQ_ERR_ARG;
  G1Jac q = {Fq::from_limbs64(jac), Fq::from_limbs64(jac + 4), Fq::from_limbs64(jac + 8)};
  G1Affine a = jac_to_affine(q);
  a.x.to_limbs64(out_affine);
  a.y.to_limbs64(out_affine + 4);
  return CQ_OK;
}
}
