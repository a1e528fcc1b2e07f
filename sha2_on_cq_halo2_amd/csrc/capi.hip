// C ABI: context, device memory, arithmetic.rs entry points.  See include/cq_halo2.h.
#include "ctx.hpp"
#include "comm.hpp"
#include "msm.hpp"
#include "poly.hpp"
#include <cstring>

using namespace cq;

const NttTables* cq_ctx::tables_for(uint32_t log_n, const Fr& omega, int* rc) {
  for (auto& t : ntt_cache)
    if (t->log_n == log_n && t->omega == omega) return t.get();
  if (ntt_cache.size() >= 16) {
    hipStreamSynchronize(stream);
    ntt_cache.erase(ntt_cache.begin());
  }
  auto t = std::make_unique<NttTables>();
  if (t->build(log_n, omega, stream) != 0) {
    *rc = fail(CQ_ERR_HIP, "building NTT twiddle tables failed");
    return nullptr;
  }
  ntt_cache.push_back(std::move(t));
  return ntt_cache.back().get();
}

extern "C" {

const char* cq_version(void) { return "cq_halo2_amd 0.1 (gfx950)"; }

int cq_ctx_create(int device, void* hip_stream, cq_ctx** out) {
  if (!out) return CQ_ERR_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return CQ_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return CQ_ERR_HIP;
  cq_ctx* c = new cq_ctx();
  c->device = device;
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return CQ_ERR_HIP;
    }
    c->own_stream = true;
  }
  *out = c;
  return CQ_OK;
}

void cq_ctx_destroy(cq_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  for (cq_ctx* lane : c->lanes) cq_ctx_destroy(lane);
  c->lanes.clear();
  hipStreamSynchronize(c->stream);
  if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
  c->ntt_cache.clear();
  for (int i = 0; i < cq_ctx::NSCRATCH; i++)
    if (c->scratch[i]) hipFree(c->scratch[i]);
  if (c->pinned) hipHostFree(c->pinned);
  if (c->pinned_msm) hipHostFree(c->pinned_msm);
  if (c->pinned_small) hipHostFree(c->pinned_small);
  cq::comm_rccl_destroy(c);
  if (c->pinned_comm) hipHostFree(c->pinned_comm);
  if (c->comm_event) hipEventDestroy(c->comm_event);
  for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
  if (c->prof_entries) hipHostFree(c->prof_entries);
  if (c->copy_done) hipEventDestroy(c->copy_done);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  if (c->msm_tail_event) hipEventDestroy(c->msm_tail_event);
  if (c->aux_done) hipEventDestroy(c->aux_done);
  if (c->aux_stream) hipStreamDestroy(c->aux_stream);
  if (c->fb_table) hipFree(c->fb_table);
  for (auto& t : c->msm_tables) hipFree(t.table);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

const char* cq_last_error(const cq_ctx* c) { return c ? c->err.c_str() : "null context"; }

int cq_rccl_unique_id(uint8_t id[CQ_RCCL_UNIQUE_ID_BYTES]) {
  if (!id) return CQ_ERR_ARG;
  return cq::comm_rccl_unique_id(id);
}
int cq_ctx_comm_init_rccl(cq_ctx* c, uint32_t rank, uint32_t world, const uint8_t id[CQ_RCCL_UNIQUE_ID_BYTES]) {
  if (!c || !id || world == 0 || rank >= world) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return cq::comm_rccl_init(c, rank, world, id);
}
int cq_ctx_comm_selftest(cq_ctx* c) {
  if (!c) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return cq::comm_rccl_selftest(c);
}
int cq_ctx_comm_destroy(cq_ctx* c) {
  if (!c) return CQ_ERR_ARG;
  cq::comm_rccl_destroy(c);
  return CQ_OK;
}

int cq_ctx_set_hip_graphs(cq_ctx* c, int on) {
  if (!c) return CQ_ERR_ARG;
  c->graphs_on = on ? 1 : 0;
  return CQ_OK;
}

int cq_ctx_sync(cq_ctx* c) {
  if (!c) return CQ_ERR_ARG;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

void* cq_ctx_stream(cq_ctx* c) { return c ? (void*)c->stream : nullptr; }

int cq_dev_alloc(cq_ctx* c, size_t bytes, void** dptr) {
  if (!c || !dptr) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  CQ_HIP(c, hipMalloc(dptr, bytes ? bytes : 1));
  return CQ_OK;
}
int cq_dev_free(cq_ctx* c, void* dptr) {
  if (!c) return CQ_ERR_ARG;
  if (!dptr) return CQ_OK;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  msm_unregister_tables(c, dptr);  // window tables built from this array must not outlive it
  CQ_HIP(c, hipFree(dptr));
  return CQ_OK;
}
int cq_dev_upload(cq_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (!dst && bytes) || (!src && bytes)) return CQ_ERR_ARG;
  if (!bytes) return CQ_OK;
  CQ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}
int cq_dev_download(cq_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (!dst && bytes) || (!src && bytes)) return CQ_ERR_ARG;
  if (!bytes) return CQ_OK;
  CQ_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}
int cq_dev_memset(cq_ctx* c, void* dptr, int value, size_t bytes) {
  if (!c || (!dptr && bytes)) return CQ_ERR_ARG;
  if (!bytes) return CQ_OK;
  CQ_HIP(c, hipMemsetAsync(dptr, value, bytes, c->stream));
  return CQ_OK;
}

// ---- per-kernel timing (HIP events on the context's stream) ---------------------------------------
int cq_profile_enable(cq_ctx* c, int on) {
  if (!c) return CQ_ERR_ARG;
  if (on && !c->prof_entries) {
    hipError_t e = hipHostMalloc((void**)&c->prof_entries, cq_ctx::PROF_COUNTERS * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) return c->hip_fail(e, "hipHostMalloc(profile counters)");
  }
  c->prof_on = on != 0;
  return CQ_OK;
}

int cq_profile_read(cq_ctx* c, int id, double* total_ms, uint64_t* calls) {
  if (!c || !total_ms || !calls) return CQ_ERR_ARG;
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  double tot = 0;
  uint64_t n = 0;
  if (id == CQ_PROF_MSM_ENTRIES) {
    for (size_t i = 0; i < c->prof_entries_n; i++) n += c->prof_entries[i];
    c->prof_entries_n = 0;
    *total_ms = 0;
    *calls = n;
    return CQ_OK;
  }
  std::vector<cq_ctx::ProfSpan> keep;
  for (auto& sp : c->prof_spans) {
    if (sp.id != id) {
      keep.push_back(sp);
      continue;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      tot += ms;
      n++;
    }
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  c->prof_spans.swap(keep);
  *total_ms = tot;
  *calls = n;
  return CQ_OK;
}

// ---- best_fft -------------------------------------------------------------------------------
static int fft_dev(cq_ctx* c, const Fr* in, Fr* out, uint32_t log_n, const Fr& omega) {
  return domain_fft(c, in, out, log_n, omega, 1, (size_t)1 << log_n, (size_t)1 << log_n);
}

int cq_best_fft_dev(cq_ctx* c, const uint64_t* in_dev, uint64_t* out_dev, uint32_t log_n,
                    const uint64_t omega[4]) {
  if (!c || !in_dev || !out_dev || !omega || log_n > FR_S) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  return fft_dev(c, (const Fr*)in_dev, (Fr*)out_dev, log_n, Fr::from_limbs64(omega));
}

int cq_best_fft(cq_ctx* c, uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
  if (!c || !a || !omega || log_n > FR_S) return CQ_ERR_ARG;
  CQ_HIP(c, hipSetDevice(c->device));
  const size_t bytes = ((size_t)1 << log_n) * sizeof(Fr);
  void *din, *dout;
  int rc;
  if ((rc = c->ensure_scratch(1, bytes, &din)) != CQ_OK) return rc;
  if ((rc = c->ensure_scratch(2, bytes, &dout)) != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(din, a, bytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = fft_dev(c, (const Fr*)din, (Fr*)dout, log_n, Fr::from_limbs64(omega))) != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(a, dout, bytes, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  return CQ_OK;
}

}  // extern "C"
