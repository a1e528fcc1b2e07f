// Library context: one GPU, one stream, grow-only scratch buffers, cached NTT tables.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>
#include "../../include/cq_halo2.h"
#include "field.hpp"
#include "ntt.hpp"
#include "hostpool.hpp"

namespace cq {
int comm_rccl_wait(cq_ctx* c, hipStream_t stream, const char* what);  // comm.hip
}

struct cq_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  std::vector<std::unique_ptr<cq::NttTables>> ntt_cache;
  static constexpr int NSCRATCH = 10;
  void* scratch[NSCRATCH] = {};
  size_t scratch_bytes[NSCRATCH] = {};
  void* pinned = nullptr;  // small pinned host staging buffer
  size_t pinned_bytes = 0;
  uint32_t msm_c = 0;  // 0 = automatic window size
  uint32_t msm_table_c = 0;  // caller's override of the window width of precomputed tables (cq_msm_set_table_window); 0 = by array length
  bool msm_precompute = true;  // build per-window tables for resident SRS arrays
  void* fb_table = nullptr;  // fixed-base table d*2^(8j)*G (setup.hip)
  // precomputed MSM window tables, keyed by the base array they were derived from
  // (an array may have tables of several widths: a small table SRS shared by keys of different sizes; `refs` = holders)
  struct MsmTable { const void* bases; size_t n; uint32_t c; void* table; uint32_t refs; };
  std::vector<MsmTable> msm_tables;
  // `bases` may point anywhere inside a registered array (a rank's slice of the SRS); *offset = first point
  // Lanes: contexts of their own (stream, scratch, twiddle cache) on the same GPU, owned by `parent`, on which
  // cq_create_proof_batch proves several instances of one key at a time.  A lane resolves window tables (read-only)
  // and worker threads through its parent.
  cq_ctx* parent = nullptr;
  std::vector<cq_ctx*> lanes;
  // `want_c` != 0: a table of that window width if the range has one, any other otherwise
  const MsmTable* find_msm_table(const void* bases, size_t len, size_t* offset = nullptr, uint32_t want_c = 0) const {
    if (parent) return parent->find_msm_table(bases, len, offset, want_c);
    const MsmTable* any = nullptr;
    size_t any_off = 0;
    for (auto& t : msm_tables) {
      const char* b0 = (const char*)t.bases;
      const char* b = (const char*)bases;
      if (b < b0 || b >= b0 + t.n * 64) continue;
      const size_t off = (size_t)(b - b0) / 64;
      if ((size_t)(b - b0) % 64 || off + len > t.n) continue;
      if (!any) { any = &t; any_off = off; }
      if (want_c == 0 || t.c == want_c) {
        if (offset) *offset = off;
        return &t;
      }
    }
    if (any && offset) *offset = any_off;
    return any;
  }
  // optional per-kernel timing with HIP events on `stream` (bench.py's roofline leg)
  struct ProfSpan { int id; hipEvent_t a, b; };
  bool prof_on = false;
  std::vector<ProfSpan> prof_spans;
  // (point, digit) entries handed to msm_accumulate_kernel, one pinned counter per launch while profiling
  static constexpr size_t PROF_COUNTERS = 4096;
  uint32_t* prof_entries = nullptr;
  size_t prof_entries_n = 0;
  hipEvent_t prof_begin(int id) {
    if (!prof_on) return nullptr;
    ProfSpan sp;
    sp.id = id;
    hipEventCreate(&sp.a);
    hipEventCreate(&sp.b);
    hipEventRecord(sp.a, stream);
    prof_spans.push_back(sp);
    return sp.b;
  }
  void prof_end(hipEvent_t b) {
    if (b) hipEventRecord(b, stream);
  }

  // Host wait for `s` inside a proof: plain hipStreamSynchronize -- unless the context holds an RCCL communicator, whose
  // collectives may be anywhere in the queue: then the wait has a time-out and watches the communicator (comm.hip), so that
  // a rank whose peer failed returns an error instead of sitting in the driver for ever.
  int wait(hipStream_t s, const char* what = "stream wait") {
    if (rccl_comm) return cq::comm_rccl_wait(this, s, what);
    const hipError_t e = hipStreamSynchronize(s);
    return e == hipSuccess ? CQ_OK : hip_fail(e, what);
  }

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  int hip_fail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return CQ_ERR_HIP;
  }
  // grow-only scratch slot
  int ensure_scratch(int slot, size_t bytes, void** out) {
    if (scratch_bytes[slot] < bytes) {
      if (scratch[slot]) {
        hipStreamSynchronize(stream);
        hipFree(scratch[slot]);
        scratch[slot] = nullptr;
        scratch_bytes[slot] = 0;
      }
      size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
      hipError_t e = hipMalloc(&scratch[slot], want);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc(scratch)");
      scratch_bytes[slot] = want;
    }
    *out = scratch[slot];
    return CQ_OK;
  }
  int ensure_pinned(size_t bytes, void** out) {
    if (pinned_bytes < bytes) {
      if (pinned) {
        hipStreamSynchronize(stream);
        hipHostFree(pinned);
        pinned = nullptr;
        pinned_bytes = 0;
      }
      hipError_t e = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
      pinned_bytes = bytes;
    }
    *out = pinned;
    return CQ_OK;
  }
  // side stream for host->device uploads that should overlap kernels on `stream` (the DMA engine is otherwise
  // serialised behind whatever the main stream has queued); `copy_done` orders the main stream after an upload
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_done = nullptr;
  int ensure_copy_stream() {
    if (copy_stream) return CQ_OK;
    hipError_t e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate(copy)");
    e = hipEventCreateWithFlags(&copy_done, hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreate(copy)");
    return CQ_OK;
  }
  // Second compute stream (lowest priority).  An MSM ends in latency-bound kernels (partial-sum combines, row /
  // column sums, weighted sums) that leave most of the GPU idle; the prover uses that time for transforms whose
  // inputs are already fixed (prover.hip, AuxFork).  msm_run records `msm_tail_event` right after its accumulate
  // kernel; work on `aux_stream` starts there; `aux_done` orders the main stream after it.  While the prover has
  // `stream` pointed at the side stream, NTTs take their scratch from `ntt_scratch_slot` (a slot of their own).
  hipStream_t aux_stream = nullptr;
  hipEvent_t msm_tail_event = nullptr, aux_done = nullptr;
  uint64_t msm_tail_seq = 0;
  bool aux_pending = false;
  int ntt_scratch_slot = 0;
  int ensure_aux_stream() {
    if (aux_stream) return CQ_OK;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipError_t e = hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, least);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate(aux)");
    e = hipEventCreateWithFlags(&msm_tail_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&aux_done, hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreate(aux)");
    return CQ_OK;
  }
  // hipGraph cache (msm.hip): a segment of kernel launches whose arguments are a pure function of `sig` is captured once
  // (thread-local capture: other threads of the process go on calling HIP) and replayed with one hipGraphLaunch.  Any
  // failure of the graph API turns the cache off for the context and the segment is launched directly.
  // Off by default -- measured on ROCm 7.2 / MI355X the replay is no faster than the plain launches (k = 18: 8.8 ms either
  // way; k = 14: 3.4 ms with graphs, 3.25 without: the host is far enough ahead of the GPU that launch overhead is not
  // what a proof waits for) -- cq_ctx_set_hip_graphs(ctx, 1) or CQ_HIP_GRAPH=1 turns it on.
  struct GraphEntry { std::vector<uint64_t> sig; int segment; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  int graphs_on = -1;  // -1: not decided yet
  template <class F>
  int run_graph(const std::vector<uint64_t>& sig, int segment, F&& enqueue) {
    if (graphs_on < 0) {
      const char* e = getenv("CQ_HIP_GRAPH");
      graphs_on = (e && e[0] == '1') ? 1 : 0;
    }
    if (!graphs_on) return enqueue();
    for (auto& g : graphs)
      if (g.segment == segment && g.sig == sig) {
        if (hipGraphLaunch(g.exec, stream) == hipSuccess) return 0;
        graphs_on = 0;
        (void)hipGetLastError();
        return enqueue();
      }
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      graphs_on = 0;
      (void)hipGetLastError();
      return enqueue();
    }
    const int rc = enqueue();
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc != 0 || ec != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      if (graph) hipGraphDestroy(graph);
      graphs_on = 0;
      (void)hipGetLastError();
      return rc != 0 ? rc : enqueue();  // nothing ran during the capture
    }
    hipGraphDestroy(graph);
    if (graphs.size() >= 64) {  // shapes come and go (tests): start over rather than grow
      for (auto& g : graphs) hipGraphExecDestroy(g.exec);
      graphs.clear();
    }
    graphs.push_back({sig, segment, exec});
    if (hipGraphLaunch(exec, stream) == hipSuccess) return 0;
    graphs_on = 0;
    (void)hipGetLastError();
    return enqueue();
  }
  // worker threads for the host-side glue (hostpool.hpp); CQ_HOST_THREADS overrides the count (0 = none)
  std::unique_ptr<cq::HostPool> pool_;
  cq::HostPool& pool() {
    if (parent) return parent->pool();
    if (!pool_) {
      unsigned hw = std::thread::hardware_concurrency();
      unsigned w = hw > 3 ? (hw - 2 < 12 ? hw - 2 : 12) : 0;  // (12: a round-2 launch's twelve bucket sets fold in one go: 53 -> 37 us)
      if (const char* e = getenv("CQ_HOST_THREADS")) w = (unsigned)atoi(e);
      pool_.reset(new cq::HostPool(w));
    }
    return *pool_;
  }
  // a page of pinned memory for the scalars a proof reads back (error flags, b(0), z values): a device-to-host copy into
  // pageable memory is staged and blocks the host for ~25 us each, into pinned memory it is a plain asynchronous DMA
  void* pinned_small = nullptr;
  static constexpr size_t PINNED_SMALL_BYTES = 16384;
  int ensure_pinned_small(void** out) {
    if (!pinned_small) {
      hipError_t e = hipHostMalloc(&pinned_small, PINNED_SMALL_BYTES, hipHostMallocDefault);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    }
    *out = pinned_small;
    return CQ_OK;
  }
  // RCCL communicator of the context (comm.hip; cq_ctx_comm_init_rccl) and the pinned staging of its small exchanges
  void* rccl_comm = nullptr;
  uint32_t rccl_rank = 0, rccl_world = 1;
  bool rccl_aborted = false;        // the communicator was given up after a failure (comm_rccl_abort)
  hipEvent_t comm_event = nullptr;  // polled by comm_rccl_wait
  void* pinned_comm = nullptr;
  size_t pinned_comm_bytes = 0;
  int ensure_pinned_comm(size_t bytes, void** out) {
    if (pinned_comm_bytes < bytes) {
      if (pinned_comm) {
        hipStreamSynchronize(stream);
        hipHostFree(pinned_comm);
        pinned_comm = nullptr;
        pinned_comm_bytes = 0;
      }
      const size_t want = bytes < 65536 ? 65536 : bytes;
      hipError_t e = hipHostMalloc(&pinned_comm, want, hipHostMallocDefault);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
      pinned_comm_bytes = want;
    }
    *out = pinned_comm;
    return CQ_OK;
  }
  void* pinned_msm = nullptr;  // MSM results (kept apart from `pinned`, which stages RNG words)
  size_t pinned_msm_bytes = 0;
  int ensure_pinned_msm(size_t bytes, void** out) {
    if (pinned_msm_bytes < bytes) {
      if (pinned_msm) {
        hipStreamSynchronize(stream);
        hipHostFree(pinned_msm);
        pinned_msm = nullptr;
        pinned_msm_bytes = 0;
      }
      size_t want = bytes < 65536 ? 65536 : bytes;
      hipError_t e = hipHostMalloc(&pinned_msm, want, hipHostMallocDefault);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
      pinned_msm_bytes = want;
    }
    *out = pinned_msm;
    return CQ_OK;
  }
  const cq::NttTables* tables_for(uint32_t log_n, const cq::Fr& omega, int* rc);
};

#define CQ_HIP(ctx, call)                                   \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return (ctx)->hip_fail(_e, #call); \
  } while (0)
