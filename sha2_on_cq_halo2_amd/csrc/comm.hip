// See comm.hpp.  RCCL is reached through dlopen/dlsym only.
#include "comm.hpp"
#include <dlfcn.h>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <rccl/rccl.h>  // types and prototypes only: nothing here is linked against librccl
#include "cq.hpp"
#include "ctx.hpp"

namespace cq {

namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;                  // optional: without them a stuck collective is only bounded
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // by the caller's own watchdog
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  bool ok = false;
};

// The process may already hold a copy (PyTorch ships its own librccl.so with SONAME librccl.so.1): RTLD_NOLOAD finds it,
// so that the communicator lives in the same RCCL -- and the same HIP runtime -- as the host application's streams.
const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, []() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) return;
#define CQ_RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name))
    CQ_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    CQ_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    CQ_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    CQ_RCCL_SYM(AllGather, "ncclAllGather");
    CQ_RCCL_SYM(Broadcast, "ncclBroadcast");
    CQ_RCCL_SYM(GroupStart, "ncclGroupStart");
    CQ_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    CQ_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    CQ_RCCL_SYM(CommAbort, "ncclCommAbort");
    CQ_RCCL_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    CQ_RCCL_SYM(Send, "ncclSend");
    CQ_RCCL_SYM(Recv, "ncclRecv");
#undef CQ_RCCL_SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.Broadcast && api.GroupStart &&
             api.GroupEnd && api.GetErrorString;
  });
  return api;
}

int rccl_fail(cq_ctx* c, ncclResult_t r, const char* what) {
  return c->fail(CQ_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(r));
}

}  // namespace

int comm_rccl_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "CQ_RCCL_UNIQUE_ID_BYTES");
  if (!rccl().ok) return CQ_ERR_NO_DEVICE;
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return CQ_ERR_HIP;
  memcpy(id, &u, 128);
  return CQ_OK;
}

int comm_rccl_init(cq_ctx* c, uint32_t rank, uint32_t world, const uint8_t id[128]) {
  if (!rccl().ok) return c->fail(CQ_ERR_NO_DEVICE, "librccl.so could not be loaded");
  comm_rccl_destroy(c);
  // Staging of the small exchanges (partial sums, status words) and the side stream are set up HERE, collectively, so that
  // a proof allocates nothing on their account: an allocation failing on one rank in the middle of a proof would leave
  // its peers inside a collective.
  void* dummy;
  int rc;
  if ((rc = c->ensure_scratch(9, (size_t)1 << 20, &dummy)) != CQ_OK) return rc;
  if ((rc = c->ensure_pinned_comm((size_t)1 << 18, &dummy)) != CQ_OK) return rc;
  if ((rc = c->ensure_aux_stream()) != CQ_OK) return rc;
  if (!c->comm_event) CQ_HIP(c, hipEventCreateWithFlags(&c->comm_event, hipEventDisableTiming));
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t r = rccl().CommInitRank(&comm, (int)world, u, (int)rank);
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclCommInitRank");
  c->rccl_comm = comm;
  c->rccl_rank = rank;
  c->rccl_world = world;
  c->rccl_aborted = false;
  return CQ_OK;
}

void comm_rccl_destroy(cq_ctx* c) {
  if (!c->rccl_comm) return;
  hipStreamSynchronize(c->stream);
  if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
  rccl().CommDestroy((ncclComm_t)c->rccl_comm);
  c->rccl_comm = nullptr;
  c->rccl_world = 1;
  c->rccl_rank = 0;
}

// Gives up the communicator: its kernels in flight on this GPU stop waiting for peers, the peers see an asynchronous error
// or run into their own time-out (comm_rccl_wait) -- they fail instead of hanging.  Called when a sharded proof returns an
// error on this rank only (its peers are, or will be, inside a collective this rank never joins).
void comm_rccl_abort(cq_ctx* c) {
  if (!c->rccl_comm) return;
  if (rccl().CommAbort) rccl().CommAbort((ncclComm_t)c->rccl_comm);
  else rccl().CommDestroy((ncclComm_t)c->rccl_comm);
  c->rccl_comm = nullptr;
  c->rccl_aborted = true;
}

// Host wait for everything queued on `stream` so far, with the communicator's health in view: polls an event instead of
// blocking in hipStreamSynchronize, looks at ncclCommGetAsyncError, and gives up after CQ_COMM_TIMEOUT_S seconds (default
// 300; 0 = wait for ever) -- then the communicator is aborted and the call fails, so a rank whose peer died or left a
// proof early returns an error to its caller.
int comm_rccl_wait(cq_ctx* c, hipStream_t stream, const char* what) {
  if (!c->rccl_comm || !c->comm_event) {
    CQ_HIP(c, hipStreamSynchronize(stream));
    return CQ_OK;
  }
  static const double limit_s = []() {
    const char* e = getenv("CQ_COMM_TIMEOUT_S");
    return e ? atof(e) : 300.0;
  }();
  CQ_HIP(c, hipEventRecord(c->comm_event, stream));
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  for (;;) {
    const hipError_t q = hipEventQuery(c->comm_event);
    if (q == hipSuccess) return CQ_OK;
    if (q != hipErrorNotReady) return c->hip_fail(q, what);
    if ((++spins & 0x3ff) == 0) {  // every ~thousand polls: the communicator's own verdict, and the clock
      ncclResult_t async = ncclSuccess;
      if (rccl().CommGetAsyncError && rccl().CommGetAsyncError((ncclComm_t)c->rccl_comm, &async) == ncclSuccess && async != ncclSuccess &&
          async != ncclInProgress) {
        comm_rccl_abort(c);
        return c->fail(CQ_ERR_HIP, std::string(what) + ": RCCL reported an asynchronous error (a peer failed): " + rccl().GetErrorString(async));
      }
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (limit_s > 0 && el > limit_s) {
        comm_rccl_abort(c);
        return c->fail(CQ_ERR_HIP, std::string(what) + ": collective did not complete within CQ_COMM_TIMEOUT_S (a peer left or died); communicator aborted");
      }
      if (el > 0.002) std::this_thread::yield();
    }
  }
}

// host -> pinned -> device, ncclAllGather on the context's stream, device -> pinned -> host
static int rccl_allgather_host(cq_ctx* c, const void* send, void* recv, size_t bytes) {
  const size_t total = bytes * c->rccl_world;
  void *dev, *pin;
  int rc;
  if ((rc = c->ensure_scratch(9, bytes + total + 512, &dev)) != CQ_OK) return rc;
  if ((rc = c->ensure_pinned_comm(bytes + total, &pin)) != CQ_OK) return rc;
  char* dsend = (char*)dev;
  char* drecv = dsend + ((bytes + 255) & ~(size_t)255);
  memcpy(pin, send, bytes);
  CQ_HIP(c, hipMemcpyAsync(dsend, pin, bytes, hipMemcpyHostToDevice, c->stream));
  ncclResult_t r = rccl().AllGather(dsend, drecv, bytes, ncclUint8, (ncclComm_t)c->rccl_comm, c->stream);
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclAllGather");
  CQ_HIP(c, hipMemcpyAsync((char*)pin + bytes, drecv, total, hipMemcpyDeviceToHost, c->stream));
  if ((rc = comm_rccl_wait(c, c->stream, "ncclAllGather")) != CQ_OK) return rc;
  memcpy(recv, (char*)pin + bytes, total);
  return CQ_OK;
}

int rccl_bcast_parts(cq_ctx* c, const BcastPart* parts, size_t nparts, hipStream_t stream);

int shard_allgather_host(const cq_pk* pk, const void* send, void* recv, size_t bytes) {
  cq_ctx* c = pk->ctx;
  if (pk->allgather) {
    if (pk->allgather(pk->allgather_user, send, recv, bytes) != 0) return c->fail(CQ_ERR_INTERNAL, "allgather callback failed");
    return CQ_OK;
  }
  if (c->rccl_aborted) return c->fail(CQ_ERR_HIP, "sharding: the context's RCCL communicator was aborted after an earlier failure; build a new one");
  if (!c->rccl_comm || c->rccl_world != pk->shard_world || c->rccl_rank != pk->shard_rank)
    return c->fail(CQ_ERR_ARG, "sharding: no collective (no callback, no matching RCCL communicator on the context)");
  return rccl_allgather_host(c, send, recv, bytes);
}

bool shard_columns_enabled(const cq_pk* pk) {
  if (!pk->sharded() || !pk->shard_columns) return false;
  return pk->bcast != nullptr || (pk->ctx->rccl_comm && !pk->allgather);
}

int shard_bcast_parts(const cq_pk* pk, const BcastPart* parts, size_t nparts, hipStream_t stream) {
  cq_ctx* c = pk->ctx;
  if (!pk->sharded() || nparts == 0) return CQ_OK;
  if (pk->bcast) {
    // host transport (tests): one part at a time through a host buffer
    size_t maxb = 0;
    for (size_t i = 0; i < nparts; i++) maxb = parts[i].bytes > maxb ? parts[i].bytes : maxb;
    std::vector<uint8_t> host(maxb);
    for (size_t i = 0; i < nparts; i++) {
      if (!parts[i].bytes) continue;
      const bool mine = parts[i].root == pk->shard_rank;
      if (mine) {
        CQ_HIP(c, hipMemcpyAsync(host.data(), parts[i].ptr, parts[i].bytes, hipMemcpyDeviceToHost, stream));
        CQ_HIP(c, hipStreamSynchronize(stream));
      }
      if (pk->bcast(pk->bcast_user, host.data(), parts[i].bytes, parts[i].root) != 0)
        return c->fail(CQ_ERR_INTERNAL, "broadcast callback failed");
      if (!mine) {
        CQ_HIP(c, hipMemcpyAsync(parts[i].ptr, host.data(), parts[i].bytes, hipMemcpyHostToDevice, stream));
        CQ_HIP(c, hipStreamSynchronize(stream));  // `host` is reused by the next part
      }
    }
    return CQ_OK;
  }
  return rccl_bcast_parts(c, parts, nparts, stream);
}

bool shard_resident_enabled(const cq_pk* pk) {
  if (!pk->sharded() || !pk->shard_resident) return false;
  return pk->exchange != nullptr || (pk->ctx->rccl_comm && !pk->allgather);
}

int shard_sum_scalars(const cq_pk* pk, Fr* vals, size_t count) {
  if (!pk->sharded() || count == 0) return CQ_OK;
  std::vector<Fr> all((size_t)pk->shard_world * count);
  int rc = shard_allgather_host(pk, vals, all.data(), count * sizeof(Fr));
  if (rc != CQ_OK) return rc;
  for (size_t i = 0; i < count; i++) {
    Fr acc = Fr::zero();
    for (uint32_t r = 0; r < pk->shard_world; r++) acc = acc + all[(size_t)r * count + i];
    vals[i] = acc;
  }
  return CQ_OK;
}

int shard_exchange(const cq_pk* pk, const Xfer* x, size_t count, hipStream_t stream) {
  cq_ctx* c = pk->ctx;
  if (!pk->sharded() || count == 0) return CQ_OK;
  const uint32_t me = pk->shard_rank;
  for (size_t i = 0; i < count; i++)
    if (x[i].from == me && x[i].to == me && x[i].bytes && x[i].src != x[i].dst)
      CQ_HIP(c, hipMemcpyAsync(x[i].dst, x[i].src, x[i].bytes, hipMemcpyDeviceToDevice, stream));
  if (pk->exchange) {
    // host transport (gloo in the tests): stage every outgoing range, hand the rank's part of the list over, upload what came
    size_t total = 0;
    for (size_t i = 0; i < count; i++)
      if (x[i].from != x[i].to && (x[i].from == me || x[i].to == me)) total += (x[i].bytes + 63) & ~(size_t)63;
    std::vector<uint8_t> host(total + 64);
    std::vector<cq_xfer> mine;
    std::vector<size_t> idx;
    size_t off = 0;
    for (size_t i = 0; i < count; i++) {
      if (x[i].from == x[i].to || (x[i].from != me && x[i].to != me) || !x[i].bytes) continue;
      const bool send = x[i].from == me;
      if (send) CQ_HIP(c, hipMemcpyAsync(host.data() + off, x[i].src, x[i].bytes, hipMemcpyDeviceToHost, stream));
      mine.push_back({send ? x[i].to : x[i].from, send ? 1u : 0u, host.data() + off, x[i].bytes});
      idx.push_back(i);
      off += (x[i].bytes + 63) & ~(size_t)63;
    }
    CQ_HIP(c, hipStreamSynchronize(stream));
    if (!mine.empty() && pk->exchange(pk->exchange_user, mine.data(), mine.size()) != 0) return c->fail(CQ_ERR_INTERNAL, "exchange callback failed");
    for (size_t j = 0; j < mine.size(); j++)
      if (!mine[j].send) CQ_HIP(c, hipMemcpyAsync(x[idx[j]].dst, mine[j].buf, mine[j].bytes, hipMemcpyHostToDevice, stream));
    CQ_HIP(c, hipStreamSynchronize(stream));  // `host` goes out of scope
    return CQ_OK;
  }
  if (c->rccl_aborted) return c->fail(CQ_ERR_HIP, "sharding: the context's RCCL communicator was aborted after an earlier failure; build a new one");
  if (!c->rccl_comm) return c->fail(CQ_ERR_ARG, "resident sharding: no transport");
  if (!rccl().Send || !rccl().Recv) return c->fail(CQ_ERR_NO_DEVICE, "resident sharding: this librccl has no ncclSend / ncclRecv");
  ncclResult_t r = rccl().GroupStart();
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclGroupStart");
  for (size_t i = 0; i < count && r == ncclSuccess; i++) {
    if (x[i].from == x[i].to || !x[i].bytes) continue;
    if (x[i].from == me) r = rccl().Send(x[i].src, x[i].bytes, ncclUint8, (int)x[i].to, (ncclComm_t)c->rccl_comm, stream);
    else if (x[i].to == me) r = rccl().Recv(x[i].dst, x[i].bytes, ncclUint8, (int)x[i].from, (ncclComm_t)c->rccl_comm, stream);
  }
  const ncclResult_t e = rccl().GroupEnd();
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclSend / ncclRecv");
  if (e != ncclSuccess) return rccl_fail(c, e, "ncclGroupEnd");
  return CQ_OK;
}

// one grouped launch: every rank is the root of its own parts (what an all-gather with unequal counts would be)
int rccl_bcast_parts(cq_ctx* c, const BcastPart* parts, size_t nparts, hipStream_t stream) {
  if (!c->rccl_comm) return c->fail(CQ_ERR_ARG, "column sharding: no transport");
  ncclResult_t r = rccl().GroupStart();
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclGroupStart");
  for (size_t i = 0; i < nparts && r == ncclSuccess; i++)
    if (parts[i].bytes)
      r = rccl().Broadcast(parts[i].ptr, parts[i].ptr, parts[i].bytes, ncclUint8, (int)parts[i].root, (ncclComm_t)c->rccl_comm, stream);
  const ncclResult_t e = rccl().GroupEnd();
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclBroadcast");
  if (e != ncclSuccess) return rccl_fail(c, e, "ncclGroupEnd");
  return CQ_OK;
}

// Both collectives on the context's communicator with patterns every rank can check, issued the way the sharded prover
// issues them (prover.hip): an all-gather on the main stream; then grouped broadcasts on the SIDE stream, ordered behind
// an event of the main stream (AuxFork) and followed by more side-stream work; an all-gather on the main stream while
// those are in flight -- two streams, one communicator, the same issue order on every rank --; the main stream joins
// the side stream last.  What cq_ctx_comm_selftest runs.
int comm_rccl_selftest(cq_ctx* c) {
  if (!c->rccl_comm) return c->fail(CQ_ERR_ARG, "comm selftest: no RCCL communicator on the context");
  const uint32_t W = c->rccl_world, R = c->rccl_rank;
  const size_t bytes = 1000;  // deliberately not a multiple of 16
  std::vector<uint8_t> send(bytes), recv(bytes * W);
  auto fill = [&](uint32_t salt) {
    for (size_t i = 0; i < bytes; i++) send[i] = (uint8_t)(i * 7 + R * 31 + salt);
  };
  auto check = [&](uint32_t salt) {
    for (uint32_t r = 0; r < W; r++)
      for (size_t i = 0; i < bytes; i++)
        if (recv[r * bytes + i] != (uint8_t)(i * 7 + r * 31 + salt)) return false;
    return true;
  };
  int rc;
  fill(1);
  if ((rc = rccl_allgather_host(c, send.data(), recv.data(), bytes)) != CQ_OK) return rc;
  if (!check(1)) return c->fail(CQ_ERR_INTERNAL, "comm selftest: all-gather payload mismatch");
  // broadcasts: part r (a different length per rank) owned by rank r, in one grouped launch on the side stream
  std::vector<size_t> len(W), off(W);
  size_t total = 0;
  for (uint32_t r = 0; r < W; r++) {
    len[r] = ((size_t)1 << 20) + 512 * r + 40;
    off[r] = total;
    total += (len[r] + 255) & ~(size_t)255;
  }
  void* dev;
  if ((rc = c->ensure_scratch(8, total + 4096, &dev)) != CQ_OK) return rc;  // (slot 9 stages the all-gathers)
  std::vector<uint8_t> host(total, 0);
  for (size_t i = 0; i < len[R]; i++) host[off[R] + i] = (uint8_t)(i * 13 + R * 17 + 3);
  CQ_HIP(c, hipMemcpyAsync(dev, host.data(), total, hipMemcpyHostToDevice, c->stream));
  if ((rc = c->ensure_aux_stream()) != CQ_OK) return rc;
  CQ_HIP(c, hipEventRecord(c->msm_tail_event, c->stream));
  CQ_HIP(c, hipStreamWaitEvent(c->aux_stream, c->msm_tail_event, 0));
  std::vector<BcastPart> parts;
  for (uint32_t r = 0; r < W; r++) parts.push_back({(char*)dev + off[r], len[r], r});
  if ((rc = rccl_bcast_parts(c, parts.data(), parts.size(), c->aux_stream)) != CQ_OK) return rc;
  std::vector<uint8_t> back(total, 0);
  CQ_HIP(c, hipMemcpyAsync(back.data(), dev, total, hipMemcpyDeviceToHost, c->aux_stream));  // "more side-stream work"
  CQ_HIP(c, hipEventRecord(c->aux_done, c->aux_stream));
  fill(2);
  if ((rc = rccl_allgather_host(c, send.data(), recv.data(), bytes)) != CQ_OK) return rc;  // main stream, broadcasts in flight
  if (!check(2)) return c->fail(CQ_ERR_INTERNAL, "comm selftest: all-gather payload mismatch (second)");
  CQ_HIP(c, hipStreamWaitEvent(c->stream, c->aux_done, 0));
  if ((rc = comm_rccl_wait(c, c->stream, "comm selftest")) != CQ_OK) return rc;
  for (uint32_t r = 0; r < W; r++)
    for (size_t i = 0; i < len[r]; i++)
      if (back[off[r] + i] != (uint8_t)(i * 13 + r * 17 + 3)) return c->fail(CQ_ERR_INTERNAL, "comm selftest: broadcast payload mismatch");
  return CQ_OK;
}

}  // namespace cq
