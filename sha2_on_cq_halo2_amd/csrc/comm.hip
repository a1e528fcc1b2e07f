// See comm.hpp.  RCCL is reached through dlopen/dlsym only.
#include "comm.hpp"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
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
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t r = rccl().CommInitRank(&comm, (int)world, u, (int)rank);
  if (r != ncclSuccess) return rccl_fail(c, r, "ncclCommInitRank");
  c->rccl_comm = comm;
  c->rccl_rank = rank;
  c->rccl_world = world;
  return CQ_OK;
}

void comm_rccl_destroy(cq_ctx* c) {
  if (!c->rccl_comm) return;
  hipStreamSynchronize(c->stream);
  rccl().CommDestroy((ncclComm_t)c->rccl_comm);
  c->rccl_comm = nullptr;
  c->rccl_world = 1;
  c->rccl_rank = 0;
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
  CQ_HIP(c, hipStreamSynchronize(c->stream));
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

// Both collectives on the context's communicator with patterns every rank can check: what cq_ctx_comm_selftest runs.
int comm_rccl_selftest(cq_ctx* c) {
  if (!c->rccl_comm) return c->fail(CQ_ERR_ARG, "comm selftest: no RCCL communicator on the context");
  const uint32_t W = c->rccl_world, R = c->rccl_rank;
  const size_t bytes = 1000;  // deliberately not a multiple of 16
  std::vector<uint8_t> send(bytes), recv(bytes * W);
  for (size_t i = 0; i < bytes; i++) send[i] = (uint8_t)(i * 7 + R * 31 + 1);
  int rc = rccl_allgather_host(c, send.data(), recv.data(), bytes);
  if (rc != CQ_OK) return rc;
  for (uint32_t r = 0; r < W; r++)
    for (size_t i = 0; i < bytes; i++)
      if (recv[r * bytes + i] != (uint8_t)(i * 7 + r * 31 + 1)) return c->fail(CQ_ERR_INTERNAL, "comm selftest: all-gather payload mismatch");
  // broadcasts: part r (a different length per rank) owned by rank r, in one grouped launch
  std::vector<size_t> len(W), off(W);
  size_t total = 0;
  for (uint32_t r = 0; r < W; r++) {
    len[r] = 4096 + 512 * r;
    off[r] = total;
    total += len[r];
  }
  void* dev;
  if ((rc = c->ensure_scratch(9, total + 4096, &dev)) != CQ_OK) return rc;
  std::vector<uint8_t> host(total, 0);
  for (size_t i = 0; i < len[R]; i++) host[off[R] + i] = (uint8_t)(i * 13 + R * 17 + 3);
  CQ_HIP(c, hipMemcpyAsync(dev, host.data(), total, hipMemcpyHostToDevice, c->stream));
  std::vector<BcastPart> parts;
  for (uint32_t r = 0; r < W; r++) parts.push_back({(char*)dev + off[r], len[r], r});
  if ((rc = rccl_bcast_parts(c, parts.data(), parts.size(), c->stream)) != CQ_OK) return rc;
  CQ_HIP(c, hipMemcpyAsync(host.data(), dev, total, hipMemcpyDeviceToHost, c->stream));
  CQ_HIP(c, hipStreamSynchronize(c->stream));
  for (uint32_t r = 0; r < W; r++)
    for (size_t i = 0; i < len[r]; i++)
      if (host[off[r] + i] != (uint8_t)(i * 13 + r * 17 + 3)) return c->fail(CQ_ERR_INTERNAL, "comm selftest: broadcast payload mismatch");
  return CQ_OK;
}

}  // namespace cq
