// Collectives of the multi-GPU proving modes (one process per GPU).  Two transports behind one interface:
//   * RCCL over xGMI, called directly from here on device buffers and on the context's stream -- the library is loaded
//     at run time (dlopen: the copy the process already holds, e.g. torch's, else the ROCm one), so libcq_halo2.so
//     has no link-time dependency on it and single-GPU users never load it;
//   * the caller's own collective through host-buffer callbacks (cq_allgather_fn / cq_bcast_fn): gloo in the CPU-side
//     and two-ranks-on-one-GPU tests, or whatever an embedding application uses.
// What travels: per-round MSM partial sums (count x 96 B per rank, all-gather) and, with column sharding, whole
// transformed columns (broadcast from their owner).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "field.hpp"

struct cq_ctx;
struct cq_pk;

namespace cq {

int comm_rccl_unique_id(uint8_t id[128]);
int comm_rccl_init(cq_ctx* c, uint32_t rank, uint32_t world, const uint8_t id[128]);
void comm_rccl_destroy(cq_ctx* c);
void comm_rccl_abort(cq_ctx* c);  // this rank gives up the communicator: its peers fail (async error / time-out) instead of hanging
int comm_rccl_wait(cq_ctx* c, hipStream_t stream, const char* what);  // host wait with time-out and async-error polling
int comm_rccl_selftest(cq_ctx* c);

// every rank contributes `bytes` of HOST memory; recv (world x bytes, rank order) is host memory too.  Synchronises
// the context's stream.
int shard_allgather_host(const cq_pk* pk, const void* send, void* recv, size_t bytes);
// Column exchange: `nparts` device ranges (ptr, bytes), part i owned by rank roots[i]; after the call every rank holds
// every part.  Enqueued on `stream` (RCCL: one grouped launch of broadcasts) or staged through the host callback.
struct BcastPart {
  void* ptr;
  size_t bytes;
  uint32_t root;
};
int shard_bcast_parts(const cq_pk* pk, const BcastPart* parts, size_t nparts, hipStream_t stream);
// Point-to-point transfers of device ranges: every rank passes the SAME global list in the same order and takes part in
// the entries that name it -- sends `src` where it is `from`, receives into `dst` where it is `to` (from == to: a device
// copy).  Enqueued on `stream` (RCCL: one group of ncclSend / ncclRecv) or staged through the host hook (synchronises).
struct Xfer {
  const void* src;
  void* dst;
  size_t bytes;
  uint32_t from, to;
};
int shard_exchange(const cq_pk* pk, const Xfer* x, size_t count, hipStream_t stream);
// sum over the ranks of `count` field elements each contributes (host): out[i] = sum_r vals_r[i] -- one all-gather
int shard_sum_scalars(const cq_pk* pk, cq::Fr* vals, size_t count);
// resident column sharding is on and has a transport
bool shard_resident_enabled(const cq_pk* pk);
// true when whole-column work may be split between the ranks (a transport for big payloads exists and it is enabled)
bool shard_columns_enabled(const cq_pk* pk);

}  // namespace cq
