// Internal interface of the NTT engine (see ntt.hip for the algorithm).
#pragma once
#include <hip/hip_runtime.h>
#include "field.hpp"
#include "field29.hpp"

struct cq_ctx;

namespace cq {

#ifndef CQ_NTT_THREADS
#define CQ_NTT_THREADS 256
#endif
constexpr uint32_t NTT_THREADS = CQ_NTT_THREADS;
#ifndef CQ_NTT_MAX_DEG
#define CQ_NTT_MAX_DEG 6
#endif
#ifndef CQ_NTT_TILE_ELEMS
#define CQ_NTT_TILE_ELEMS 1024
#endif
constexpr uint32_t NTT_MAX_DEG = CQ_NTT_MAX_DEG;        // bits resolved per pass (2^deg-point FFT in LDS)
constexpr uint32_t NTT_TILE_ELEMS = CQ_NTT_TILE_ELEMS;  // elements per workgroup tile (x32 B of LDS)

enum : uint32_t { NTT_IN_COSET = 1, NTT_OUT_MUL = 2, NTT_OUT_COSET = 4, NTT_CRITICAL = 8 };  // CRITICAL: raised wave priority (field.hpp)

struct NttPassArgs {
  const Fr* in;
  Fr* out;
  size_t in_stride, out_stride;  // batch strides, in elements
  uint32_t log_n, lgp, deg, log_t;
  uint32_t next_deg;  // bits the NEXT pass resolves (0: this is the last pass); its twiddles are applied at this pass's store
  // twiddle tables hold the kernels' own form: canonical values in Montgomery form with R' = 2^261, packed into 8 x u32
  const Fr* tw_lo;  // w^j, j < 2^tw_l
  const Fr* tw_hi;  // w^(j << tw_l)
  const Fr* tw_full;  // w^j, j < n (nullptr: compose tw_lo * tw_hi)
  uint32_t tw_l;
  const Fr* pq;  // (w^(n >> pq_log))^j, j < 2^(pq_log-1)
  uint32_t pq_shift;
  uint32_t flags;
  uint32_t in_len;   // first pass: elements >= in_len read as zero (zero padding)
  uint32_t out_len;  // last pass: elements >= out_len are not written (truncation)
  Fr in_coset[2];    // first pass: element g multiplied by in_coset[g%3 - 1] when g%3 != 0
  Fr out_mul[3];     // last pass: element g multiplied by out_mul[g%3] (or out_mul[0] without OUT_COSET)
};

// Device-resident twiddles for one (log_n, omega).
struct NttTables {
  uint32_t log_n = 0;
  Fr omega;
  Fr* tw_lo = nullptr;
  Fr* tw_hi = nullptr;
  Fr* tw_full = nullptr;  // every power of omega (n x 32 B): the inter-pass twiddle is then one load, not a product
  Fr* pq = nullptr;
  uint32_t tw_l = 0, pq_log = 0;
  NttTables() = default;
  NttTables(const NttTables&) = delete;
  NttTables& operator=(const NttTables&) = delete;
  ~NttTables();
  int build(uint32_t log_n, const Fr& omega, hipStream_t stream);
};

struct NttIo {
  uint32_t batch = 1;
  size_t in_stride = 0, out_stride = 0;
  uint32_t in_len = 0, out_len = 0;
  bool in_coset = false;
  Fr in_coset_mul[2];
  bool out_mul = false, out_coset = false;
  Fr out_mul_v[3];
  cq_ctx* prof = nullptr;  // when set and profiling is on, passes are bracketed with HIP events
  bool critical = false;   // the transform is on the path a proof waits for (main stream): NTT_CRITICAL for its passes
};

// out = NTT(in) over `tb`.  `scratch` holds 2 * batch * 2^log_n elements; `in` may be shorter
// (io.in_len, zero-padded), `out` may be shorter (io.out_len, truncated) and may alias `in`.
int ntt_run(const NttTables& tb, const Fr* in, Fr* out, Fr* scratch, const NttIo& io, hipStream_t stream);

}  // namespace cq
