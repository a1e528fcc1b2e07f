// Internal interface of the MSM engine (see msm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "curve.hpp"

struct cq_ctx;

namespace cq {

constexpr uint32_t MSM_ACC_THREADS = 128;

#ifndef CQ_MSM_S1
#define CQ_MSM_S1 24
#endif
constexpr uint32_t MSM_S1 = CQ_MSM_S1;  // max point indices summed by one lane (level 1)
#ifndef CQ_MSM_S1_BIG
#define CQ_MSM_S1_BIG 32
#endif
#ifndef CQ_MSM_S1_BIG_LOAD
#define CQ_MSM_S1_BIG_LOAD 512
#endif
constexpr uint32_t MSM_S1_BIG = CQ_MSM_S1_BIG;  // ... at least this when a bucket expects more than MSM_S1_BIG_LOAD entries
constexpr uint32_t MSM_S1_BIG_LOAD = CQ_MSM_S1_BIG_LOAD;
#ifndef CQ_MSM_PARTIALS_TARGET
#define CQ_MSM_PARTIALS_TARGET 48
#endif
constexpr uint32_t MSM_PARTIALS_TARGET = CQ_MSM_PARTIALS_TARGET;  // partial sums per bucket aimed at under heavy load (< MSM_SHORT)
static_assert(MSM_S1_BIG >= MSM_S1, "the workspace is sized for MSM_S1");
// Small launches (fewer than MSM_SMALL_LANES sub-lists of MSM_S1 entries): shorter sub-lists, down to MSM_S1_MIN, so that
// the accumulate kernel -- a chain of dependent additions per lane, ~7 us each -- has a lane per few entries instead of
// 17-24 of them on a fraction of the SIMDs (k = 14: one 16 384-term MSM was 140 us of pure latency).  A function of the
// launch's entry BOUND (batch x windows x n), so that the workspace layout and the launch agree.
#ifndef CQ_MSM_SMALL_LANES
#define CQ_MSM_SMALL_LANES 65536
#endif
constexpr uint32_t MSM_S1_MIN = 4;
constexpr uint64_t MSM_SMALL_LANES = CQ_MSM_SMALL_LANES;
__host__ __device__ inline uint32_t msm_small_launch_s1(uint64_t entry_bound) {
  if (entry_bound >= (uint64_t)MSM_S1 * MSM_SMALL_LANES) return MSM_S1;
  const uint64_t s = entry_bound / MSM_SMALL_LANES;
  return (uint32_t)(s < MSM_S1_MIN ? MSM_S1_MIN : s);
}
// sub-lists of a launch whose entry COUNT (known on the device only) is small: at most 65536 x (s1 + 1) / s1 <= 1.25 x 65536
// lanes' worth, plus an MSM that reads another one's lists (counted once more); 3 x 65536 bounds both
constexpr uint64_t MSM_SMALL_PARTIALS = 3 * MSM_SMALL_LANES;
constexpr uint32_t MSM_S2 = 64;         // max partial sums summed by one wave (levels >= 2): one load per lane, six shuffle levels
constexpr uint32_t MSM_SHORT = 64;       // level >= 2 lists up to this long are summed by a lane group when the launch is throughput-bound
constexpr uint32_t MSM_SHORT_MIN = 16;   // ... and lists up to this long always; the ones in between go to a wave each when they are few
#ifndef CQ_MSM_WAVE_BUDGET
#define CQ_MSM_WAVE_BUDGET 4096
#endif
constexpr uint32_t MSM_WAVE_BUDGET = CQ_MSM_WAVE_BUDGET;  // "few": at most this many wave slots in the level (4 per SIMD)
constexpr uint32_t MSM_COMBINE_WAVE_BLOCKS = 2048;  // blocks of the combine levels' wave half (8 192 waves striding over the slots)
constexpr uint32_t MSM_MAX_BATCH = 32;  // MSMs per launch
constexpr uint32_t MSM_SET_POINTS = 15; // points a bucket set leaves for the host: 7 + 7 bit-plane sums and a total (msm_set_value)

struct MsmPtrs {
  const void* p[MSM_MAX_BATCH];
};
struct MsmStrides {
  uint64_t s[MSM_MAX_BATCH];
};

// Workspace carve-up for `batch` MSMs of n terms each with c-bit signed windows.
struct MsmLayout {
  // M: buckets per bucket SET (the reduction's rows x cols matrix), Wb: sets per MSM, B = Wb * M: buckets per MSM.
  // Table mode has one set per MSM up to c = 15 (M = 2^(c-1)) and 2^(c-15) sets of 2^14 buckets for wider windows.
  uint32_t n, c, batch, W, Wb, M, B, Bt, levels, nseq, nblk, rows, cols, npart, nfinal, shift1;
  bool pre, part_sort, mid;
  uint64_t tmax[8];
  size_t off_ptrs, off_ranks, off_poff, off_poff2, off_counts, off_cursor, off_psize, off_psize2, off_buckets, off_ticket, zero_end, off_blocksums,
      off_off, off_tk, off_sorted, off_part[2], off_pairs, total;
  MsmLayout(uint32_t n_, uint32_t c_, uint32_t batch_, bool pre_ = false);
};

uint32_t msm_window_bits(uint32_t n);
// Enqueues the whole pipeline on ctx->stream.
// `scalars` / `bases`: HOST arrays of `batch` device pointers; window_sums_dev receives MSM_SET_POINTS points per
// bucket set -- bit-plane sums the host folds into the set's value (msm_set_value; cols = MsmLayout::cols).
// `pre`: every bases[j] points to a precomputed table [W][table_stride] with table[w][i] = 2^(c*w)*base[i]
// (msm_precompute_tables); then there is ONE bucket set per MSM.
// lens[j] <= n: per-MSM lengths (one launch may mix lengths; n is the maximum)
int msm_run(cq_ctx* ctx, const Fr* const* scalars, const G1Affine* const* bases, const size_t* lens, uint32_t n, uint32_t c,
            uint32_t batch, bool pre, const size_t* table_strides, void* workspace, G1Jac* window_sums_dev);
int msm_precompute_tables(cq_ctx* ctx, const G1Affine* bases, uint32_t n, uint32_t c, G1Affine* table);
// Host: value of one bucket set from its MSM_SET_POINTS points; sum_w 2^(c*w) * (value of set w) over W consecutive sets.
G1Jac msm_set_value(const G1Jac* planes, uint32_t cols);
G1Jac msm_fold_windows(const G1Jac* pairs, uint32_t W, uint32_t c, uint32_t cols);
// Host, table mode with wide windows: the MSM's value from its Wb consecutive sets of M buckets each (set s holds the
// buckets s * M + 1 .. (s + 1) * M):  sum_s ( value(set s) + s * M * total(set s) )
G1Jac msm_fold_sets(const G1Jac* pairs, uint32_t Wb, uint32_t M, uint32_t cols);

}  // namespace cq

// `count` MSMs of equal length, each with its own device base array; Jacobian results on the host
// (count x 12 limbs); one stream synchronisation per launch batch.
int cq_msm_multi(cq_ctx* c, const cq::Fr* const* scalars, const cq::G1Affine* const* bases, size_t len, size_t count,
                 uint64_t* out_jac);
// per-MSM lengths; MSMs over registered (precomputed) bases of any length share launches
int cq_msm_multi_v(cq_ctx* c, const cq::Fr* const* scalars, const cq::G1Affine* const* bases, const size_t* lens, size_t count,
                   uint64_t* out_jac);
// asynchronous form: begin() enqueues every launch and the device->host copy of the results,
// end() waits for the stream and folds them
struct MsmPending {
  struct Launch {
    size_t first = 0, slot = 0;
    uint32_t batch = 0, c = 0, nmax = 0, W = 0, Wb = 0, M = 0, cols = 0;
    bool pre = false, empty = false;
  };
  std::vector<Launch> launches;
  void* host = nullptr;
  size_t slots = 0, count = 0;
};
int msm_multi_begin(cq_ctx* c, const cq::Fr* const* scalars, const cq::G1Affine* const* bases, const size_t* lens, size_t count,
                    MsmPending& pend);
int msm_multi_end(cq_ctx* c, MsmPending& pend, uint64_t* out_jac);
// Window width of the precomputed tables: one width per context, so that MSMs over different base arrays can share a
// launch.  15 (17 windows, 2^14 buckets) up to 2^19 terms; wider from 2^20 on, where two or four fewer additions per
// scalar outweigh the larger bucket reduction (msm_table_window_bits; cq_msm_set_table_window overrides).
constexpr uint32_t MSM_TABLE_C = 15, MSM_TABLE_C_MIN = 8, MSM_TABLE_C_MAX = 20;
uint32_t msm_table_window_bits(size_t n);

// Registry of the window tables (capi_msm.hip).  msm_register_tables: makes sure MSMs over [bases, bases + n) find a table --
// of width `want_c` when given, else of the width the array's own length calls for (or the caller's override).  *held
// (optional) tells whether the caller now holds a reference of an entry for exactly (bases, n, width) -- created, or found
// and shared -- that it must give back with msm_release_table; false when the range is served by a larger array's table.
// msm_unregister_tables drops every table of an array that is going away (its owner's call), whoever else held them.
int msm_register_tables(cq_ctx* c, const cq::G1Affine* bases, size_t n, uint32_t want_c = 0, bool* held = nullptr);
void msm_release_table(cq_ctx* c, const void* bases, size_t n, uint32_t c_bits);
void msm_unregister_tables(cq_ctx* c, const void* bases);
