// Pippenger multi-scalar multiplication on BN254 G1 for gfx950 -- replaces `best_multiexp`
// (halo2_proofs/src/arithmetic.rs:132-159, serial core :13-101) and therefore
// `ParamsKZG::commit` / `commit_lagrange` (poly/kzg/commitment.rs:496-504,539-543).
//
// The reference chunks the input per CPU thread and runs unsigned c=ceil(ln n)-bit windows with
// `None/Affine/Projective` buckets.  None of that shape survives here; only the result (a group
// element, canonical once normalised) has to match.  A launch handles a BATCH of independent
// MSMs of equal length (e.g. all advice columns of a phase, plonk/prover.rs:356-360); every
// (msm, window, bucket) triple is one entry of a flat bucket array, so the batch only adds
// parallelism.  GPU pipeline, all on one stream:
//   1. digits    : scalar -> canonical (one Montgomery reduction) -> SIGNED c-bit digits (halves the bucket
//                  count).  An entry = (point, window, sign) of a non-zero digit; its bucket = |digit| - 1.
//   2. sort      : counting sort of the entries by bucket.  Table mode: two passes through LDS (partition = bucket
//                  >> 7, then bucket), every global write a run of consecutive entries -- see "Precomputed-table
//                  mode" below.  Plain mode: one device atomic per entry (issued back to back) whose return value
//                  is the entry's rank, then a scatter to list start + rank.
//   3. plan      : exclusive scans of the histogram (reduce / spine / apply): where each bucket's index list
//                  starts, and how many bounded sub-lists it is cut into per level.
//   4. accumulate: level 1 -- one lane per sub-list of <= MSM_S1 entries gathers affine points (64 B each, from
//                  the per-window tables or the caller's array) into an XYZZ accumulator (8M+2S per mixed add,
//                  no inversion) on the lazy 9x29-bit field (curve29.hpp); the next point's gather is in flight
//                  while the current addition runs.  Levels >= 2: one to four LANES per short list of partial sums
//                  (indexed by bucket), one WAVE per long one (strided lane sums + 6-step shuffle tree).  A bucket is written as soon
//                  as one lane/wave owns all of it.  Bounding the per-lane work keeps the kernel balanced for
//                  skewed digit distributions (0/1 selector columns, SHA limb columns) where a lane-per-bucket
//                  kernel would serialise on one huge bucket.  256-bit modular integer work: VALU-bound, MFMA
//                  does not apply.
//   5. reduce    : sum_b b*B_b per bucket set, shaped for depth: buckets as a rows x cols matrix, 16 lanes per
//                  row / column sum, then two weighted wave sums (the reference does this serially,
//                  arithmetic.rs:95-99).
//   6. host      : table mode returns one Jacobian point per MSM; plain mode W window sums per MSM, folded by a
//                  Horner over windows (c doublings each) on the host.
#include <algorithm>
#include <cstdlib>
#include "msm.hpp"
#include "curve29.hpp"
#include "ctx.hpp"

namespace cq {

// ---- pointer tables (per-MSM scalar / base arrays), written from a by-value kernel argument ------
// `ln`: the length the SORT sees (0 for an MSM whose entry lists are another MSM's, see msm_run); `src`: the MSM whose
// lists an MSM accumulates from (itself, unless it shares its scalars with an earlier one)
// The launch's first kernel: the pointer / length tables the later kernels read, and the counts and sort cursors cleared
// (16-byte stores, grid-stride) -- one launch instead of a 5 us kernel and a 5 us memset at the head of every front.
__global__ __launch_bounds__(256) void msm_set_ptrs_kernel(MsmPtrs sc, MsmPtrs bs, MsmStrides st, MsmStrides ln, MsmStrides src, const void** dst,
                                                          uint32_t batch, uint4* __restrict__ zero, size_t zero_quads) {
  CQ_CRITICAL_WAVES();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_quads; i += (size_t)gridDim.x * blockDim.x)
    zero[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x) return;
  const uint32_t t = threadIdx.x;
  if (t < batch) {
    dst[t] = sc.p[t];
    dst[batch + t] = bs.p[t];
    ((uint64_t*)(dst + 2 * batch))[t] = st.s[t];
    ((uint64_t*)(dst + 3 * batch))[t] = ln.s[t];
    ((uint64_t*)(dst + 4 * batch))[t] = src.s[t];
  }
}

// ---- 1. digits + histogram -------------------------------------------------------------------
// signed c-bit digit of window w of the canonical scalar v, with the running carry
static __device__ __forceinline__ uint32_t signed_digit(const U256& v, uint32_t w, uint32_t c, uint32_t& carry, uint32_t& neg) {
  const uint32_t M = 1u << (c - 1);
  const uint32_t bitpos = w * c;
  const uint32_t word = bitpos >> 5, sh = bitpos & 31;
  uint32_t raw = 0;
  if (word < 8) {
    uint64_t two = v.l[word];
    if (word + 1 < 8) two |= (uint64_t)v.l[word + 1] << 32;
    raw = (uint32_t)(two >> sh) & ((1u << c) - 1);
  }
  uint32_t d = raw + carry;
  neg = 0;
  carry = 0;
  if (d > M) {
    d = (1u << c) - d;
    neg = 1;
    carry = 1;
  }
  return d;
}

// One lane per scalar: every window's digit bumps the histogram; the value the atomic returns is the
// entry's rank inside its bucket, kept for the scatter pass (so the sort needs one atomic per entry).
//
// The kernel is bound by atomic round trips, so a wave never waits for one before issuing the next: windows are
// handled DIGIT_GROUP at a time -- all their atomics go out back to back (one instruction per window), and only
// then are the returned values turned into ranks.  Within a window the lanes that share the key of the first
// two pending lanes (hot keys: 0/1 columns, sparse top windows) are folded into their leader's atomic, which
// adds the group size; `meta` remembers leader lane and position inside the group.
constexpr uint32_t DIGIT_GROUP = 6;
__global__ __launch_bounds__(256) void msm_digits_kernel(const Fr* const* __restrict__ scalars, const uint64_t* __restrict__ lens,
                                                         uint32_t nmax, uint32_t c, uint32_t nwin, uint32_t pre,
                                                         uint32_t* __restrict__ ranks, uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  const uint32_t lane = threadIdx.x & 63;
  const bool live = i < (uint32_t)lens[m];
  U256 v;
  if (live) v = scalars[m][i].to_canonical();
  else for (int k = 0; k < 8; k++) v.l[k] = 0;
  const uint32_t M = 1u << (c - 1);
  uint32_t carry = 0, neg;
  for (uint32_t w0 = 0; w0 < nwin; w0 += DIGIT_GROUP) {
    uint32_t ret[DIGIT_GROUP], meta[DIGIT_GROUP];  // meta: 0xffffffff = inactive, else leader lane << 8 | offset, leader 64 = none
#pragma unroll
    for (uint32_t q = 0; q < DIGIT_GROUP; q++) {
      const uint32_t w = w0 + q;
      meta[q] = 0xffffffffu;
      ret[q] = 0;
      if (w >= nwin) continue;  // wave-uniform
      const uint32_t d = signed_digit(v, w, c, carry, neg);
      const size_t mw = (size_t)m * nwin + w;
      // precomputed-table mode folds every window into one bucket set per MSM
      const bool act = live && d != 0;
      const uint32_t key = (uint32_t)((pre ? (size_t)m : mw) * M) + (d - 1);
      bool pending = act;
      uint32_t add = 1, lead = 64, below = 0;
      bool issue = act;
#pragma unroll
      for (int round = 0; round < 2; round++) {
        const unsigned long long pend = __ballot(pending);
        if (pend) {  // wave-uniform
          const int leader = __ffsll((long long)pend) - 1;
          const uint32_t lkey = __shfl(key, leader, 64);
          const bool mine = pending && key == lkey;
          const unsigned long long grp = __ballot(mine);
          if (mine) {
            lead = (uint32_t)leader;
            below = __popcll(grp & ((1ull << lane) - 1ull));
            add = __popcll(grp);
            issue = (int)lane == leader;
            pending = false;
          }
        }
      }
      if (issue) ret[q] = atomicAdd(&counts[key], (lead == lane || lead == 64) ? add : 1u);
      if (act) meta[q] = (lead << 8) | below;
    }
#pragma unroll
    for (uint32_t q = 0; q < DIGIT_GROUP; q++) {
      const uint32_t w = w0 + q;
      if (w >= nwin) continue;
      const uint32_t lead = (meta[q] >> 8) & 0xff;
      const uint32_t base = __shfl(ret[q], lead & 63, 64);  // every lane takes part in the shuffle
      if (meta[q] != 0xffffffffu) {
        const uint32_t rank = lead < 64 ? base + (meta[q] & 0xff) : ret[q];
        ranks[((size_t)m * nwin + w) * nmax + i] = rank;
      }
    }
  }
}

// Precomputed-table mode (one bucket set of M = 2^14 buckets per MSM): a two-pass (MSD) counting sort.
//
// A single-pass sort drops every 4-byte entry at a random place of its MSM's list array; the L2 evicts those as
// 32-byte partial writes (measured: 32 B of write traffic per entry, 1.3 TB/s, twice that only while a whole launch
// fits the 256 MB memory-side cache), and the per-entry rank array costs another 8 B.  Here every pass writes runs:
//   a. part_hist    : per chunk of 2048 scalars, LDS histogram of the entries' PARTITION (bucket >> 7: 128
//                     partitions of 128 buckets per MSM); one device atomic per (chunk, partition).
//   b. part_scan    : exclusive scan of the partition sizes (<= 32 x 128 values, one block).
//   c. part_scatter : same chunks; each reserves its range in every partition with one device atomic and writes its
//                     entries there as runs -- 4-byte payloads and, in a byte array of their own, the bucket inside the
//                     partition (5 bytes per entry; the counting pass then reads one byte per entry).
//   d. bucket_count : per partition, LDS histogram of its 128 buckets -> the global bucket counts (the plan's input).
//   e. bucket_place : after the plan: per tile of 4096 entries of a partition, LDS histogram, one device atomic
//                     per non-empty bucket to reserve the tile's run inside the bucket's list, entries written
//                     there (runs of ~32 entries).  PART_SPLIT workgroups share a partition tile by tile.
#ifndef CQ_PART_CHUNK_THREADS
#define CQ_PART_CHUNK_THREADS 256
#endif
constexpr uint32_t DIGITS_LDS_THREADS = CQ_PART_CHUNK_THREADS;
constexpr uint32_t DIGITS_LDS_PER_LANE = 8;
constexpr uint32_t DIGITS_LDS_CHUNK = DIGITS_LDS_THREADS * DIGITS_LDS_PER_LANE;
constexpr uint32_t DIGITS_LDS_MAX_M = 1u << 14;
constexpr uint32_t PART_BITS = 7, PART_BUCKETS = 1u << PART_BITS;  // buckets per partition (c <= 15 and behind a refinement pass)
constexpr uint32_t PART_BITS_WIDE = 8;                             // ... 256 of them for c = 16 and c = 17
constexpr uint32_t PART_MAX = 256;                                 // partitions per MSM (128 up to c = 16, 256 for c = 17)
constexpr uint32_t PART_THREADS = 256, PART_PER_LANE = 16, PART_TILE = PART_THREADS * PART_PER_LANE;
constexpr uint32_t PART_SPLIT = 8;  // workgroups per partition in the bucket passes
static_assert(PART_MAX <= DIGITS_LDS_THREADS && (1u << PART_BITS_WIDE) <= PART_THREADS, "one thread per histogram bin");

// CW != 0: the window width and count are compile-time constants (the tables' c = 15, 17 windows): the digit loop unrolls,
// every limb index of the canonical scalar is an immediate (with a run-time window index the limbs sat in scratch memory)
// `shift`: bucket -> partition of the first pass (PART_BITS up to c = 15; c - 8, i.e. always 128 partitions, beyond)
static constexpr uint32_t part_shift_of(uint32_t c) { return c > 17 ? c - 8 : c > 15 ? PART_BITS_WIDE : PART_BITS; }
template <uint32_t CW, uint32_t NW>
__global__ __launch_bounds__(DIGITS_LDS_THREADS) void msm_part_hist_kernel(const Fr* const* __restrict__ scalars,
                                                                           const uint64_t* __restrict__ lens, uint32_t c_, uint32_t nwin_,
                                                                           uint32_t npart, uint32_t per_lane, uint32_t* __restrict__ psize) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t hist[PART_MAX];
  const uint32_t c = CW ? CW : c_, nwin = CW ? NW : nwin_;
  const uint32_t shift = part_shift_of(c);
  const uint32_t m = blockIdx.y, t = threadIdx.x;
  const uint32_t len = (uint32_t)lens[m];
  const uint32_t base = blockIdx.x * DIGITS_LDS_THREADS * per_lane;
  if (base >= len) return;  // block-uniform
  if (t < npart) hist[t] = 0;
  __syncthreads();
  for (uint32_t k = 0; k < per_lane; k++) {
    const uint32_t i = base + k * DIGITS_LDS_THREADS + t;
    if (i >= len) continue;
    const U256 v = scalars[m][i].to_canonical();
    uint32_t carry = 0, neg;
#pragma unroll
    for (uint32_t w = 0; w < (CW ? NW : 32u); w++) {
      if (!CW && w >= nwin) break;
      const uint32_t d = signed_digit(v, w, c, carry, neg);
      if (d) atomicAdd(&hist[(d - 1) >> shift], 1u);
    }
  }
  __syncthreads();
  if (t < npart && hist[t]) atomicAdd(&psize[m * npart + t], hist[t]);
}

// exclusive scan of P <= 4096 values in one block: poff[0..P], poff[P] = total
// `s1_out` (first pass only): the sub-list length of the accumulate level for THIS launch's entries -- the host picks it from
// the bound batch x windows x n, but a launch of sparse or small scalars (advice columns of 24-bit limbs on a tenth of the
// rows, multiplicities) has a fiftieth of that, and its accumulate kernel is then a chain of s1 dependent additions on a
// fraction of the SIMDs: msm_small_launch_s1 of the entries actually there (an MSM that reads another one's lists counts
// with that one's), never more than the host's value.
__global__ __launch_bounds__(1024) void msm_part_scan_kernel(const uint32_t* __restrict__ psize, uint32_t P, uint32_t* __restrict__ poff,
                                                             const uint64_t* __restrict__ list_src = nullptr, uint32_t batch = 0,
                                                             uint32_t npart = 0, uint32_t s1_host = 0, uint32_t* __restrict__ s1_out = nullptr) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t part[1024];
  const uint32_t per = (P + 1023) / 1024;
  const uint32_t lo = min(threadIdx.x * per, P), hi = min(lo + per, P);
  uint32_t s = 0;
  for (uint32_t j = lo; j < hi; j++) s += psize[j];
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (uint32_t j = lo; j < hi; j++) {
    poff[j] = run;
    run += psize[j];
  }
  if (threadIdx.x == 1023) poff[P] = part[1023];
  if (s1_out) {
    __syncthreads();  // poff is complete (this block wrote all of it)
    if (threadIdx.x == 0) {
      uint64_t entries = 0;
      for (uint32_t m = 0; m < batch; m++) {
        const uint32_t src = (uint32_t)list_src[m];
        entries += poff[(src + 1) * npart] - poff[src * npart];
      }
      const uint32_t s1 = msm_small_launch_s1(entries);  // (the host's value stands for a launch that is not small)
      *s1_out = entries < (uint64_t)MSM_S1 * MSM_SMALL_LANES && s1 < s1_host ? s1 : s1_host;
    }
  }
}

// exclusive scan of N (128 or 256) histogram bins by the first wave: lane l owns bins l * N/64 .. (l + 1) * N/64 - 1
template <uint32_t N>
static __device__ __forceinline__ void wave0_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ start) {
  constexpr uint32_t PER = N / 64;
  const uint32_t l = threadIdx.x;
  if (l >= 64) return;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    v[k] = hist[PER * l + k];
    sum += v[k];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if ((int)l >= d) incl += o;
  }
  uint32_t run = incl - sum;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    start[PER * l + k] = run;
    run += v[k];
  }
}
static __device__ __forceinline__ void wave0_scan128(const uint32_t* __restrict__ hist, uint32_t* __restrict__ start) {
  wave0_scan<128>(hist, start);
}

// One scalar per lane; the chunk's entries are sorted by partition in LDS first, so that the write-out is a linear
// copy of runs (~34 entries = 272 B per partition) instead of one uncoalesced 8-byte store per entry (which kept the
// per-CU memory pipeline, not HBM, busy for 0.5 ms on a 21-MSM launch).
constexpr uint32_t PSC_THREADS = 256, PSC_MAX_WIN = 17;
// LowT: the bucket inside the partition -- a byte up to c = 15 (7 bits), 16 bits for the wider windows (c - 8 bits)
template <uint32_t CW, uint32_t NW, typename LowT>
__global__ __launch_bounds__(PSC_THREADS) void msm_part_scatter_kernel(const Fr* const* __restrict__ scalars,
                                                                       const uint64_t* __restrict__ lens, uint32_t c_, uint32_t nwin_,
                                                                       uint32_t npart, const uint32_t* __restrict__ poff,
                                                                       uint32_t* __restrict__ pcursor, uint32_t* __restrict__ part_pay,
                                                                       LowT* __restrict__ part_low) {
  CQ_CRITICAL_WAVES();
  __shared__ uint2 stage[PSC_THREADS * (CW ? NW : PSC_MAX_WIN)];
  __shared__ uint32_t hist[PART_MAX], lstart[PART_MAX], gbase[PART_MAX];
  const uint32_t c = CW ? CW : c_, nwin = CW ? NW : nwin_;
  const uint32_t shift = part_shift_of(c);
  const uint32_t m = blockIdx.y, t = threadIdx.x;
  const uint32_t len = (uint32_t)lens[m];
  const uint32_t i = blockIdx.x * PSC_THREADS + t;
  if (blockIdx.x * PSC_THREADS >= len) return;  // block-uniform
  if (t < PART_MAX) hist[t] = 0;
  U256 v;
  if (i < len) v = scalars[m][i].to_canonical();
  else for (int q = 0; q < 8; q++) v.l[q] = 0;  // zero scalar: no non-zero digit
  __syncthreads();
  // the digits once, kept for the placement below: |digit| - 1 in the low half, the sign in bit 31, 0xffffffff = zero digit
  uint32_t dg[CW ? NW : PSC_MAX_WIN];
  {
    uint32_t carry = 0, neg;
#pragma unroll
    for (uint32_t w = 0; w < (CW ? NW : PSC_MAX_WIN); w++) {
      dg[w] = 0xffffffffu;
      if (!CW && w >= nwin) continue;
      const uint32_t d = signed_digit(v, w, c, carry, neg);
      if (d) {
        dg[w] = (d - 1) | (neg << 31);
        atomicAdd(&hist[(d - 1) >> shift], 1u);
      }
    }
  }
  __syncthreads();
  wave0_scan<PART_MAX>(hist, lstart);
  if (t < npart) {
    const uint32_t h = hist[t];
    gbase[t] = h ? poff[m * npart + t] + atomicAdd(&pcursor[m * npart + t], h) : 0u;
  }
  __syncthreads();
  const uint32_t total = lstart[PART_MAX - 1] + hist[PART_MAX - 1];
  __syncthreads();
  if (t < PART_MAX) hist[t] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t w = 0; w < (CW ? NW : PSC_MAX_WIN); w++) {
    if (dg[w] == 0xffffffffu) continue;
    const uint32_t bk = dg[w] & 0x7fffffffu;
    const uint32_t pt = bk >> shift;
    // entry: point index (26 bits) | window << 26 | sign << 31, and the bucket inside the MSM
    stage[lstart[pt] + atomicAdd(&hist[pt], 1u)] = make_uint2(i | (w << 26) | (dg[w] & 0x80000000u), bk);
  }
  __syncthreads();
  // 5 bytes per entry leave the chunk: the payload and, in a byte array of its own, the bucket inside the partition (the
  // bucket-count pass then reads one byte per entry, the placement pass five, instead of eight each)
  for (uint32_t j = t; j < total; j += PSC_THREADS) {
    const uint2 e = stage[j];
    const uint32_t pt = e.y >> shift;
    const uint32_t dst = gbase[pt] + (j - lstart[pt]);
    part_pay[dst] = e.x;
    part_low[dst] = (LowT)(e.y & ((1u << shift) - 1u));
  }
}

// partition pid = msm * npart + p owns the flat buckets [pid << PART_BITS, (pid + 1) << PART_BITS)
// (PB: bits of the bucket inside a partition, 7 or 8)
template <uint32_t PB>
__global__ __launch_bounds__(PART_THREADS) void msm_bucket_count_kernel(const uint8_t* __restrict__ part_low, const uint32_t* __restrict__ poff,
                                                                        uint32_t* __restrict__ counts) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t hist[1u << PB];
  const uint32_t pid = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = poff[pid], hi = poff[pid + 1];
  if (lo + blockIdx.y * PART_THREADS >= hi) return;  // block-uniform
  if (t < (1u << PB)) hist[t] = 0;
  __syncthreads();
  for (uint32_t e = lo + blockIdx.y * PART_THREADS + t; e < hi; e += gridDim.y * PART_THREADS) atomicAdd(&hist[part_low[e]], 1u);
  __syncthreads();
  if (t < (1u << PB) && hist[t]) atomicAdd(&counts[(pid << PB) + t], hist[t]);
}

// an MSM that accumulates from another one's lists (same scalar vector) takes a copy of its bucket counts
__global__ __launch_bounds__(256) void msm_alias_counts_kernel(uint32_t* __restrict__ counts, const uint64_t* __restrict__ list_src, uint32_t B) {
  CQ_CRITICAL_WAVES();
  const uint32_t m = blockIdx.y, src = (uint32_t)list_src[m];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (src != m && i < B) counts[(size_t)m * B + i] = counts[(size_t)src * B + i];
}

// Same idea one level down: a tile's entries are sorted by bucket in LDS, then copied out run by run (~32 entries =
// one 128-byte line per bucket and tile).
template <uint32_t PB>
__global__ __launch_bounds__(PART_THREADS) void msm_bucket_place_kernel(const uint32_t* __restrict__ part_pay, const uint8_t* __restrict__ part_low,
                                                                        const uint32_t* __restrict__ poff,
                                                                        const uint32_t* __restrict__ off0, uint32_t* __restrict__ cursor,
                                                                        uint32_t* __restrict__ sorted) {
  CQ_CRITICAL_WAVES();
  constexpr uint32_t NB = 1u << PB;
  __shared__ uint32_t hist[NB], lstart[NB], base[NB];
  __shared__ uint32_t spay[PART_TILE];
  __shared__ uint8_t sbkt[PART_TILE];
  const uint32_t pid = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = poff[pid], hi = poff[pid + 1];
  const uint32_t ntiles = (hi - lo + PART_TILE - 1) / PART_TILE;
  for (uint32_t tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {  // block-uniform trip count
    const uint32_t tlo = lo + tile * PART_TILE, thi = min(hi, tlo + PART_TILE);
    if (t < NB) hist[t] = 0;
    __syncthreads();
    uint2 ent[PART_PER_LANE];
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_LANE; k++) {
      const uint32_t e = tlo + k * PART_THREADS + t;
      if (e < thi) {
        ent[k] = make_uint2(part_pay[e], part_low[e]);
        atomicAdd(&hist[ent[k].y], 1u);
      }
    }
    __syncthreads();
    wave0_scan<NB>(hist, lstart);
    if (t < NB) {
      const uint32_t h = hist[t], g = (pid << PB) + t;
      base[t] = h ? off0[g] + atomicAdd(&cursor[g], h) : 0u;
    }
    __syncthreads();
    if (t < NB) hist[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_LANE; k++) {
      const uint32_t e = tlo + k * PART_THREADS + t;
      if (e < thi) {
        const uint32_t b = ent[k].y;
        const uint32_t lp = lstart[b] + atomicAdd(&hist[b], 1u);
        spay[lp] = ent[k].x;
        sbkt[lp] = (uint8_t)b;
      }
    }
    __syncthreads();
    for (uint32_t j = t; j < thi - tlo; j += PART_THREADS) {
      const uint32_t b = sbkt[j];
      sorted[base[b] + (j - lstart[b])] = spay[j];
    }
    __syncthreads();
  }
}

// Wide windows (c > 17, 2^(c-1) buckets per MSM): the first pass still cuts an MSM into 128 partitions -- now of
// 2^(c-8) buckets -- and a REFINEMENT pass of the same shape as the two kernels above splits every such partition into
// its 2^(c-15) final partitions of 128 buckets, which the kernels above then finish.  The few bins of a partition are
// spread over 128 histogram cells (bin, thread mod R) so that the LDS atomics of a wave do not pile up on 2..32
// addresses; cells of one bin are adjacent, so the LDS-sorted tile is still a sequence of per-bin runs.
__global__ __launch_bounds__(PART_THREADS) void msm_refine_count_kernel(const uint16_t* __restrict__ low1, const uint32_t* __restrict__ poff1,
                                                                        uint32_t bins_log, uint32_t* __restrict__ psize2) {
  __shared__ uint32_t hist[PART_BUCKETS];
  const uint32_t cp = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = poff1[cp], hi = poff1[cp + 1];
  if (lo + blockIdx.y * PART_THREADS >= hi) return;  // block-uniform
  const uint32_t rlog = PART_BITS - bins_log, rep = t & ((1u << rlog) - 1u);
  if (t < PART_BUCKETS) hist[t] = 0;
  __syncthreads();
  for (uint32_t e = lo + blockIdx.y * PART_THREADS + t; e < hi; e += gridDim.y * PART_THREADS)
    atomicAdd(&hist[((uint32_t)(low1[e] >> PART_BITS) << rlog) | rep], 1u);
  __syncthreads();
  if (t < (1u << bins_log)) {
    uint32_t sum = 0;
    for (uint32_t r = 0; r < (1u << rlog); r++) sum += hist[(t << rlog) + r];
    if (sum) atomicAdd(&psize2[(cp << bins_log) + t], sum);
  }
}

__global__ __launch_bounds__(PART_THREADS) void msm_refine_place_kernel(const uint32_t* __restrict__ pay1, const uint16_t* __restrict__ low1,
                                                                        const uint32_t* __restrict__ poff1, uint32_t bins_log,
                                                                        const uint32_t* __restrict__ poff2, uint32_t* __restrict__ cursor2,
                                                                        uint32_t* __restrict__ pay2, uint8_t* __restrict__ low2) {
  __shared__ uint32_t hist[PART_BUCKETS], lstart[PART_BUCKETS], base[PART_BUCKETS];
  __shared__ uint32_t spay[PART_TILE];
  __shared__ uint16_t slow[PART_TILE];
  const uint32_t cp = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = poff1[cp], hi = poff1[cp + 1];
  const uint32_t ntiles = (hi - lo + PART_TILE - 1) / PART_TILE;
  const uint32_t bins = 1u << bins_log, rlog = PART_BITS - bins_log, rep = t & ((1u << rlog) - 1u);
  for (uint32_t tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {  // block-uniform trip count
    const uint32_t tlo = lo + tile * PART_TILE, thi = min(hi, tlo + PART_TILE);
    if (t < PART_BUCKETS) hist[t] = 0;
    __syncthreads();
    uint2 ent[PART_PER_LANE];
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_LANE; k++) {
      const uint32_t e = tlo + k * PART_THREADS + t;
      if (e < thi) {
        ent[k] = make_uint2(pay1[e], low1[e]);
        atomicAdd(&hist[((ent[k].y >> PART_BITS) << rlog) | rep], 1u);
      }
    }
    __syncthreads();
    wave0_scan128(hist, lstart);
    __syncthreads();
    if (t < bins) {  // the tile's run inside final partition (cp, t)
      const uint32_t first = lstart[t << rlog];
      const uint32_t end = t + 1 < bins ? lstart[(t + 1) << rlog] : lstart[PART_BUCKETS - 1] + hist[PART_BUCKETS - 1];
      const uint32_t f = (cp << bins_log) + t;
      base[t] = end > first ? poff2[f] + atomicAdd(&cursor2[f], end - first) : 0u;
    }
    __syncthreads();
    if (t < PART_BUCKETS) hist[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_LANE; k++) {
      const uint32_t e = tlo + k * PART_THREADS + t;
      if (e < thi) {
        const uint32_t cell = ((ent[k].y >> PART_BITS) << rlog) | rep;
        const uint32_t lp = lstart[cell] + atomicAdd(&hist[cell], 1u);
        spay[lp] = ent[k].x;
        slow[lp] = (uint16_t)ent[k].y;
      }
    }
    __syncthreads();
    for (uint32_t j = t; j < thi - tlo; j += PART_THREADS) {
      const uint32_t l = slow[j], bin = l >> PART_BITS;
      const uint32_t dst = base[bin] + (j - lstart[bin << rlog]);
      pay2[dst] = spay[j];
      low2[dst] = (uint8_t)(l & (PART_BUCKETS - 1));
    }
    __syncthreads();
  }
}

// ---- 2. plan: exclusive scans over the flat bucket array -----------------------------------------
// Sequence 0 is the histogram itself (-> list start of every bucket); sequence k >= 1 is the
// number of level-k sub-lists of every bucket: t1 = ceil(cnt/S1); t_k = ceil(t_{k-1}/S2) while t_{k-1} > MSM_SHORT_MIN,
// else 0: a bucket with 2..MSM_SHORT_MIN partial sums is always finished by the lane groups of msm_combine_level_kernel,
// which are indexed by bucket and need no slot; one with a single partial sum was written to its bucket by the level
// before.  Lists of MSM_SHORT_MIN+1..MSM_SHORT partial sums own a slot (one wave) that is used only when the launch
// has few of them -- see combine_short_limit.
static __device__ __forceinline__ uint32_t level_value(uint32_t cnt, uint32_t lv, uint32_t s1) {
  if (lv == 0) return cnt;
  uint32_t t = (cnt + s1 - 1) / s1;
  for (uint32_t k = 2; k <= lv; k++) t = t <= MSM_SHORT_MIN ? 0 : (t + MSM_S2 - 1) / MSM_S2;
  return t;
}
// Who sums a list of MSM_SHORT_MIN < t <= MSM_SHORT partial sums?  A lane group walks it serially (t / Q dependent
// additions, work-optimal), a wave sums it as a tree (one load per lane and six shuffle levels: ~8x the lane-operations,
// a third of the depth).  With hundreds of thousands of such lists the launch is throughput-bound and the lane groups
// win; with a few (a sparse or skewed column: the SHA limb columns, the multiplicities m) the launch is waiting for its
// longest dependent chain and the waves win.  Decided on the device from the number of wave slots of the level
// (`slots` = off_cur[Bt], already there from the plan scan), the same way by both halves of the kernel.
static __device__ __forceinline__ uint32_t combine_short_limit(uint32_t slots) {
  return slots <= MSM_WAVE_BUDGET ? MSM_SHORT_MIN : MSM_SHORT;
}

// exclusive scan of one value per thread over a 256-thread block; returns prefix, total via out param
static __device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t* sh /*>=4*/, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if ((int)lane >= off) x += y;
  }
  __syncthreads();
  if (lane == 63) sh[wid] = x;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t s = sh[k];
    if (k < wid) wbase += s;
    tot += s;
  }
  total = tot;
  return wbase + x - v;
}

constexpr uint32_t SCAN_TILE = 2048;  // elements per block (8 per thread)

__global__ __launch_bounds__(256) void msm_scan_reduce_kernel(const uint32_t* __restrict__ cnt, uint32_t Bt, uint32_t nseq, uint32_t s1,
                                                              const uint32_t* __restrict__ s1_dev /* the launch's own choice, or null */,
                                                              uint32_t* blocksums /*[nseq][nblk]*/, uint32_t* ticket) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t sh[4];
  if (s1_dev) s1 = *s1_dev;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  uint32_t c8[8];
  for (int k = 0; k < 8; k++) c8[k] = (base + k < Bt) ? cnt[base + k] : 0;
  for (uint32_t lv = 0; lv < nseq; lv++) {
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) s += level_value(c8[k], lv, s1);
    uint32_t tot;
    block_excl_scan256(s, sh, tot);
    if (threadIdx.x == 0) blocksums[(size_t)lv * gridDim.x + blockIdx.x] = tot;
  }
  // The block that finishes LAST turns the tile totals into exclusive offsets (what msm_scan_spine_kernel did in a launch of
  // its own, ~10 us of a front): a ticket per block behind a device-wide fence; the ticket word lies in the region the
  // launch's first kernel clears.
  __shared__ uint32_t last;
  __threadfence();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  const uint32_t nblk = gridDim.x;
  const uint32_t per = (nblk + 255) / 256;
  const uint32_t lo = min(threadIdx.x * per, nblk), hi = min(lo + per, nblk);
  for (uint32_t lv = 0; lv < nseq; lv++) {
    volatile uint32_t* a = blocksums + (size_t)lv * nblk;
    uint32_t s = 0;
    for (uint32_t j = lo; j < hi; j++) s += a[j];
    uint32_t tot;
    uint32_t run = block_excl_scan256(s, sh, tot);
    for (uint32_t j = lo; j < hi; j++) {
      const uint32_t v = a[j];
      a[j] = run;
      run += v;
    }
  }
}

__global__ __launch_bounds__(256) void msm_scan_apply_kernel(const uint32_t* __restrict__ cnt, uint32_t Bt, uint32_t nseq, uint32_t s1,
                                                             const uint32_t* __restrict__ s1_dev,
                                                             const uint32_t* __restrict__ blocksums,
                                                             uint32_t* __restrict__ off /*[nseq][Bt+1]*/,
                                                             uint32_t* __restrict__ tk /*[nseq-1][Bt]*/) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t sh[4];
  if (s1_dev) s1 = *s1_dev;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  uint32_t c8[8];
  for (int k = 0; k < 8; k++) c8[k] = (base + k < Bt) ? cnt[base + k] : 0;
  for (uint32_t lv = 0; lv < nseq; lv++) {
    uint32_t v8[8];
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) {
      v8[k] = level_value(c8[k], lv, s1);
      s += v8[k];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan256(s, sh, tot) + blocksums[(size_t)lv * gridDim.x + blockIdx.x];
    uint32_t* o = off + (size_t)lv * (Bt + 1);
    for (int k = 0; k < 8; k++) {
      if (base + k < Bt) {
        o[base + k] = run;
        if (lv) tk[(size_t)(lv - 1) * Bt + base + k] = v8[k];
      }
      run += v8[k];
      if (base + k == Bt - 1) o[Bt] = run;
    }
  }
}

// ---- 3. scatter (counting sort) ----------------------------------------------------------------
// Recomputes the digits (cheaper than storing them) and drops every entry at list start + rank.
__global__ __launch_bounds__(256) void msm_scatter_kernel(const Fr* const* __restrict__ scalars, const uint64_t* __restrict__ lens,
                                                          uint32_t nmax, uint32_t c, uint32_t nwin, uint32_t pre,
                                                          const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ off0,
                                                          uint32_t* __restrict__ sorted) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  if (i >= (uint32_t)lens[m]) return;
  const U256 v = scalars[m][i].to_canonical();
  const uint32_t M = 1u << (c - 1);
  uint32_t carry = 0, neg;
  for (uint32_t w = 0; w < nwin; w++) {
    const uint32_t d = signed_digit(v, w, c, carry, neg);
    if (!d) continue;
    const size_t mw = (size_t)m * nwin + w;
    const uint32_t g = (uint32_t)((pre ? (size_t)m : mw) * M) + (d - 1);
    // entry: point index (26 bits) | window << 26 (precomputed-table mode) | sign << 31
    sorted[off0[g] + ranks[mw * nmax + i]] = i | (pre ? (w << 26) : 0u) | (neg << 31);
  }
}

// ---- 4. bucket accumulation ----------------------------------------------------------------
// largest g in [0,B) with off[g] <= j   (off is non-decreasing, off[0] = 0, j < off[B])
static __device__ __forceinline__ uint32_t owner_of(const uint32_t* __restrict__ off, uint32_t B, uint32_t j) {
  uint32_t lo = 0, hi = B;  // invariant: off[lo] <= j < off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}

// level 1: lane j owns sub-list j of <= MSM_S1 point indices
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accumulate_kernel(
    const G1Affine* const* __restrict__ bases, uint32_t buckets_per_msm, uint32_t pre,
    const uint64_t* __restrict__ table_strides, const uint64_t* __restrict__ list_src, const uint32_t* __restrict__ sorted,
    const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off0,
    const uint32_t* __restrict__ t1, const uint32_t* __restrict__ off1, uint32_t Bt, XYZZ* __restrict__ partial,
    XYZZ* __restrict__ buckets) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= off1[Bt]) return;
  const uint32_t g = owner_of(off1, Bt, j);
  const uint32_t msm = g / buckets_per_msm;
  const G1Affine* __restrict__ pts = bases[msm];
  const size_t table_stride = pre ? (size_t)table_strides[msm] : 0;
  const uint32_t r = j - off1[g];
  // the bucket's cnt entries are cut into t = ceil(cnt / s1) sub-lists of EQUAL length (+-1): a wave runs as
  // long as its longest lane, so 40 entries are better served as 20 + 20 than as 32 + 8
  // an MSM over the same scalars as an earlier one of the launch (b0 and p of a CQ lookup) reads that one's lists
  const uint32_t list0 = off0[(uint32_t)list_src[msm] * buckets_per_msm + (g - msm * buckets_per_msm)];
  const uint32_t t = t1[g], n_g = cnt[g];
  const uint32_t lo = list0 + (uint32_t)(((uint64_t)n_g * r) / t);
  const uint32_t hi = list0 + (uint32_t)(((uint64_t)n_g * (r + 1)) / t);
  XYZZ29 acc = XYZZ29::identity();
  // Software pipeline: the table point of entry e+1 (a dependent, essentially random 64-byte gather) and the
  // index of entry e+2 are requested before the ~10 products of entry e are computed.
  // precomputed mode: table[w][i] = 2^(c*w) * base[i], stored in the kernels' R' limb form; otherwise the
  // caller's array of R = 2^256 values is converted on the fly
  auto addr_of = [&](uint32_t ix) {
    return pre ? (size_t)((ix >> 26) & 31u) * table_stride + (ix & 0x03ffffffu) : (size_t)(ix & 0x7fffffffu);
  };
  uint32_t ix_cur = lo < hi ? sorted[lo] : 0u;
  uint32_t ix_next = lo + 1 < hi ? sorted[lo + 1] : 0u;
  uint32_t wx[8], wy[8];
  if (lo < hi) {
    const G1Affine* q = pts + addr_of(ix_cur);
    ld8(&q->x, wx);
    ld8(&q->y, wy);
  }
  for (uint32_t e = lo; e < hi; e++) {
    Affine29 p;
    p.x = Fq29::unpack(wx);
    p.y = Fq29::unpack(wy);
    const uint32_t neg = ix_cur >> 31;
    ix_cur = ix_next;
    if (e + 1 < hi) {
      const G1Affine* q = pts + addr_of(ix_cur);
      ld8(&q->x, wx);
      ld8(&q->y, wy);
    }
    ix_next = e + 2 < hi ? sorted[e + 2] : 0u;
    if (!pre) {  // R = 2^256 values of a caller's array
      Fq29 f;
      CQ_UNROLL for (int i = 0; i < 9; i++) f.a[i] = CONSTS29<FqP>.from256[i];
      p.x = Fq29::mul(p.x, f);
      p.y = Fq29::mul(p.y, f);
    }
    if (neg && !p.is_identity()) p.y = Fq29::neg<2>(p.y);
    xyzz29_add_affine(acc, p);
  }
  store_xyzz29(t == 1 ? buckets + g : partial + j, acc);
}

// Level k >= 2 in ONE launch.  Blocks [0, short_blocks): Q lanes per BUCKET sum the short lists (2..limit partial sums
// of level k-1, always the whole bucket): indexed by bucket, no search, a bucket that is empty, already final (one
// partial sum) or long costs one load; a group sums strided parts and folds them with log2(Q) shuffles.  The remaining
// blocks: one WAVE per sub-list of <= MSM_S2 partial sums of the longer lists (strided lane sums + 6-step shuffle tree);
// only those lists own slots at level k, so off_cur[Bt] is small (usually zero) and almost every wave leaves at the
// first test.  The two halves write different buckets and do not depend on each other: one launch instead of two
// keeps the level's latency at the longer of the two chains.
template <uint32_t Q>
__global__ __launch_bounds__(256, 3) void msm_combine_level_kernel(
    const XYZZ* __restrict__ prev, const uint32_t* __restrict__ t_prev, const uint32_t* __restrict__ off_prev,
    const uint32_t* __restrict__ t_cur, const uint32_t* __restrict__ off_cur, uint32_t Bt, uint32_t short_blocks,
    XYZZ* __restrict__ partial, XYZZ* __restrict__ buckets) {
  CQ_CRITICAL_WAVES();
  const uint32_t slots = off_cur[Bt];
  const uint32_t limit = combine_short_limit(slots);
  if (blockIdx.x < short_blocks) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = gt / Q, sub = gt % Q;
    const uint32_t tp = g < Bt ? t_prev[g] : 0;
    const bool mine = tp >= 2 && tp <= limit;
    if (!__any(mine)) return;  // wave-uniform
    const uint32_t lo = mine ? off_prev[g] : 0;
    XYZZ29 acc = XYZZ29::identity();
    if (mine && sub < tp) {  // the next partial sum is on its way while the current one is added
      XYZZRaw nxt = load_xyzz_raw(prev + lo + sub);
      for (uint32_t e = sub; e < tp; e += Q) {
        const XYZZ29 cur = unpack_xyzz29(nxt);
        if (e + Q < tp) nxt = load_xyzz_raw(prev + lo + e + Q);
        xyzz29_add(acc, cur);
      }
    }
#pragma unroll 1
    for (int delta = Q / 2; delta >= 1; delta >>= 1) {
      XYZZ29 o = xyzz29_shfl_down(acc, delta);  // all lanes take part in the shuffle
      if (mine && sub + delta < Q) xyzz29_add(acc, o);
    }
    if (mine && sub == 0) store_xyzz29(buckets + g, acc);
    return;
  }
  // (The slots are few -- usually none -- but their BOUND, which is all the host knows, is in the millions at k >= 20: the
  // wave half is a capped number of blocks that stride over the slots, not one block per four possible slots; the empty
  // blocks alone were ~0.3 ms of a k = 20 proof and ~1 ms at k = 22.)
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = (gridDim.x - short_blocks) * (blockDim.x >> 6);
#pragma unroll 1
  // (j through readfirstlane: wave-uniform, and everything derived from it -- the owner search, the list bounds -- stays in
  // scalar registers; as a vector value the loop took the kernel over the 168 registers of its three waves per SIMD, into scratch)
  for (uint32_t j = __builtin_amdgcn_readfirstlane((blockIdx.x - short_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6)); j < slots; j += nwaves) {
  const uint32_t g = owner_of(off_cur, Bt, j);
  const uint32_t tp = t_prev[g];
  if (tp <= limit) continue;  // a lane group of the other half owns this list (throughput-bound launch)
  const uint32_t r = j - off_cur[g];
  const uint32_t lo = off_prev[g] + r * MSM_S2;
  const uint32_t hi = min(off_prev[g] + tp, lo + MSM_S2);
  // Four lanes per addition (quad_add, curve29.hpp): this half only runs for the few long lists of a launch, which the
  // launch then waits for.  Quad q sums the partial sums lo + q, lo + q + 16, .. (lane r of the quad their coordinate r),
  // an XOR butterfly over the quads that hold data folds them: <= 4 + 4 additions of ~2.4 us instead of 1 + 6 of ~5 us.
  const uint32_t role = lane & 3u, quad = lane >> 2, len = hi - lo;
  auto load_coord = [&](uint32_t e) {
    uint32_t w[8];
    ld8(reinterpret_cast<const char*>(prev + e) + 32 * role, w);
    return Fq29::unpack(w);
  };
  Fq29 F = Fq29::zero();
  if (quad < len) F = load_coord(lo + quad);
#pragma unroll 1
  for (uint32_t e = quad + 16; e < ((len + 15u) & ~15u); e += 16) {  // (the same trip count for every quad: the permutes need all lanes)
    Fq29 G = Fq29::zero();
    if (e < len) G = load_coord(lo + e);
    F = quad_add(F, G);
  }
  const uint32_t nq = len < 16 ? len : 16;  // quads holding data
#pragma unroll 1
  for (int d = nq > 8 ? 32 : nq > 4 ? 16 : nq > 2 ? 8 : nq > 1 ? 4 : 0; d >= 4; d >>= 1) {
    Fq29 o;
    CQ_UNROLL for (int k = 0; k < 9; k++) o.a[k] = __shfl_xor(F.a[k], d, 64);
    F = quad_add(F, o);
  }
  const Fq29 red = F.reduced();  // packed form: x below 2^256
  if (lane < 4) {
    uint32_t w[8];
    (role == 0 ? red : F).pack(w);
    st8(reinterpret_cast<char*>(t_cur[g] == 1 ? buckets + g : partial + j) + 32 * role, w);
  }
  }
}

// ---- precomputed window tables (one-off setup) -----------------------------------------------------------
// Tables are library-internal, so they hold the accumulate kernel's own representation: coordinates in
// Montgomery form with R' = 2^261, packed into 8 x u32 (field29.hpp).  Window 0 = the caller's points converted.
static __device__ __forceinline__ void store_affine29(G1Affine* dst, const Affine29& p) {
  uint32_t w[8];
  p.x.pack(w);
  st8(&dst->x, w);
  p.y.pack(w);
  st8(&dst->y, w);
}
__global__ __launch_bounds__(256) void msm_table0_kernel(const G1Affine* __restrict__ bases, G1Affine* __restrict__ table, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_affine29(table + i, load_affine29(bases + i, true));
}
// next[i] = 2^c * prev[i] (affine in, affine out): the doublings and the inversion run on the canonical field type
__global__ __launch_bounds__(256) void msm_pre_kernel(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next, uint32_t n,
                                                      uint32_t c) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine29 p29 = load_affine29(prev + i, false);
  const G1Affine p = {p29.x.to_mont256(), p29.y.to_mont256()};
  XYZZ a = xyzz_dbl_affine(p);
  for (uint32_t k = 1; k < c; k++) a = xyzz_dbl(a);
  G1Affine r = G1Affine::identity();
  if (!a.is_identity()) {
    const Fq iv = (a.zz * a.zzz).inv();
    r.x = a.x * (iv * a.zzz);
    r.y = a.y * (iv * a.zz);
  }
  store_affine29(next + i, {Fq29::from_mont256(r.x), Fq29::from_mont256(r.y)});
}

int msm_precompute_tables(cq_ctx* ctx, const G1Affine* bases, uint32_t n, uint32_t c, G1Affine* table /* W*n */) {
  const uint32_t W = (255 + c - 1) / c;
  hipStream_t s = ctx->stream;
  msm_table0_kernel<<<(n + 255) / 256, 256, 0, s>>>(bases, table, n);
  for (uint32_t w = 1; w < W; w++)
    msm_pre_kernel<<<(n + 255) / 256, 256, 0, s>>>(table + (size_t)(w - 1) * n, table + (size_t)w * n, n, c);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- 5. reduction: sum_{b=1..M} b * B_b per bucket set ---------------------------------------------------
// Every dependent EC addition costs a lone wave ~9 us whatever the number of busy lanes, so the reduction is shaped
// for depth, not work.  Buckets are read as a rows x cols matrix (b = cols * hi + lo + 1, cols = min(M, 128)):
//     sum_b b B_b  =  cols * sum_hi hi R_hi  +  sum_lo (lo + 1) C_lo,     R = row sums, C = column sums.
// Kernel 1: 16 or 64 lanes per row / column sum (a few buckets per lane serially, then a shuffle tree), all lines
// independent.  Kernel 2: per bucket set two waves, one per weighted sum of <= 128 terms
// (pair sums, suffix scan by shuffles, one tree); the host applies "cols *" and joins the two.  ~11 + ~16 dependent
// operations instead of the ~46 of a lane-serial running sum over 1024-bucket groups.
// A wave serves 64 / SEG rows (or columns): SEG lanes per line, each summing cols / SEG consecutive buckets of it
// serially, then a log2(SEG)-level shuffle tree.  SEG = 16 (depth 8 + 4, every lane busy in the serial part) when a
// launch has many bucket sets -- with one wave per line and a 6-level tree only a quarter of the lane-operations were
// useful and the kernel was VALU-throughput bound at batch 16; SEG = 64 (depth 2 + 6) for launches of a few sets,
// which are latency-bound and leave most SIMDs idle anyway.
template <uint32_t SEG>
__global__ __launch_bounds__(64) void msm_rowcol_kernel(const XYZZ* __restrict__ buckets, const uint32_t* __restrict__ t1, uint32_t M,
                                                        uint32_t rows, uint32_t cols, XYZZ* __restrict__ sums /*[sets][rows + cols]*/) {
  CQ_CRITICAL_WAVES();
  constexpr uint32_t LINES = 64 / SEG;
  const uint32_t set = blockIdx.y, lane = threadIdx.x;
  const uint32_t q = blockIdx.x * LINES + lane / SEG, seg = lane % SEG;  // line: row q or column q - rows
  const XYZZ* Bk = buckets + (size_t)set * M;
  // a bucket without entries (no level-1 sub-list) was never written -- the workspace's buckets are not cleared, 128 bytes
  // each: 180 MB per round-2 launch at k >= 20 -- and stands for the identity
  const uint32_t* Tk = t1 + (size_t)set * M;
  XYZZ29 acc = XYZZ29::identity();
  if (q < rows) {  // R_q = sum_lo B[q][lo]
    const uint32_t per = (cols + SEG - 1) / SEG;
    for (uint32_t lo = seg * per; lo < min(cols, (seg + 1) * per); lo++)
      if (Tk[(size_t)q * cols + lo]) xyzz29_add(acc, load_xyzz29(Bk + (size_t)q * cols + lo));
  } else if (q < rows + cols) {  // C_lo = sum_hi B[hi][lo]
    const uint32_t lo = q - rows, per = (rows + SEG - 1) / SEG;
    for (uint32_t hi = seg * per; hi < min(rows, (seg + 1) * per); hi++)
      if (Tk[(size_t)hi * cols + lo]) xyzz29_add(acc, load_xyzz29(Bk + (size_t)hi * cols + lo));
  }
#pragma unroll 1
  for (int delta = SEG / 2; delta >= 1; delta >>= 1) {
    XYZZ29 o = xyzz29_shfl_down(acc, delta);  // lanes seg + delta >= SEG read the next line's lanes: not used
    if (seg + delta < SEG) xyzz29_add(acc, o);
  }
  if (seg == 0 && q < rows + cols) store_xyzz29(sums + (size_t)set * (rows + cols) + q, acc);
}

// sum_{j < count} (j + first_weight) X_j, count <= 128, as sum_t 2^t S_t with S_t = the sum of the X_j whose index has
// bit t set: seven independent 64-term trees (six shuffle levels) instead of a suffix scan followed by a tree (fifteen
// dependent additions, ~190 us for a lone wave -- the longest single stretch of a launch's tail; a doubling costs a
// lone wave as much as an addition, so the "2^t" may not run here either).  The kernel stops at the S_t: the Horner
// over t, the plain total of the "+ first_weight" term, the log2(cols) doublings and the last additions are ~33 group
// operations per bucket set for the host (msm_set_value), which needs a fraction of a microsecond for each where a
// lone wave needs ~10 us.  out[set][0..6] = row planes, [7..13] = column planes, [14] = column total (MSM_SET_POINTS).
// One WAVE per block (and so, with a few dozen blocks on 256 CUs, per SIMD): two waves of such a chain on one SIMD share
// its issue slots and each runs at half speed -- eight-wave blocks made this kernel take 150-220 us for six levels.
__global__ __launch_bounds__(64) void msm_weighted_kernel(const XYZZ* __restrict__ sums, uint32_t rows, uint32_t cols,
                                                          G1Jac* __restrict__ out) {
  CQ_CRITICAL_WAVES();
  const uint32_t set = blockIdx.x / MSM_SET_POINTS, plane = blockIdx.x % MSM_SET_POINTS, lane = threadIdx.x;
  const uint32_t which = plane >= 7 ? 1u : 0u;
  const XYZZ* X = sums + (size_t)set * (rows + cols) + (which ? rows : 0);
  const uint32_t count = which ? cols : rows;
  XYZZ29 v = XYZZ29::identity();
  if (plane < 14) {
    const uint32_t t = plane - 7 * which;
    const uint32_t j = ((lane >> t) << (t + 1)) | (1u << t) | (lane & ((1u << t) - 1u));  // lane-th index with bit t set
    if (j < count) v = load_xyzz29(X + j);
  } else {
    if (lane < count) v = load_xyzz29(X + lane);
    if (lane + 64 < count) xyzz29_add(v, load_xyzz29(X + lane + 64));
  }
#pragma unroll 1
  for (int delta = 32; delta >= 1; delta >>= 1) {
    XYZZ29 o = xyzz29_shfl_down(v, delta);
    if ((int)lane < delta) xyzz29_add(v, o);  // the lanes that still matter (the others would add a point to itself: the doubling path)
  }
  if (lane == 0) out[(size_t)set * MSM_SET_POINTS + plane] = xyzz29_to_jac(v);
}

// Row / column sums with FOUR lanes per addition (quad_add), for launches of a few bucket sets: one 256-thread block per
// line of <= 128 buckets -- quad q takes buckets q and q + 64 of the line (lane r of the quad their coordinate r), then the
// same butterfly as msm_weighted_quad_kernel: seven additions of ~2.4 us deep instead of 2 + 6 of ~5 us.
__global__ __launch_bounds__(256) void msm_rowcol_quad_kernel(const XYZZ* __restrict__ buckets, const uint32_t* __restrict__ t1, uint32_t M,
                                                              uint32_t rows, uint32_t cols, XYZZ* __restrict__ sums /*[sets][rows + cols]*/) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t xs[4][4][9];
  const uint32_t set = blockIdx.y, q = blockIdx.x;  // line: row q or column q - rows
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  const XYZZ* Bk = buckets + (size_t)set * M;
  const uint32_t* Tk = t1 + (size_t)set * M;
  const bool is_row = q < rows;
  const uint32_t len = is_row ? cols : rows;
  auto load_coord = [&](uint32_t e) {  // coordinate `role` of the line's e-th bucket; identity for a bucket without entries
    Fq29 r = Fq29::zero();
    if (e < len) {
      const size_t idx = is_row ? (size_t)q * cols + e : (size_t)e * cols + (q - rows);
      if (Tk[idx]) {
        uint32_t w[8];
        ld8(reinterpret_cast<const char*>(Bk + idx) + 32 * role, w);
        r = Fq29::unpack(w);
      }
    }
    return r;
  };
  const Fq29 F0 = load_coord(quad), G0 = load_coord(quad + 64);
  Fq29 F = quad_add(F0, G0);
  auto other = [&](const Fq29& a, int lanes) {
    Fq29 r;
    CQ_UNROLL for (int k = 0; k < 9; k++) r.a[k] = __shfl_xor(a.a[k], lanes, 64);
    return r;
  };
#pragma unroll 1
  for (int d = 32; d >= 4; d >>= 1) F = quad_add(F, other(F, d));
  if (lane < 4) {
    CQ_UNROLL for (int k = 0; k < 9; k++) xs[wave][role][k] = F.a[k];
  }
  __syncthreads();
  if (wave != 0) return;
  F = Fq29::zero();
  if (lane < 16) {
    CQ_UNROLL for (int k = 0; k < 9; k++) F.a[k] = xs[lane >> 2][role][k];
  }
#pragma unroll 1
  for (int d = 8; d >= 4; d >>= 1) F = quad_add(F, other(F, d));
  // packed form: x below 2^256 needs the reduction (8 p > 2^256); the other coordinates are below 4 p
  const Fq29 red = F.reduced();
  if (lane < 4) {
    uint32_t w[8];
    (role == 0 ? red : F).pack(w);
    st8(reinterpret_cast<char*>(sums + (size_t)set * (rows + cols) + q) + 32 * role, w);
  }
}

// The same bit-plane sums with FOUR lanes per addition (quad_add, curve29.hpp), for launches of a few bucket sets -- the
// plain kernel's six tree levels are six dependent general additions of ~5 us each on a lone wave; a quad needs ~2.4.
// 256 threads = 64 quads per plane: quad q takes the plane's q-th term (lane r of the quad its coordinate r), an XOR
// butterfly over the 16 quads of a wave (4 levels, every quad ends with the wave's sum), the four waves' sums through
// LDS, two more levels in wave 0.  Results are the same group elements as msm_weighted_kernel's (sums in another order).
__global__ __launch_bounds__(256) void msm_weighted_quad_kernel(const XYZZ* __restrict__ sums, uint32_t rows, uint32_t cols,
                                                                G1Jac* __restrict__ out) {
  CQ_CRITICAL_WAVES();
  __shared__ uint32_t xs[4][4][9];
  const uint32_t set = blockIdx.x / MSM_SET_POINTS, plane = blockIdx.x % MSM_SET_POINTS;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  const uint32_t which = plane >= 7 ? 1u : 0u;
  const XYZZ* X = sums + (size_t)set * (rows + cols) + (which ? rows : 0);
  const uint32_t count = which ? cols : rows;
  auto load_coord = [&](uint32_t j) {  // coordinate `role` of X[j]: 32 bytes of the packed form
    uint32_t w[8];
    ld8(reinterpret_cast<const char*>(X + j) + 32 * role, w);
    return Fq29::unpack(w);
  };
  Fq29 F = Fq29::zero();
  if (plane < 14) {
    const uint32_t t = plane - 7 * which;
    const uint32_t j = ((quad >> t) << (t + 1)) | (1u << t) | (quad & ((1u << t) - 1u));  // quad-th index with bit t set
    if (j < count) F = load_coord(j);
  } else {
    if (quad < count) F = load_coord(quad);
    Fq29 G = Fq29::zero();
    if (quad + 64 < count) G = load_coord(quad + 64);
    F = quad_add(F, G);
  }
  auto other = [&](const Fq29& a, int lanes) {
    Fq29 r;
    CQ_UNROLL for (int k = 0; k < 9; k++) r.a[k] = __shfl_xor(a.a[k], lanes, 64);
    return r;
  };
#pragma unroll 1
  for (int d = 32; d >= 4; d >>= 1) F = quad_add(F, other(F, d));  // 16 quads of the wave -> every quad holds their sum
  if (lane < 4) {
    CQ_UNROLL for (int k = 0; k < 9; k++) xs[wave][role][k] = F.a[k];
  }
  __syncthreads();
  if (wave != 0) return;
  F = Fq29::zero();
  if (lane < 16) {
    CQ_UNROLL for (int k = 0; k < 9; k++) F.a[k] = xs[lane >> 2][role][k];
  }
#pragma unroll 1
  for (int d = 8; d >= 4; d >>= 1) F = quad_add(F, other(F, d));
  const XYZZ29 v = {quad_perm<0, 0, 0, 0>(F), quad_perm<1, 1, 1, 1>(F), quad_perm<2, 2, 2, 2>(F), quad_perm<3, 3, 3, 3>(F)};
  if (lane == 0) out[(size_t)set * MSM_SET_POINTS + plane] = xyzz29_to_jac(v);
}

// -------------------------------------------------------------------------------------------------
uint32_t msm_window_bits(uint32_t n) {
  // Measured on MI355X (tests/perf/msm_sweep.py, uniform scalars).  15 is also the width whose top
  // window (bits 240..254) is well filled; widths that leave 1-7 bits for the top window pile
  // n/2^bits entries on a handful of buckets.  (The reference uses ceil(ln n), arithmetic.rs:16-22.)
  if (n >= (1u << 15)) return 15;
  if (n >= (1u << 13)) return 10;
  if (n >= 256) return 8;
  if (n >= 16) return 4;
  return 2;
}

MsmLayout::MsmLayout(uint32_t n_, uint32_t c_, uint32_t batch_, bool pre_) : n(n_), c(c_), batch(batch_), pre(pre_) {
  W = (255 + c - 1) / c;  // signed digits need W*c >= 255 for 254-bit scalars
  // bucket sets per MSM: one per window in plain mode; with per-window tables every window feeds the same 2^(c-1)
  // buckets, kept as sets of at most 2^14 for the reduction
  M = pre && c > 15 ? 1u << 14 : 1u << (c - 1);
  Wb = pre ? (1u << (c - 1)) / M : W;
  B = Wb * M;
  Bt = batch * B;
  levels = 1;
  {
    uint64_t cap = msm_small_launch_s1((uint64_t)batch * W * n);
    const uint64_t maxlist = pre ? (uint64_t)n * W : n;  // longest possible bucket list
    while (cap < maxlist) { cap *= MSM_S2; levels++; }
    // a launch that turns out to hold few entries cuts shorter sub-lists (msm_part_scan_kernel): a list is then at most
    // MSM_SMALL_PARTIALS partial sums long whatever its length
    uint32_t lv2 = 1;
    for (uint64_t t = std::min<uint64_t>((maxlist + MSM_S1_MIN - 1) / MSM_S1_MIN, MSM_SMALL_PARTIALS); t > 1; t = (t + MSM_S2 - 1) / MSM_S2) lv2++;
    levels = std::max(levels, lv2);
  }
  nseq = levels + 1;
  nblk = (Bt + 2047) / 2048;
  cols = M < 128 ? M : 128;  // bucket matrix of the reduction (msm_rowcol_kernel)
  rows = M / cols;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const uint64_t E = (uint64_t)batch * W * n;  // upper bound on (point,digit) entries
  size_t o = 0;
  off_ptrs = o;    o = up(o + (size_t)5 * batch * sizeof(void*));
  // table mode: (payload, bucket) pairs of the partition pass; plain mode: one rank per entry
  part_sort = pre && B <= (DIGITS_LDS_MAX_M << (MSM_TABLE_C_MAX - 15)) && B >= PART_BUCKETS && W <= PSC_MAX_WIN;
  mid = part_sort && c > 17;               // c = 16, 17: two passes with 256 buckets per partition
  shift1 = part_shift_of(c);
  npart = part_sort ? B >> shift1 : 0;     // partitions of the first pass (<= PART_MAX)
  nfinal = part_sort ? (mid ? B >> PART_BITS : npart) : 0;  // partitions the bucket passes work on
  // table mode: E payloads and E partition-local buckets (a byte each; with a refinement pass 16 bits, and a second pair
  // of arrays for its output)
  off_ranks = o;   o = up(o + (size_t)E * (part_sort ? (mid ? 11 : 5) : sizeof(uint32_t)));
  off_poff = o;    o = up(o + ((size_t)batch * npart + 1) * sizeof(uint32_t));
  off_poff2 = o;   o = up(o + (mid ? (size_t)batch * nfinal + 1 : 0) * sizeof(uint32_t));
  off_counts = o;  o = up(o + (size_t)Bt * sizeof(uint32_t));
  off_cursor = o;  o = up(o + (part_sort ? (size_t)Bt : 0) * sizeof(uint32_t));        // per-bucket fill cursors
  off_psize = o;   o = up(o + (size_t)2 * batch * npart * sizeof(uint32_t));          // partition sizes, cursors
  off_psize2 = o;  o = up(o + (mid ? (size_t)2 * batch * nfinal : 0) * sizeof(uint32_t));
  off_ticket = o;  o = up(o + sizeof(uint32_t));  // msm_scan_reduce_kernel's last-block ticket
  zero_end = o;    // counts .. ticket are cleared by the launch's first kernel; the buckets are not (msm_rowcol_kernel)
  off_buckets = o; o = up(o + (size_t)Bt * sizeof(XYZZ));
  off_blocksums = o; o = up(o + ((size_t)nseq * nblk + 1) * sizeof(uint32_t));  // + the launch's own s1 (msm_part_scan_kernel)
  off_off = o;     o = up(o + (size_t)nseq * (Bt + 1) * sizeof(uint32_t));
  off_tk = o;      o = up(o + (size_t)levels * Bt * sizeof(uint32_t));
  off_sorted = o;  o = up(o + (size_t)E * sizeof(uint32_t));
  // partial sums: level 1 has at most ceil(E/S1) + Bt sub-lists; level k >= 2 only reserves slots for lists of more
  // than MSM_SHORT_MIN items, so at most items/S2 + items/SHORT_MIN + 1 sub-lists
  for (uint32_t k = 0; k < 8; k++) tmax[k] = 0;
  const uint32_t s1_min = msm_small_launch_s1(E);  // MSM_S1 unless the launch is small (msm.hpp)
  tmax[0] = std::max<uint64_t>((E + s1_min - 1) / s1_min, std::min<uint64_t>((E + MSM_S1_MIN - 1) / MSM_S1_MIN, MSM_SMALL_PARTIALS)) + Bt;
  for (uint32_t k = 1; k < levels; k++) tmax[k] = tmax[k - 1] / MSM_S2 + tmax[k - 1] / MSM_SHORT_MIN + 1;
  off_part[0] = o;
  o = up(o + (size_t)tmax[0] * sizeof(XYZZ));
  off_part[1] = o;
  o = up(o + (size_t)(levels > 1 ? tmax[1] : 0) * sizeof(XYZZ));
  off_pairs = o;   o = up(o + (size_t)batch * Wb * (rows + cols) * sizeof(XYZZ));
  total = o;
}

int msm_run(cq_ctx* ctx, const Fr* const* scalars_host_ptrs, const G1Affine* const* bases_host_ptrs, const size_t* lens,
            uint32_t n, uint32_t c, uint32_t batch, bool pre, const size_t* table_strides, void* workspace,
            G1Jac* window_sums_dev) {
  hipStream_t s = ctx->stream;
  MsmLayout L(n, c, batch, pre);
  if (pre && (n > (1u << 26) || L.W > 32)) return -4;
  if (L.tmax[0] > 0xfffffff0ull || (uint64_t)batch * L.W * n > 0xfffffff0ull) return -3;
  char* ws = (char*)workspace;
  const Fr** d_scalars = (const Fr**)(ws + L.off_ptrs);
  const G1Affine** d_bases = (const G1Affine**)(ws + L.off_ptrs) + batch;
  uint32_t* ranks = (uint32_t*)(ws + L.off_ranks);
  uint32_t* counts = (uint32_t*)(ws + L.off_counts);
  XYZZ* buckets = (XYZZ*)(ws + L.off_buckets);
  uint32_t* blocksums = (uint32_t*)(ws + L.off_blocksums);
  uint32_t* ticket = (uint32_t*)(ws + L.off_ticket);
  uint32_t* s1_dev = blocksums + (size_t)L.nseq * L.nblk;  // table mode: the sub-list length the launch settles on
  uint32_t* off = (uint32_t*)(ws + L.off_off);
  uint32_t* tk = (uint32_t*)(ws + L.off_tk);
  uint32_t* sorted = (uint32_t*)(ws + L.off_sorted);
  XYZZ* part[2] = {(XYZZ*)(ws + L.off_part[0]), (XYZZ*)(ws + L.off_part[1])};
  XYZZ* pairs = (XYZZ*)(ws + L.off_pairs);
  const uint32_t M = L.M, W = L.W, Bt = L.Bt;
  if (batch == 0 || batch > MSM_MAX_BATCH) return -2;
  MsmPtrs sp, bp;
  MsmStrides stv, lnv, srcv;
  // MSMs of one launch over the SAME scalar vector (b0 and p of a CQ lookup: same coefficients, shifted bases) have
  // the same entry lists: the later one is not sorted (the sort sees length 0), takes a copy of the bucket counts and
  // accumulates from the earlier one's lists.
  uint32_t alias[MSM_MAX_BATCH];
  bool any_alias = false;
  for (uint32_t i = 0; i < MSM_MAX_BATCH; i++) {
    sp.p[i] = i < batch ? (const void*)scalars_host_ptrs[i] : nullptr;
    bp.p[i] = i < batch ? (const void*)bases_host_ptrs[i] : nullptr;
    stv.s[i] = (i < batch && pre) ? (uint64_t)table_strides[i] : 0;
    lnv.s[i] = i < batch ? (uint64_t)lens[i] : 0;
    alias[i] = i;
    if (i < batch && L.part_sort)
      for (uint32_t j = 0; j < i; j++)
        if (alias[j] == j && scalars_host_ptrs[j] == scalars_host_ptrs[i] && lens[j] == lens[i] && lens[i]) {
          alias[i] = j;
          lnv.s[i] = 0;
          any_alias = true;
          break;
        }
    srcv.s[i] = alias[i];
  }
  const uint64_t* d_strides = (const uint64_t*)((const void**)d_scalars + 2 * batch);
  const uint64_t* d_lens = (const uint64_t*)((const void**)d_scalars + 3 * batch);
  const uint64_t* d_src = (const uint64_t*)((const void**)d_scalars + 4 * batch);
  // Sub-list length of the accumulate level, from the expected load of a bucket (entries of the launch's longest
  // MSM / buckets).  Short lists keep a k = 18 launch (272 entries per bucket) parallel and balanced; with thousands
  // of entries per bucket there are lanes to spare, and the partial sums per bucket must stay few: past MSM_SHORT of
  // them a bucket falls to the wave-per-256 combine, which at k = 22 (4352 entries per bucket, 136 partial sums of
  // 32) cost 17 ms of a 133 ms proof.  (The workspace is sized for MSM_S1, the smallest value.)
  uint64_t load = 0;
  for (uint32_t i = 0; i < batch; i++) load = std::max<uint64_t>(load, (uint64_t)lens[i] * W / (L.Wb * M));
  // (Longer sub-lists for the very large launches -- fewer partial sums for the combine, 2.6 ms of a k = 20 proof -- were
  // swept at k = 20 and k = 22, 32..384 entries: the accumulate kernel loses what the combine gains, within 1 %.)
  // (With several bucket sets per MSM -- wide table windows -- there are four or more times the buckets and a quarter of
  // the partial sums per bucket keeps the lanes as many: k = 22 at c = 17, 91.8 -> 90.8 ms.)
  const uint32_t partials = pre && L.Wb > 1 ? MSM_PARTIALS_TARGET / 4 : MSM_PARTIALS_TARGET;
  const uint32_t s1 = load <= MSM_S1_BIG_LOAD ? msm_small_launch_s1((uint64_t)batch * W * n)
                                              : std::max<uint32_t>(MSM_S1_BIG, (uint32_t)((load + partials - 1) / partials));
  const uint32_t* off0 = off;
  // The launch in three segments: front (pointer tables, sort, plan), the accumulate kernel, tail (combine levels, bucket
  // reduction).  Front and tail are a dozen and half a dozen small kernels whose arguments depend only on the launch's
  // shape and buffers -- the same proof after proof for a given key -- so each is captured once as a hipGraph and replayed
  // with a single hipGraphLaunch (BASELINE configs[4]: "hipGraph-captured rounds"); the accumulate kernel stays a plain
  // launch between them so that the profiling events and the side-stream hand-over (msm_tail_event) bracket it.
  auto front = [&]() -> int {
  {
    // counts and the sort's cursors are adjacent (and 256-byte aligned): cleared by the same kernel
    const size_t quads = (L.zero_end - L.off_counts) / sizeof(uint4);
    const uint32_t blocks = (uint32_t)std::min<size_t>((quads + 255) / 256, 1024);
    msm_set_ptrs_kernel<<<std::max(blocks, 1u), 256, 0, s>>>(sp, bp, stv, lnv, srcv, (const void**)d_scalars, batch, (uint4*)counts, quads);
  }
  if (L.part_sort) {
    const uint32_t P = batch * L.npart;
    const uint64_t E = (uint64_t)batch * W * n;
    uint32_t* part_pay = (uint32_t*)(ws + L.off_ranks);
    // two passes: pay (4 E bytes) | low, a byte per entry.  With the refinement pass: pay | pay2 | low1 (u16) | low2 (u8),
    // the 32-bit arrays first so that both stay 4-byte aligned whatever the parity of E
    void* part_low = part_pay + (L.mid ? 2 * E : E);
    uint32_t* poff = (uint32_t*)(ws + L.off_poff);
    uint32_t* cursor = (uint32_t*)(ws + L.off_cursor);
    uint32_t* psize = (uint32_t*)(ws + L.off_psize);
    uint32_t* pcursor = psize + P;
    // scalars per lane of the partition histogram: few in a small launch (more workgroups in flight: the kernel is
    // latency-bound there), DIGITS_LDS_PER_LANE in a large one (fewer device atomics: 128 per workgroup)
    const uint32_t per_lane = (uint64_t)n * batch <= (1u << 21) ? 2u : DIGITS_LDS_PER_LANE;
    const uint32_t hist_chunk = DIGITS_LDS_THREADS * per_lane;
    const dim3 hgrid((n + hist_chunk - 1) / hist_chunk, batch), sgrid((n + PSC_THREADS - 1) / PSC_THREADS, batch);
    // window width and count as compile-time constants for the table widths in use (see msm_part_hist_kernel)
#define CQ_PART_PASS1(CW, NW, LOWT)                                                                                               \
  do {                                                                                                                            \
    msm_part_hist_kernel<CW, NW><<<hgrid, DIGITS_LDS_THREADS, 0, s>>>(d_scalars, d_lens, c, W, L.npart, per_lane, psize);          \
    msm_part_scan_kernel<<<1, 1024, 0, s>>>(psize, P, poff, d_src, batch, L.npart, s1, s1_dev);                                  \
    msm_part_scatter_kernel<CW, NW, LOWT><<<sgrid, PSC_THREADS, 0, s>>>(d_scalars, d_lens, c, W, L.npart, poff, pcursor, part_pay, \
                                                                        (LOWT*)part_low);                                         \
  } while (0)
    if (c == 15 && W == 17) CQ_PART_PASS1(15, 17, uint8_t);
    else if (c == 16 && W == 16) CQ_PART_PASS1(16, 16, uint8_t);
    else if (c == 17 && W == 15) CQ_PART_PASS1(17, 15, uint8_t);
    else if (c == 18 && W == 15) CQ_PART_PASS1(18, 15, uint16_t);
    else if (c == 19 && W == 14) CQ_PART_PASS1(19, 14, uint16_t);
    else if (c == 20 && W == 13) CQ_PART_PASS1(20, 13, uint16_t);
    else if (L.mid) CQ_PART_PASS1(0, 0, uint16_t);
    else CQ_PART_PASS1(0, 0, uint8_t);
#undef CQ_PART_PASS1
    // workgroups per partition in the bucket passes: PART_SPLIT when partitions hold many tiles, one when they hold one
    auto split_for = [&](uint32_t parts) {
      const uint64_t tiles = (uint64_t)W * n * batch / ((uint64_t)parts * PART_TILE) + 1;
      return (uint32_t)std::min<uint64_t>(PART_SPLIT, tiles);
    };
    const uint32_t* fpay = part_pay;
    const uint8_t* flow = (const uint8_t*)part_low;
    const uint32_t* fpoff = poff;
    uint32_t F = P;
    if (L.mid) {  // 128 partitions of 2^(c-8) buckets -> partitions of 128 buckets
      const uint16_t* low1 = (const uint16_t*)part_low;
      uint32_t* pay2 = part_pay + E;
      uint8_t* low2 = (uint8_t*)(low1 + E);
      uint32_t* poff2 = (uint32_t*)(ws + L.off_poff2);
      uint32_t* psize2 = (uint32_t*)(ws + L.off_psize2);
      F = batch * L.nfinal;
      const uint32_t bins_log = L.shift1 - PART_BITS;
      msm_refine_count_kernel<<<dim3(P, PART_SPLIT), PART_THREADS, 0, s>>>(low1, poff, bins_log, psize2);
      msm_part_scan_kernel<<<1, 1024, 0, s>>>(psize2, F, poff2);
      msm_refine_place_kernel<<<dim3(P, PART_SPLIT), PART_THREADS, 0, s>>>(part_pay, low1, poff, bins_log, poff2, psize2 + F, pay2, low2);
      fpay = pay2;
      flow = low2;
      fpoff = poff2;
    }
    const uint32_t split = split_for(F);
    const bool wide = !L.mid && L.shift1 == PART_BITS_WIDE;  // 256 buckets per partition
    if (wide) msm_bucket_count_kernel<PART_BITS_WIDE><<<dim3(F, split), PART_THREADS, 0, s>>>(flow, fpoff, counts);
    else msm_bucket_count_kernel<PART_BITS><<<dim3(F, split), PART_THREADS, 0, s>>>(flow, fpoff, counts);
    if (any_alias) msm_alias_counts_kernel<<<dim3((L.B + 255) / 256, batch), 256, 0, s>>>(counts, d_src, L.B);
    msm_scan_reduce_kernel<<<L.nblk, 256, 0, s>>>(counts, Bt, L.nseq, s1, s1_dev, blocksums, ticket);
    msm_scan_apply_kernel<<<L.nblk, 256, 0, s>>>(counts, Bt, L.nseq, s1, s1_dev, blocksums, off, tk);
    if (wide) msm_bucket_place_kernel<PART_BITS_WIDE><<<dim3(F, split), PART_THREADS, 0, s>>>(fpay, flow, fpoff, off0, cursor, sorted);
    else msm_bucket_place_kernel<PART_BITS><<<dim3(F, split), PART_THREADS, 0, s>>>(fpay, flow, fpoff, off0, cursor, sorted);
  } else {
    msm_digits_kernel<<<dim3((n + 255) / 256, batch), 256, 0, s>>>(d_scalars, d_lens, n, c, W, pre ? 1u : 0u, ranks, counts);
    msm_scan_reduce_kernel<<<L.nblk, 256, 0, s>>>(counts, Bt, L.nseq, s1, nullptr, blocksums, ticket);
    msm_scan_apply_kernel<<<L.nblk, 256, 0, s>>>(counts, Bt, L.nseq, s1, nullptr, blocksums, off, tk);
    msm_scatter_kernel<<<dim3((n + 255) / 256, batch), 256, 0, s>>>(d_scalars, d_lens, n, c, W, pre ? 1u : 0u, ranks, off0, sorted);
  }
  return 0;
  };
  // everything the captured kernels' arguments are made of
  std::vector<uint64_t> sig{(uint64_t)(uintptr_t)workspace, (uint64_t)(uintptr_t)window_sums_dev, n, c, batch, pre ? 1u : 0u, s1};
  for (uint32_t i = 0; i < batch; i++) {
    sig.push_back((uint64_t)(uintptr_t)scalars_host_ptrs[i]);
    sig.push_back((uint64_t)(uintptr_t)bases_host_ptrs[i]);
    sig.push_back((uint64_t)lens[i]);
    sig.push_back(pre ? (uint64_t)table_strides[i] : 0);
  }
  if (ctx->run_graph(sig, 0, front) != 0) return -1;
  if (ctx->prof_on && ctx->prof_entries && ctx->prof_entries_n < cq_ctx::PROF_COUNTERS)
    (void)hipMemcpyAsync(ctx->prof_entries + ctx->prof_entries_n++, off0 + Bt, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  // Large launches release the side stream BEFORE the accumulate kernel: its lowest-priority workgroups are then placed
  // whenever the accumulate kernel has none left to place, i.e. they fill its drain (~0.3 ms of half-empty GPU at k = 22).
  // Measured (profiles/r03_msm_tail_early_ab.txt): k = 22 -0.6 .. -0.8 ms, k = 20 -0.05 .. -0.5; at k = 18 the transforms
  // that get in first delay the accumulate kernel's start by more than its shorter drain gives back (+0.25 ms on one box,
  // -0.07 on another), hence the size floor.  CQ_MSM_TAIL_EARLY=0/1 pins it.
  static const int tail_env = getenv("CQ_MSM_TAIL_EARLY") ? atoi(getenv("CQ_MSM_TAIL_EARLY")) : -1;
  const bool tail_early = tail_env >= 0 ? tail_env != 0 : n >= ((size_t)1 << 20);
  if (tail_early && ctx->msm_tail_event) {
    (void)hipEventRecord(ctx->msm_tail_event, s);
    ctx->msm_tail_seq++;
  }
  hipEvent_t pe = ctx->prof_begin(CQ_PROF_MSM_ACCUMULATE);
  msm_accumulate_kernel<<<(uint32_t)((L.tmax[0] + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS), MSM_ACC_THREADS, 0, s>>>(
      d_bases, L.B, pre ? 1u : 0u, d_strides, d_src, sorted, counts, off0, tk, off + (size_t)(Bt + 1), Bt, part[0], buckets);
  ctx->prof_end(pe);
  if (ctx->msm_tail_event && !tail_early) {  // from here on the launch is latency-bound: side-stream work may start (ctx.hpp)
    (void)hipEventRecord(ctx->msm_tail_event, s);
    ctx->msm_tail_seq++;
  }
  auto tail = [&]() -> int {
  for (uint32_t k = 1; k < L.levels; k++) {
    const uint32_t* t_prev = tk + (size_t)(k - 1) * Bt;
    const uint32_t* off_prev = off + (size_t)k * (Bt + 1);
    const uint32_t* t_cur = tk + (size_t)k * Bt;
    const uint32_t* off_cur = off + (size_t)(k + 1) * (Bt + 1);
    // ping-pong: level k+1 reads part[(k-1)&1], writes part[k&1]
    const uint32_t q = Bt <= (1u << 18) ? 4u : Bt <= (1u << 19) ? 2u : 1u;  // lanes per bucket, see the kernel
    const uint32_t cs_blocks = (uint32_t)(((uint64_t)Bt * q + 255) / 256);
    // (CQ_MSM_COMBINE_WAVE_BLOCKS: tests shrink the cap so that small launches stride too)
    static const uint32_t wave_cap = getenv("CQ_MSM_COMBINE_WAVE_BLOCKS") ? (uint32_t)std::max(1, atoi(getenv("CQ_MSM_COMBINE_WAVE_BLOCKS"))) : MSM_COMBINE_WAVE_BLOCKS;
    const uint32_t wv_blocks = (uint32_t)std::min<uint64_t>((L.tmax[k] + 3) / 4, wave_cap);
    if (q == 4) msm_combine_level_kernel<4><<<cs_blocks + wv_blocks, 256, 0, s>>>(part[(k - 1) & 1], t_prev, off_prev, t_cur, off_cur, Bt, cs_blocks, part[k & 1], buckets);
    else if (q == 2) msm_combine_level_kernel<2><<<cs_blocks + wv_blocks, 256, 0, s>>>(part[(k - 1) & 1], t_prev, off_prev, t_cur, off_cur, Bt, cs_blocks, part[k & 1], buckets);
    else msm_combine_level_kernel<1><<<cs_blocks + wv_blocks, 256, 0, s>>>(part[(k - 1) & 1], t_prev, off_prev, t_cur, off_cur, Bt, cs_blocks, part[k & 1], buckets);
  }
  const uint32_t sets = batch * L.Wb;
  // (few sets: the launch waits for chains of dependent additions, and four lanes per addition shorten them; many sets: the
  // SIMDs are busy, and one lane per addition is half the instructions.  CQ_MSM_QUAD=0/1 pins the choice.)
  static const int quad_env = getenv("CQ_MSM_QUAD") ? atoi(getenv("CQ_MSM_QUAD")) : -1;
  static const uint32_t seg32_sets = getenv("CQ_MSM_ROWCOL_SEG32_SETS") ? (uint32_t)atoi(getenv("CQ_MSM_ROWCOL_SEG32_SETS")) : 12u;
  static const uint32_t quad_rowcol_sets = getenv("CQ_MSM_QUAD_ROWCOL_SETS") ? (uint32_t)atoi(getenv("CQ_MSM_QUAD_ROWCOL_SETS")) : 4u;
  if (quad_env != 0 && sets <= quad_rowcol_sets)
    msm_rowcol_quad_kernel<<<dim3(L.rows + L.cols, sets), 256, 0, s>>>(buckets, tk, M, L.rows, L.cols, pairs);
  else if (sets <= 4)
    msm_rowcol_kernel<64><<<dim3(L.rows + L.cols, sets), 64, 0, s>>>(buckets, tk, M, L.rows, L.cols, pairs);
  else if (sets <= seg32_sets)
    msm_rowcol_kernel<32><<<dim3((L.rows + L.cols + 1) / 2, sets), 64, 0, s>>>(buckets, tk, M, L.rows, L.cols, pairs);
  else
    msm_rowcol_kernel<16><<<dim3((L.rows + L.cols + 3) / 4, sets), 64, 0, s>>>(buckets, tk, M, L.rows, L.cols, pairs);
  // (measured at k = 18: 21 sets -- 1 260 four-wave blocks -- 88 -> 54 us; from ~2 waves per SIMD on the plain kernel's
  // half-as-many instructions win)
  static const uint32_t quad_weighted_waves = getenv("CQ_MSM_QUAD_WEIGHTED_WAVES") ? (uint32_t)atoi(getenv("CQ_MSM_QUAD_WEIGHTED_WAVES")) : 2048u;
  const bool quad = quad_env >= 0 ? quad_env != 0 : sets * MSM_SET_POINTS * 4 <= quad_weighted_waves;
  if (quad) msm_weighted_quad_kernel<<<MSM_SET_POINTS * sets, 256, 0, s>>>(pairs, L.rows, L.cols, window_sums_dev);
  else msm_weighted_kernel<<<MSM_SET_POINTS * sets, 64, 0, s>>>(pairs, L.rows, L.cols, window_sums_dev);
  return 0;
  };
  if (ctx->run_graph(sig, 1, tail) != 0) return -1;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// cols * V + U from the bit-plane sums of msm_weighted_kernel: V = sum_t 2^t R_t, U = sum_t 2^t C_t + C_total.
// With cols = 2^7 (every set of 2^14 buckets) the two Horner chains are one: 2^7 V + sum_t 2^t C_t is the Horner over
// R_6 .. R_0, C_6 .. C_0 -- 13 doublings and 13 additions instead of 19 and 12.
G1Jac msm_set_value(const G1Jac* planes, uint32_t cols) {
  G1Jac v = planes[6];
  for (int t = 5; t >= 0; t--) v = jac_add(jac_dbl(v), planes[t]);
  if (cols == 128) {
    for (int t = 6; t >= 0; t--) v = jac_add(jac_dbl(v), planes[7 + t]);
    return jac_add(v, planes[14]);
  }
  G1Jac u = planes[13];
  for (int t = 5; t >= 0; t--) u = jac_add(jac_dbl(u), planes[7 + t]);
  for (uint32_t l = cols; l > 1; l >>= 1) v = jac_dbl(v);
  return jac_add(jac_add(v, u), planes[14]);
}

// sum_s ( value(set s) + s * M * total(set s) ).  The value of a set is linear in its fifteen points, so the sets' points
// are added plane by plane first and folded ONCE (15 (Wb - 1) additions + 27 operations instead of 27 Wb: the host's
// share per 17-bit-window MSM goes from ~155 group operations to ~95 -- at k >= 20 the two largest host gaps of a proof).
G1Jac msm_fold_sets(const G1Jac* pairs, uint32_t Wb, uint32_t M, uint32_t cols) {
  if (Wb == 1) return msm_set_value(pairs, cols);
  G1Jac planes[MSM_SET_POINTS];
  for (uint32_t q = 0; q < MSM_SET_POINTS; q++) {
    planes[q] = pairs[q];
    for (uint32_t s = 1; s < Wb; s++) planes[q] = jac_add(planes[q], pairs[(size_t)MSM_SET_POINTS * s + q]);
  }
  G1Jac acc = msm_set_value(planes, cols);
  // sum_s s * T_s by suffix sums (T_s: the set's plain total, its last point), then "* M" by doublings
  G1Jac run = G1Jac::identity(), weighted = G1Jac::identity();
  for (uint32_t s = Wb - 1; s >= 1; s--) {
    run = jac_add(run, pairs[(size_t)MSM_SET_POINTS * s + MSM_SET_POINTS - 1]);
    weighted = jac_add(weighted, run);
  }
  for (uint32_t l = M; l > 1; l >>= 1) weighted = jac_dbl(weighted);
  return jac_add(acc, weighted);
}

G1Jac msm_fold_windows(const G1Jac* pairs, uint32_t W, uint32_t c, uint32_t cols) {
  G1Jac acc = G1Jac::identity();
  for (int w = (int)W - 1; w >= 0; w--) {
    for (uint32_t j = 0; j < c; j++) acc = jac_dbl(acc);
    acc = jac_add(acc, msm_set_value(pairs + (size_t)MSM_SET_POINTS * w, cols));
  }
  return acc;
}

}  // namespace cq
