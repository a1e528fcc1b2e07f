// `permute_expression_pair` of the legacy (halo2) lookup argument on the device -- halo2_proofs/src/plonk/lookup/prover.rs:400-502.
//
// The reference sorts the compressed input expression, walks it against a BTreeMap of the table's values and hands the
// table values that no first occurrence claimed to the repeated rows.  Restated as data-parallel steps over the
// canonical 256-bit values (the order the reference's `Ord` for field elements uses):
//   1. sort the input values A and the table values T (bitonic network, see below);
//   2. first[i] = A[i] != A[i-1];  every first row claims the first element of its value's run in T (binary search;
//      a value that T does not hold is Error::ConstraintSystemFailure);
//   3. the unclaimed elements of T, in ascending order, go to the repeated rows in DESCENDING row order (the
//      reference pops its `repeated_input_rows` from the back): two exclusive scans and one gather.
// The sort is comparison-based on whole keys: a bitonic network over 2^k slots (the unusable tail padded with all-ones
// keys, larger than any canonical field element) -- the steps with a compare distance below 2048 elements run inside
// LDS (one launch per merge phase), the others are one streaming launch each over an array that fits the L2 / MALL.
// 36 launches for 2^18 rows, ~0.4 ms, where the host std::sort took ~50 ms; equal keys are identical, so the network's
// lack of stability does not matter.
#include "plonk.hpp"
#include "ctx.hpp"

namespace cq {

namespace {

constexpr uint32_t SORT_THREADS = 256, SORT_CHUNK_LOG = 11, SORT_CHUNK = 1u << SORT_CHUNK_LOG;  // elements per LDS tile

struct Key {
  uint64_t l[4];
};
static __device__ __forceinline__ bool key_less(const Key& a, const Key& b) {
  if (a.l[3] != b.l[3]) return a.l[3] < b.l[3];
  if (a.l[2] != b.l[2]) return a.l[2] < b.l[2];
  if (a.l[1] != b.l[1]) return a.l[1] < b.l[1];
  return a.l[0] < b.l[0];
}
static __device__ __forceinline__ bool key_eq(const Key& a, const Key& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}
static __device__ __forceinline__ Key g_load(const uint64_t* p, size_t i) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p + 4 * i);
  const ulonglong2 a = q[0], b = q[1];
  return {{a.x, a.y, b.x, b.y}};
}
static __device__ __forceinline__ void g_store(uint64_t* p, size_t i, const Key& k) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(p + 4 * i);
  q[0] = make_ulonglong2(k.l[0], k.l[1]);
  q[1] = make_ulonglong2(k.l[2], k.l[3]);
}

// pads [len, n) of both arrays (blockIdx.y) with all-ones keys
__global__ void sort_pad_kernel(uint64_t* a, uint64_t* b, uint32_t len, uint32_t n) {
  const uint32_t i = len + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Key inf = {{~0ull, ~0ull, ~0ull, ~0ull}};
  g_store(blockIdx.y ? b : a, i, inf);
}

// one step of the network with compare distance j >= SORT_CHUNK inside merge phase kk (both powers of two)
__global__ __launch_bounds__(SORT_THREADS) void sort_global_step_kernel(uint64_t* a, uint64_t* b, uint32_t n, uint32_t kk, uint32_t j) {
  uint64_t* arr = blockIdx.y ? b : a;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
  const Key x = g_load(arr, i), y = g_load(arr, l);
  const bool up = (i & kk) == 0;
  if (key_less(y, x) == up) {  // out of order for this direction (equal keys: no swap either way)
    if (!key_eq(x, y)) {
      g_store(arr, i, y);
      g_store(arr, l, x);
    }
  }
}

// all steps with distance < SORT_CHUNK of the merge phases kk = k_first .. k_last (k_first == k_last beyond the chunk
// size; 2 .. SORT_CHUNK for the initial sort of every chunk), inside LDS.  Limb planes: conflict-free 8-byte accesses.
__global__ __launch_bounds__(SORT_THREADS) void sort_lds_kernel(uint64_t* a, uint64_t* b, uint32_t n, uint32_t k_first, uint32_t k_last) {
  __shared__ uint64_t sm[4 * SORT_CHUNK];
  uint64_t* arr = blockIdx.y ? b : a;
  const uint32_t base = blockIdx.x * SORT_CHUNK;
  for (uint32_t e = threadIdx.x; e < SORT_CHUNK; e += SORT_THREADS) {
    const Key x = g_load(arr, base + e);
#pragma unroll
    for (int q = 0; q < 4; q++) sm[q * SORT_CHUNK + e] = x.l[q];
  }
  __syncthreads();
  for (uint32_t kk = k_first; kk <= k_last && kk != 0; kk <<= 1) {
    for (uint32_t j = min(kk >> 1, SORT_CHUNK >> 1); j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < SORT_CHUNK / 2; t += SORT_THREADS) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        Key x, y;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          x.l[q] = sm[q * SORT_CHUNK + i];
          y.l[q] = sm[q * SORT_CHUNK + l];
        }
        const bool up = ((base + i) & kk) == 0;
        if (key_less(y, x) == up && !key_eq(x, y)) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            sm[q * SORT_CHUNK + i] = y.l[q];
            sm[q * SORT_CHUNK + l] = x.l[q];
          }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t e = threadIdx.x; e < SORT_CHUNK; e += SORT_THREADS) {
    Key x;
#pragma unroll
    for (int q = 0; q < 4; q++) x.l[q] = sm[q * SORT_CHUNK + e];
    g_store(arr, base + e, x);
  }
}

// small arrays (n < SORT_CHUNK): the whole network by one block on global memory
__global__ __launch_bounds__(SORT_THREADS) void sort_small_kernel(uint64_t* a, uint64_t* b, uint32_t n) {
  uint64_t* arr = blockIdx.y ? b : a;
  for (uint32_t kk = 2; kk <= n; kk <<= 1)
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < n / 2; t += SORT_THREADS) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const Key x = g_load(arr, i), y = g_load(arr, l);
        const bool up = (i & kk) == 0;
        if (key_less(y, x) == up && !key_eq(x, y)) {
          g_store(arr, i, y);
          g_store(arr, l, x);
        }
      }
      __syncthreads();
    }
}

// step 2: first-occurrence flags of the sorted input, claims on the sorted table.  flags[i] = 1 for a REPEATED row;
// used[] must be zero on entry; *bad counts input values the table does not hold.
__global__ void lk_claim_kernel(const uint64_t* __restrict__ A, const uint64_t* __restrict__ T, uint32_t u, uint32_t* __restrict__ repeated,
                                uint32_t* __restrict__ used, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= u) return;
  const Key x = g_load(A, i);
  const bool first = i == 0 || !key_eq(x, g_load(A, i - 1));
  repeated[i] = first ? 0u : 1u;
  if (!first) return;
  uint32_t lo = 0, hi = u;  // lower bound of x in T[0, u)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (key_less(g_load(T, mid), x)) lo = mid + 1; else hi = mid;
  }
  if (lo < u && key_eq(g_load(T, lo), x)) used[lo] = 1u;  // one claim per distinct value: no race
  else atomicAdd(bad, 1u);
}

// exclusive scans of two 0/1 arrays at once (blockIdx.y): `repeated` as is, `used` NEGATED (unclaimed table elements).
// Three kernels: per-tile sums, one-block spine, apply.
constexpr uint32_t SCAN_PER = 8, SCAN_TILE_ELEMS = SORT_THREADS * SCAN_PER;
static __device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* sh, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if ((int)lane >= off) x += y;
  }
  __syncthreads();
  if (lane == 63) sh[wid] = x;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
  for (uint32_t k = 0; k < SORT_THREADS / 64; k++) {
    if (k < wid) wbase += sh[k];
    tot += sh[k];
  }
  total = tot;
  return wbase + x - v;
}
static __device__ __forceinline__ uint32_t flag_of(const uint32_t* repeated, const uint32_t* used, uint32_t which, uint32_t i, uint32_t u) {
  if (i >= u) return 0;
  return which ? 1u - used[i] : repeated[i];
}
__global__ __launch_bounds__(SORT_THREADS) void lk_scan_reduce_kernel(const uint32_t* repeated, const uint32_t* used, uint32_t u, uint32_t* sums) {
  __shared__ uint32_t sh[SORT_THREADS / 64];
  uint32_t s = 0;
  for (uint32_t q = 0; q < SCAN_PER; q++) s += flag_of(repeated, used, blockIdx.y, blockIdx.x * SCAN_TILE_ELEMS + threadIdx.x * SCAN_PER + q, u);
  uint32_t tot;
  block_scan(s, sh, tot);
  if (threadIdx.x == 0) sums[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void lk_scan_spine_kernel(uint32_t* sums, uint32_t nblk, uint32_t* totals) {
  __shared__ uint32_t part[1024];
  for (uint32_t which = 0; which < 2; which++) {
    uint32_t* a = sums + which * nblk;
    const uint32_t per = (nblk + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    uint32_t s = 0;
    for (uint32_t j = lo; j < hi; j++) s += a[j];
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
      const uint32_t v = a[j];
      a[j] = run;
      run += v;
    }
    if (threadIdx.x == 1023) totals[which] = part[1023];
    __syncthreads();
  }
}
// rank[which][i] = exclusive prefix; the unclaimed table elements are compacted on the way (which == 1)
__global__ __launch_bounds__(SORT_THREADS) void lk_scan_apply_kernel(const uint32_t* repeated, const uint32_t* used, uint32_t u, const uint32_t* sums,
                                                                       const uint64_t* __restrict__ T, uint32_t* __restrict__ rep_rank,
                                                                       uint64_t* __restrict__ leftover) {
  __shared__ uint32_t sh[SORT_THREADS / 64];
  const uint32_t which = blockIdx.y;
  const uint32_t i0 = blockIdx.x * SCAN_TILE_ELEMS + threadIdx.x * SCAN_PER;
  uint32_t f[SCAN_PER], s = 0;
  for (uint32_t q = 0; q < SCAN_PER; q++) {
    f[q] = flag_of(repeated, used, which, i0 + q, u);
    s += f[q];
  }
  uint32_t tot;
  uint32_t run = block_scan(s, sh, tot) + sums[which * gridDim.x + blockIdx.x];
  for (uint32_t q = 0; q < SCAN_PER; q++) {
    const uint32_t i = i0 + q;
    if (i < u) {
      if (which == 0) rep_rank[i] = run;
      else if (f[q]) g_store(leftover, run, g_load(T, i));
    }
    run += f[q];
  }
}
// step 3: S'[i] = A'[i] on first rows; the q-th repeated row (ascending) takes leftover[r - 1 - q]
__global__ void lk_assign_kernel(const uint64_t* __restrict__ A, const uint32_t* __restrict__ repeated, const uint32_t* __restrict__ rep_rank,
                                 const uint64_t* __restrict__ leftover, const uint32_t* __restrict__ totals, uint32_t u,
                                 uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= u) return;
  const uint32_t r = totals[0];
  Key v;
  if (!repeated[i]) v = g_load(A, i);
  else if (totals[1] == r) v = g_load(leftover, r - 1 - rep_rank[i]);
  else v = {{0, 0, 0, 0}};  // inconsistent counts: reported through `bad` by the caller
  g_store(out, i, v);
}

}  // namespace

size_t lookup_permute_scratch_bytes(uint32_t k) {
  const size_t n = (size_t)1 << k;
  const size_t nblk = (n + SCAN_TILE_ELEMS - 1) / SCAN_TILE_ELEMS;
  return n * 32 /* leftover */ + 3 * n * sizeof(uint32_t) /* repeated, used, rep_rank */ + 2 * nblk * sizeof(uint32_t) + 256;
}

// in_canon / tab_canon: arrays of 2^k slots of canonical values, the first u valid; both are sorted in place (in_canon then
// IS the permuted input); out_tab receives the u permuted table values.  status (device, 3 words): [0] input values the
// table does not hold, [1] repeated rows, [2] unclaimed table elements -- a valid pair has [0] == 0 and [1] == [2].
int lookup_permute_dev(cq_ctx* c, uint64_t* in_canon, uint64_t* tab_canon, uint32_t u, uint32_t k, uint64_t* out_tab, void* scratch,
                       uint32_t* status_dev) {
  hipStream_t s = c->stream;
  const uint32_t n = 1u << k;
  if (u > n) return c->fail(CQ_ERR_ARG, "lookup_permute: u > n");
  if (n > u) sort_pad_kernel<<<dim3((n - u + 255) / 256, 2), 256, 0, s>>>(in_canon, tab_canon, u, n);
  if (n < SORT_CHUNK) {
    if (n >= 2) sort_small_kernel<<<dim3(1, 2), SORT_THREADS, 0, s>>>(in_canon, tab_canon, n);
  } else {
    const dim3 lgrid(n / SORT_CHUNK, 2), ggrid((n / 2 + SORT_THREADS - 1) / SORT_THREADS, 2);
    sort_lds_kernel<<<lgrid, SORT_THREADS, 0, s>>>(in_canon, tab_canon, n, 2, SORT_CHUNK);
    for (uint32_t kk = SORT_CHUNK << 1; kk <= n && kk != 0; kk <<= 1) {
      for (uint32_t j = kk >> 1; j >= SORT_CHUNK; j >>= 1) sort_global_step_kernel<<<ggrid, SORT_THREADS, 0, s>>>(in_canon, tab_canon, n, kk, j);
      sort_lds_kernel<<<lgrid, SORT_THREADS, 0, s>>>(in_canon, tab_canon, n, kk, kk);
    }
  }
  char* w = (char*)scratch;
  uint64_t* leftover = (uint64_t*)w;
  uint32_t* repeated = (uint32_t*)(w + (size_t)n * 32);
  uint32_t* used = repeated + n;
  uint32_t* rep_rank = used + n;
  uint32_t* sums = rep_rank + n;
  const uint32_t nblk = (u + SCAN_TILE_ELEMS - 1) / SCAN_TILE_ELEMS;
  CQ_HIP(c, hipMemsetAsync(used, 0, (size_t)n * sizeof(uint32_t), s));
  CQ_HIP(c, hipMemsetAsync(status_dev, 0, 3 * sizeof(uint32_t), s));
  if (u) {
    lk_claim_kernel<<<(u + 255) / 256, 256, 0, s>>>(in_canon, tab_canon, u, repeated, used, status_dev);
    lk_scan_reduce_kernel<<<dim3(nblk, 2), SORT_THREADS, 0, s>>>(repeated, used, u, sums);
    lk_scan_spine_kernel<<<1, 1024, 0, s>>>(sums, nblk, status_dev + 1);
    lk_scan_apply_kernel<<<dim3(nblk, 2), SORT_THREADS, 0, s>>>(repeated, used, u, sums, tab_canon, rep_rank, leftover);
    lk_assign_kernel<<<(u + 255) / 256, 256, 0, s>>>(in_canon, repeated, rep_rank, leftover, status_dev + 1, u, out_tab);
  }
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "lookup_permute launch failed");
}

}  // namespace cq
