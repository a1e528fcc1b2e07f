// CQ ("cached quotients") static-lookup argument, device side: table objects and the per-row /
// per-table-entry kernels of prover rounds 1 and 2 (halo2_proofs/src/plonk/static_lookup.rs,
// plonk/static_lookup/prover.rs).  The reference walks BTreeMaps row by row on one thread and adds
// commitments with serial double-and-add; here the value->index map is an open-addressing hash table
// in HBM, multiplicities are a dense atomic histogram, and every commitment is a (dense) MSM.
// Also: the SHA-256 word -> (dense limb, spread limb) witness fill (sha/src/tables.rs).
#include "cq.hpp"
#include "ctx.hpp"

namespace cq {

static __device__ __forceinline__ Fr ld(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
static __device__ __forceinline__ void st(Fr* p, const Fr& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v.l[0], r.v.l[1], r.v.l[2], r.v.l[3]);
  q[1] = make_uint4(r.v.l[4], r.v.l[5], r.v.l[6], r.v.l[7]);
}

// ---- value -> index hash table (replaces BTreeMap<Fr, usize>, static_lookup.rs:72,82-85) ------------
static __device__ __forceinline__ uint32_t hash_fr(const Fr& v) {
  uint32_t h = 0x9e3779b9u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    h ^= v.v.l[i];
    h *= 0x85ebca6bu;
    h ^= h >> 15;
  }
  return h;
}

constexpr uint32_t EMPTY = 0xffffffffu;

// slot array of `nslots` (power of two) table indices; values are compared in full (256 bits)
__global__ void table_insert_kernel(const Fr* __restrict__ values, uint32_t N, uint32_t* __restrict__ slots, uint32_t nslots,
                                    uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Fr v = ld(values + i);
  uint32_t s = hash_fr(v) & (nslots - 1);
  for (uint32_t probe = 0; probe < nslots; probe++) {
    const uint32_t prev = atomicCAS(&slots[s], EMPTY, i);
    if (prev == EMPTY) return;
    if (prev != i && ld(values + prev) == v) {
      atomicExch(err, 1u);  // duplicate table value (static_lookup.rs:84-85 assert)
      return;
    }
    s = (s + 1) & (nslots - 1);
  }
  atomicExch(err, 2u);
}

static __device__ __forceinline__ uint32_t table_find(const Fr* __restrict__ values, const uint32_t* __restrict__ slots,
                                                      uint32_t nslots, const Fr& v) {
  uint32_t s = hash_fr(v) & (nslots - 1);
  for (uint32_t probe = 0; probe < nslots; probe++) {
    const uint32_t idx = slots[s];
    if (idx == EMPTY) return EMPTY;
    if (ld(values + idx) == v) return idx;
    s = (s + 1) & (nslots - 1);
  }
  return EMPTY;
}

// ---- round 1 (static_lookup/prover.rs:132-161): row -> table index, multiplicities --------------------
// Same-address device atomics serialise in L2 (~15 ns each): the padded rows of a sparse witness all hit one table
// entry, and one atomic per wave still meant ~3000 of them on that entry (45 of the kernel's 52 us).  Every lane
// therefore looks up R1_ROWS rows (strided, so loads stay coalesced), folds equal neighbours into a running (index, count) pair, and what
// is left at the end is aggregated across the wave before it touches L2.
constexpr uint32_t R1_ROWS = 4;
__global__ __launch_bounds__(256) void cq_round1_kernel(CqRound1Batch batch, uint32_t u, uint32_t* __restrict__ err) {
  const CqRound1Args& a = batch.a[blockIdx.y];
  uint32_t* __restrict__ m_counts = batch.m_counts[blockIdx.y];
  uint32_t found[R1_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < R1_ROWS; r++) {  // independent probe chains, in flight together
    const uint32_t row = (blockIdx.x * R1_ROWS + r) * blockDim.x + threadIdx.x;
    uint32_t idx = EMPTY;
    bool bad = row >= u;
    for (uint32_t j = 0; !bad && j < a.width; j++) {
      const Fr v = ld(a.cols[j] + row);
      const uint32_t ix = table_find(a.values[j], a.slots[j], a.nslots[j], v);
      if (ix == EMPTY) {
        atomicExch(err, 1u);  // "{:?} not in table" (:141)
        bad = true;
      } else if (j && ix != idx) {
        atomicExch(err, 2u);  // "Vector lookup must be on the same table row" (:148)
        bad = true;
      }
      idx = ix;
    }
    found[r] = bad ? EMPTY : idx;
  }
  uint32_t held = EMPTY, held_cnt = 0;
#pragma unroll
  for (uint32_t r = 0; r < R1_ROWS; r++) {
    if (found[r] == EMPTY) continue;
    if (found[r] == held) {
      held_cnt++;
    } else {
      if (held_cnt) atomicAdd(&m_counts[held], held_cnt);
      held = found[r];
      held_cnt = 1;
    }
  }
  // aggregate equal indices within the wave (all lanes reach this point)
  bool pending = held_cnt != 0;
#pragma unroll 1
  for (int round = 0; round < 4 && __ballot(pending); round++) {
    const unsigned long long pend = __ballot(pending);
    const int leader = __ffsll((long long)pend) - 1;
    const uint32_t lidx = __shfl(held, leader, 64);
    const bool mine = pending && held == lidx;
    uint32_t c = mine ? held_cnt : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&m_counts[lidx], c);
    if (mine) pending = false;
  }
  if (pending) atomicAdd(&m_counts[held], held_cnt);
}

// ---- round 2 (static_lookup/prover.rs:245-257), dense over the table -----------------------------------
// den[i] = m[i] ? t[i] + beta : 0           (t = theta-compressed table values)
__global__ void cq_a_denominators_kernel(const Fr* __restrict__ t, const uint32_t* __restrict__ m, uint32_t N, Fr beta,
                                         Fr* __restrict__ den) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  st(den + i, m[i] ? ld(t + i) + beta : Fr::zero());
}
// a[i] = m[i] * den_inv[i]; scaled copies a[i]*theta^(w-1-j) for the q_a MSM over [qs_0 | qs_1 | ...]
__global__ void cq_a_values_kernel(const Fr* __restrict__ den_inv, const uint32_t* __restrict__ m, uint32_t N, CqThetaPowers tp,
                                   Fr* __restrict__ a, Fr* __restrict__ a_scaled /* width*N */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t mi = m[i];
  Fr ai = Fr::zero();
  if (mi) ai = Fr::from_u64(mi) * ld(den_inv + i);
  st(a + i, ai);
  for (uint32_t j = 0; j < tp.width; j++) st(a_scaled + (size_t)j * N + i, mi ? ai * tp.pow[j] : Fr::zero());
}
// ---- the same front work for several lookups per launch (cq.hpp: CqFoldBatch) -------------------------------------------
static __device__ __forceinline__ Fr fold_theta(const CqFoldBatch& b, uint32_t l, uint32_t i) {
  const uint32_t w = b.width[l];
  Fr acc = Fr::zero();
  for (uint32_t j = 0; j < w; j++) acc = acc + ld(b.src[l][j] + i) * b.theta_pow[w - 1 - j];
  return acc;
}
// f = sum_j theta^(w-1-j) e_j   (static_lookup/prover.rs:108-116)
__global__ __launch_bounds__(256) void cq_fold_inputs_kernel(CqFoldBatch b, uint32_t n) {
  CQ_CRITICAL_WAVES();
  const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !b.out[l]) return;
  st(b.out[l] + i, fold_theta(b, l, i));
}
// B_r = f_r + beta (r < u) / beta (:261-269) and den_i = m_i ? t_i + beta : 0 with the compressed table t (:224-240, 245-247)
__global__ __launch_bounds__(256) void cq_round2_prep_kernel(CqFoldBatch b, uint32_t n, uint32_t N, uint32_t u, Fr beta) {
  CQ_CRITICAL_WAVES();
  const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && b.f[l]) st(b.b[l] + i, i < u ? ld(b.f[l] + i) + beta : beta);
  if (i < N) st(b.den[l] + i, b.m[l][i] ? fold_theta(b, l, i) + beta : Fr::zero());
}
__global__ __launch_bounds__(256) void cq_a_values_batch_kernel(CqAValuesBatch b, uint32_t N) {
  CQ_CRITICAL_WAVES();
  const uint32_t l = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t mi = b.m[l][i], w = b.width[l];
  Fr ai = Fr::zero();
  if (mi) ai = Fr::from_u64(mi) * ld(b.den_inv[l] + i);
  st(b.a[l] + i, ai);
  for (uint32_t j = 0; j < w; j++) st(b.a_scaled[l] + (size_t)j * N + i, mi ? ai * b.theta_pow[w - 1 - j] : Fr::zero());
}
// multiplicities as field elements (for m_cm)
__global__ void cq_m_to_fr_kernel(const uint32_t* __restrict__ m, uint32_t N, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  st(out + i, m[i] ? Fr::from_u64(m[i]) : Fr::zero());
}

// ---- closed-form cached quotients for test tables: sc[i] = (T(s) - T(w^i)) / (s - w^i) * w^i / N -----------
__global__ void cq_qs_scalars_kernel(const Fr* __restrict__ values, uint32_t N, Fr ts, Fr s, Fr omega, Fr n_inv,
                                     Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Fr g = omega.pow_u64(i);
  st(out + i, (ts - ld(values + i)) * (s - g).inv() * g * n_inv);
}

// ---- SHA-256 witness fill: 32-bit words -> 12/10/10 limbs (LongLimbs, sha/src/tables.rs:70-75) as
//      (dense, spread) pairs, Montgomery-encoded, round-robin over `pairs` column pairs ---------------------
static __device__ __forceinline__ uint32_t spread_bits(uint32_t x) {
  // bit i of x -> bit 2i
  x &= 0xffffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

__global__ __launch_bounds__(256) void sha_witness_fill_kernel(const uint32_t* __restrict__ words, uint32_t nwords, uint32_t pairs,
                                                               uint32_t n, ShaCols cols) {
  // limb t (0..3*nwords) goes to pair (t % pairs), row (t / pairs)
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t rows = (3 * nwords + pairs - 1) / pairs;
  if (t >= rows * pairs) return;
  const uint32_t pr = t % pairs, row = t / pairs;
  if (row >= n) return;
  uint32_t limb = 0;
  if (t < 3 * nwords) {
    const uint32_t w = words[t / 3];
    const uint32_t which = t % 3;  // tables.rs:139-149: x = a >> 20, y = (a >> 10) & 0x3ff, z = a & 0x3ff
    limb = which == 0 ? (w >> 20) : which == 1 ? ((w >> 10) & 0x3ffu) : (w & 0x3ffu);
  }
  st(cols.p[2 * pr] + row, limb ? Fr::from_u64(limb) : Fr::zero());
  const uint32_t sp = spread_bits(limb);
  st(cols.p[2 * pr + 1] + row, sp ? Fr::from_u64(sp) : Fr::zero());
}

// dense / spread table columns of size N = 2^bits: values i and spread(i)
__global__ void sha_spread_table_kernel(uint32_t N, Fr* __restrict__ dense, Fr* __restrict__ spread) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  st(dense + i, i ? Fr::from_u64(i) : Fr::zero());
  const uint32_t sp = spread_bits(i);
  st(spread + i, sp ? Fr::from_u64(sp) : Fr::zero());
}

// =================================================================================================
static inline uint32_t blocks_for(uint32_t n) { return (n + 255) / 256; }

int cq_table_build_index(cq_ctx* c, const Fr* values, uint32_t N, uint32_t** slots_out, uint32_t* nslots_out) {
  uint32_t nslots = 4;
  while (nslots < 4 * N) nslots <<= 1;
  uint32_t* slots = nullptr;
  uint32_t* err = nullptr;
  if (hipMalloc(&slots, (size_t)nslots * 4) != hipSuccess) return c->fail(CQ_ERR_HIP, "hipMalloc(table slots)");
  if (hipMalloc(&err, 4) != hipSuccess) {
    hipFree(slots);
    return c->fail(CQ_ERR_HIP, "hipMalloc(err)");
  }
  hipMemsetAsync(slots, 0xff, (size_t)nslots * 4, c->stream);
  hipMemsetAsync(err, 0, 4, c->stream);
  table_insert_kernel<<<blocks_for(N), 256, 0, c->stream>>>(values, N, slots, nslots, err);
  uint32_t herr = 0;
  hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, c->stream);
  hipStreamSynchronize(c->stream);
  hipFree(err);
  if (herr) {
    hipFree(slots);
    return c->fail(CQ_ERR_ARG, herr == 1 ? "static table values are not unique" : "static table index overflow");
  }
  *slots_out = slots;
  *nslots_out = nslots;
  return CQ_OK;
}

int cq_round1(cq_ctx* c, const CqRound1Batch& b, uint32_t u, uint32_t* err_dev) {
  if (u && b.count) cq_round1_kernel<<<dim3((u + 256 * R1_ROWS - 1) / (256 * R1_ROWS), b.count), 256, 0, c->stream>>>(b, u, err_dev);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_round1 launch failed");
}

int cq_a_denominators(cq_ctx* c, const Fr* t, const uint32_t* m, uint32_t N, const Fr& beta, Fr* den) {
  cq_a_denominators_kernel<<<blocks_for(N), 256, 0, c->stream>>>(t, m, N, beta, den);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_a_den launch failed");
}
int cq_a_values(cq_ctx* c, const Fr* den_inv, const uint32_t* m, uint32_t N, const CqThetaPowers& tp, Fr* a, Fr* a_scaled) {
  cq_a_values_kernel<<<blocks_for(N), 256, 0, c->stream>>>(den_inv, m, N, tp, a, a_scaled);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_a_values launch failed");
}
int cq_fold_inputs(cq_ctx* c, const CqFoldBatch& b, uint32_t n) {
  if (!b.count || !n) return CQ_OK;
  cq_fold_inputs_kernel<<<dim3(blocks_for(n), b.count), 256, 0, c->stream>>>(b, n);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_fold_inputs launch failed");
}
int cq_round2_prep(cq_ctx* c, const CqFoldBatch& b, uint32_t n, uint32_t N, uint32_t u, const Fr& beta) {
  if (!b.count) return CQ_OK;
  cq_round2_prep_kernel<<<dim3(blocks_for(n > N ? n : N), b.count), 256, 0, c->stream>>>(b, n, N, u, beta);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_round2_prep launch failed");
}
int cq_a_values_batch(cq_ctx* c, const CqAValuesBatch& b, uint32_t N) {
  if (!b.count || !N) return CQ_OK;
  cq_a_values_batch_kernel<<<dim3(blocks_for(N), b.count), 256, 0, c->stream>>>(b, N);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_a_values_batch launch failed");
}
int cq_m_to_fr(cq_ctx* c, const uint32_t* m, uint32_t N, Fr* out) {
  cq_m_to_fr_kernel<<<blocks_for(N), 256, 0, c->stream>>>(m, N, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_m_to_fr launch failed");
}
int cq_qs_scalars(cq_ctx* c, const Fr* values, uint32_t N, const Fr& ts, const Fr& s, const Fr& omega, const Fr& n_inv, Fr* out) {
  cq_qs_scalars_kernel<<<blocks_for(N), 256, 0, c->stream>>>(values, N, ts, s, omega, n_inv, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "cq_qs_scalars launch failed");
}
int sha_witness_fill(cq_ctx* c, const uint32_t* words_dev, uint32_t nwords, uint32_t pairs, uint32_t n, const ShaCols& cols) {
  const uint32_t rows = (3 * nwords + pairs - 1) / pairs;
  if (rows > n) return c->fail(CQ_ERR_ARG, "sha witness does not fit the usable rows");
  if (rows) sha_witness_fill_kernel<<<blocks_for(rows * pairs), 256, 0, c->stream>>>(words_dev, nwords, pairs, n, cols);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "sha_witness_fill launch failed");
}
int sha_spread_table(cq_ctx* c, uint32_t N, Fr* dense, Fr* spread) {
  sha_spread_table_kernel<<<blocks_for(N), 256, 0, c->stream>>>(N, dense, spread);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "sha_spread_table launch failed");
}

}  // namespace cq

// =================================================================================================
// sha/src/tables.rs table generators (integer work; rows of four u64: (x, y, z, f(x,y,z)))
// =================================================================================================
namespace cq {

// `rotation::<L, N>` (tables.rs:98-103) with the reference's quirk: Bits::BITS_LEN is 8 for u8, u16 AND u32
// (tables.rs:30-38), so the word is truncated to its low 8 bits and rotated right within 8 bits.
static __device__ __forceinline__ uint64_t rot8(uint64_t word, uint32_t n) {
  const uint32_t w = (uint32_t)word & 0xffu;
  const uint32_t r = n % 8;
  return r ? (((w >> r) | (w << (8 - r))) & 0xffu) : w;
}

__global__ void sha_synthesis_table_kernel(uint32_t kind, uint32_t first, uint32_t second, uint64_t* __restrict__ out) {
  const uint64_t rows = 1ull << (first + 2 * second);
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows) return;
  // create_synthesis_table (tables.rs:78-89): x outermost, z innermost
  const uint64_t z = t & ((1ull << second) - 1);
  const uint64_t y = (t >> second) & ((1ull << second) - 1);
  const uint64_t x = t >> (2 * second);
  uint64_t f;
  if (kind == 0 || kind == 1) {
    const uint64_t xyz = (x << (2 * second)) | (y << second) | z;  // combine (:91-96)
    f = kind == 0 ? (rot8(xyz, 2) ^ rot8(xyz, 13) ^ rot8(xyz, 22))   // create_rot0_table (:113-115)
                  : (rot8(xyz, 6) ^ rot8(xyz, 11) ^ rot8(xyz, 25));  // create_rot1_table (:117-119)
  } else if (kind == 2) {
    f = (x & y) ^ (x & z) ^ (y & z);  // create_maj_table (:121-126)
  } else {
    f = (x & y) ^ ((~x) & z);  // create_ch_table (:128-133)
  }
  out[4 * t] = x;
  out[4 * t + 1] = y;
  out[4 * t + 2] = z;
  out[4 * t + 3] = f;
}

// create_decomposition_table::<L, K> (tables.rs:135-154): rows (a, x, y, z) for a in [0, 2^K)
__global__ void sha_decomposition_table_kernel(uint32_t first, uint32_t second, uint32_t kbits, uint64_t* __restrict__ out) {
  const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= (1ull << kbits)) return;
  const uint32_t full = first + 2 * second;
  const uint64_t a_mod = full >= 64 ? a : a % (1ull << full);
  const uint64_t x = a_mod >> (2 * second);
  const uint64_t y = (a_mod >> second) & ((1ull << second) - 1);
  const uint64_t z = a_mod & ((1ull << second) - 1);
  out[4 * a] = a;
  out[4 * a + 1] = x;
  out[4 * a + 2] = y;
  out[4 * a + 3] = z;
}

int sha_synthesis_table(cq_ctx* c, uint32_t kind, uint32_t first, uint32_t second, uint64_t* out) {
  const uint64_t rows = 1ull << (first + 2 * second);
  sha_synthesis_table_kernel<<<(uint32_t)((rows + 255) / 256), 256, 0, c->stream>>>(kind, first, second, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "sha_synthesis_table launch failed");
}
int sha_decomposition_table(cq_ctx* c, uint32_t first, uint32_t second, uint32_t kbits, uint64_t* out) {
  const uint64_t rows = 1ull << kbits;
  sha_decomposition_table_kernel<<<(uint32_t)((rows + 255) / 256), 256, 0, c->stream>>>(first, second, kbits, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "sha_decomposition_table launch failed");
}

// ---- StaticTableValues::new (static_lookup.rs:99-119): scaled quotients for a group of roots ---------
// row r: quotient of kate_division(table_coeffs, g) scaled by g/N, g = w^(first_root + r)
__global__ void cq_table_quotients_kernel(const Fr* __restrict__ coeffs, uint32_t N, Fr omega, Fr n_inv, uint32_t first_root,
                                          uint32_t nroots, Fr* __restrict__ out /* nroots x N, last column zero */) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nroots) return;
  const Fr g = omega.pow_u64(first_root + r);
  const Fr scale = g * n_inv;
  Fr* row = out + (size_t)r * N;
  Fr carry = Fr::zero();
  st(row + (N - 1), Fr::zero());
  for (uint32_t i = N; i-- > 1;) {  // q_{i-1} = a_i + g q_i   (arithmetic.rs:361-368)
    carry = ld(coeffs + i) + g * carry;
    st(row + (i - 1), carry * scale);
  }
}

int cq_table_quotients(cq_ctx* c, const Fr* coeffs, uint32_t N, const Fr& omega, const Fr& n_inv, uint32_t first_root,
                       uint32_t nroots, Fr* out) {
  cq_table_quotients_kernel<<<(nroots + 63) / 64, 64, 0, c->stream>>>(coeffs, N, omega, n_inv, first_root, nroots, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "table quotients launch failed");
}

// RawBytes check of an affine point array (derive/curve.rs read_raw): coordinates < q, on the curve or (0,0)
__global__ void g1_validate_kernel(const G1Affine* __restrict__ pts, uint32_t n, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(pts + i);
  uint4 a = q[0], b = q[1], cc = q[2], d = q[3];
  Fq x, y;
  x.v.l[0] = a.x; x.v.l[1] = a.y; x.v.l[2] = a.z; x.v.l[3] = a.w; x.v.l[4] = b.x; x.v.l[5] = b.y; x.v.l[6] = b.z; x.v.l[7] = b.w;
  y.v.l[0] = cc.x; y.v.l[1] = cc.y; y.v.l[2] = cc.z; y.v.l[3] = cc.w; y.v.l[4] = d.x; y.v.l[5] = d.y; y.v.l[6] = d.z; y.v.l[7] = d.w;
  auto lt_mod = [](const Fq& v) {
    for (int k = 7; k >= 0; k--) {
      if (v.v.l[k] < FqP::MOD[k]) return true;
      if (v.v.l[k] > FqP::MOD[k]) return false;
    }
    return false;
  };
  bool ok = lt_mod(x) && lt_mod(y);
  if (ok && !(x.is_zero() && y.is_zero())) ok = (y.sqr() == x.sqr() * x + Fq::from_u64(3));
  if (!ok) atomicExch(bad, 1u);
}

// SerdeFormat::RawBytes for field elements (helpers.rs:62-79): raw Montgomery limbs must be below the modulus
__global__ void fr_validate_kernel(const Fr* __restrict__ v, size_t n, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(v + i);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  bool lt = false;
  for (int k = 7; k >= 0; k--) {
    if (w[k] < FrP::MOD[k]) { lt = true; break; }
    if (w[k] > FrP::MOD[k]) break;
  }
  if (!lt) atomicExch(bad, 1u);
}

int fr_validate(cq_ctx* c, const Fr* v, size_t n, uint32_t* bad_dev) {
  if (n) fr_validate_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, c->stream>>>(v, n, bad_dev);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fr_validate launch failed");
}

int g1_validate(cq_ctx* c, const G1Affine* pts, uint32_t n, uint32_t* bad_dev) {
  if (n) g1_validate_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(pts, n, bad_dev);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "g1_validate launch failed");
}

}  // namespace cq
