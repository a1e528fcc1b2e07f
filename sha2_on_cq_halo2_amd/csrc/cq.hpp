// Internal interface of the CQ device kernels (cq.hip) and the proving-key objects.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/cq_halo2.h"
#include "curve.hpp"
#include "poly.hpp"

struct cq_ctx;
struct cq_params;

namespace cq {

constexpr uint32_t CQ_MAX_WIDTH = 8;  // table columns per vector lookup

struct CqRound1Args {
  const Fr* cols[CQ_MAX_WIDTH];      // input expression values (advice columns), n each
  const Fr* values[CQ_MAX_WIDTH];    // table values, N each
  const uint32_t* slots[CQ_MAX_WIDTH];
  uint32_t nslots[CQ_MAX_WIDTH];
  uint32_t width;
};
// several lookups of one proof in a single launch (blockIdx.y = lookup): the kernel is bound by the latency of its
// dependent probes, so independent lookups overlap instead of queueing
constexpr uint32_t CQ_ROUND1_BATCH = 8;
struct CqRound1Batch {
  CqRound1Args a[CQ_ROUND1_BATCH];
  uint32_t* m_counts[CQ_ROUND1_BATCH];
  uint32_t count;
};
struct CqThetaPowers {
  Fr pow[CQ_MAX_WIDTH];  // theta^(width-1-j)
  uint32_t width;
};
// Round-1 / round-2 front work of SEVERAL lookups in one launch (blockIdx.y = lookup): the per-lookup kernels are a few
// microseconds of work each and queue behind one another (twelve launches between theta and the inversions at L = 4).
constexpr uint32_t CQ_FOLD_BATCH = 8;
struct CqFoldBatch {
  const Fr* src[CQ_FOLD_BATCH][CQ_MAX_WIDTH];  // the w vectors folded with theta: lookup inputs (n each) or table columns (N each)
  uint32_t width[CQ_FOLD_BATCH];
  Fr* out[CQ_FOLD_BATCH];            // round 1: f = sum_j theta^(w-1-j) src_j (n elements); may be nullptr (skip)
  const Fr* f[CQ_FOLD_BATCH];        // round 2: f (n), or nullptr when this rank does not hold b of that lookup
  Fr* b[CQ_FOLD_BATCH];              // round 2: B_r = f_r + beta (r < u), beta (r >= u)                      (n)
  const uint32_t* m[CQ_FOLD_BATCH];  // round 2: multiplicities                                              (N)
  Fr* den[CQ_FOLD_BATCH];            // round 2: m ? t + beta : 0 with t = sum_j theta^(w-1-j) src_j          (N)
  Fr theta_pow[CQ_MAX_WIDTH];        // theta^0 .. theta^(CQ_MAX_WIDTH - 1)
  uint32_t count;
};
int cq_fold_inputs(cq_ctx* c, const CqFoldBatch& b, uint32_t n);
int cq_round2_prep(cq_ctx* c, const CqFoldBatch& b, uint32_t n, uint32_t N, uint32_t u, const Fr& beta);
struct CqAValuesBatch {
  const Fr* den_inv[CQ_FOLD_BATCH];
  const uint32_t* m[CQ_FOLD_BATCH];
  Fr* a[CQ_FOLD_BATCH];
  Fr* a_scaled[CQ_FOLD_BATCH];
  uint32_t width[CQ_FOLD_BATCH];
  Fr theta_pow[CQ_MAX_WIDTH];
  uint32_t count;
};
int cq_a_values_batch(cq_ctx* c, const CqAValuesBatch& b, uint32_t N);

struct ShaCols {
  Fr* p[16];
};

int cq_table_build_index(cq_ctx* c, const Fr* values, uint32_t N, uint32_t** slots_out, uint32_t* nslots_out);
int cq_round1(cq_ctx* c, const CqRound1Batch& b, uint32_t u, uint32_t* err_dev);
int cq_a_denominators(cq_ctx* c, const Fr* t, const uint32_t* m, uint32_t N, const Fr& beta, Fr* den);
int cq_a_values(cq_ctx* c, const Fr* den_inv, const uint32_t* m, uint32_t N, const CqThetaPowers& tp, Fr* a, Fr* a_scaled);
int cq_m_to_fr(cq_ctx* c, const uint32_t* m, uint32_t N, Fr* out);
int cq_qs_scalars(cq_ctx* c, const Fr* values, uint32_t N, const Fr& ts, const Fr& s, const Fr& omega, const Fr& n_inv, Fr* out);
int sha_witness_fill(cq_ctx* c, const uint32_t* words_dev, uint32_t nwords, uint32_t pairs, uint32_t n, const ShaCols& cols);
int sha_spread_table(cq_ctx* c, uint32_t N, Fr* dense, Fr* spread);
int sha_synthesis_table(cq_ctx* c, uint32_t kind, uint32_t first, uint32_t second, uint64_t* out);
int sha_decomposition_table(cq_ctx* c, uint32_t first, uint32_t second, uint32_t kbits, uint64_t* out);
int cq_table_quotients(cq_ctx* c, const Fr* coeffs, uint32_t N, const Fr& omega, const Fr& n_inv, uint32_t first_root,
                       uint32_t nroots, Fr* out);
int g1_validate(cq_ctx* c, const G1Affine* pts, uint32_t n, uint32_t* bad_dev);
int fr_validate(cq_ctx* c, const Fr* v, size_t n, uint32_t* bad_dev);

}  // namespace cq

namespace cq {
// contiguous slice [lo, hi) of an n-term multiexp (or of `n` columns) owned by `rank`; sizes differ by at most one
inline void shard_range(size_t n, uint32_t rank, uint32_t world, size_t& lo, size_t& hi) {
  const size_t base = n / world, rem = n % world;
  lo = rank * base + (rank < rem ? rank : rem);
  hi = lo + base + (rank < rem ? 1 : 0);
}
}  // namespace cq

// ParamsKZG G1 part (poly/kzg/commitment.rs:31-39), SRS resident in HBM
struct cq_params {
  cq_ctx* ctx;
  uint32_t k;
  size_t n;
  cq::G1Affine* g;           // [s^i]_1
  cq::G1Affine* g_lagrange;  // [L_i(s)]_1
  uint32_t key_users = 0;    // proving keys built on this object (a sharded key drops the whole-array tables only when alone)
  bool owner_released = false;  // cq_params_destroy was called while keys still used the object: the last key frees it
};

// StaticTableConfig (plonk/static_lookup.rs:47-66): Lagrange SRS of the table-sized domain
struct cq_table_config {
  cq_ctx* ctx;
  uint32_t log_n;
  size_t N;
  cq::G1Affine* g1_lagrange = nullptr;
  cq::G1Affine* g_lagrange_opening_at_0 = nullptr;
  uint32_t key_users = 0;
  bool owner_released = false;  // as for cq_params
};

// StaticTableValues (plonk/static_lookup.rs:68-75)
struct cq_static_table {
  cq_ctx* ctx;
  size_t N;
  cq::Fr* values = nullptr;
  cq::G1Affine* qs = nullptr;  // cached quotient commitments, affine
  uint32_t* slots = nullptr;   // value -> index hash table
  uint32_t nslots = 0;
};

struct cq_lookup_desc {
  std::vector<uint32_t> cols;             // advice column per table column (input = advice[col] @ Rotation::cur())
  std::vector<int64_t> prog;              // or: word offset of the input's program in cq_pk::lookup_prog (-1 = plain column)
  std::vector<cq_static_table*> tables;
};

// the slice of ProvingKey (plonk.rs:291-308) the CQ-only proving path reads
struct cq_pk {
  cq_ctx* ctx;
  cq_params* params;
  cq_domain* domain = nullptr;
  uint32_t k, num_advice, bf, u;
  std::vector<cq_lookup_desc> lookups;
  std::vector<std::pair<uint32_t, int32_t>> advice_queries;  // (column, rotation), registration order
  cq_table_config* table_cfg;
  cq::G1Affine* b0_g1_bound = nullptr;  // n-1 points
  bool own_b0 = false;
  cq::Fr* l_active_row = nullptr;       // extended coset
  // ---- general PLONK part (empty for a CQ-only circuit) ----
  uint32_t num_fixed = 0, num_instance = 0, cs_degree = 3;
  std::vector<std::pair<uint32_t, int32_t>> fixed_queries;
  cq::Fr* fixed_values = nullptr;   // num_fixed x n   (pk.fixed_values, keygen.rs:320-326)
  cq::Fr* fixed_polys = nullptr;    // num_fixed x n   (:328-331)
  cq::Fr* fixed_cosets = nullptr;   // num_fixed x ext (:333-336)
  cq::Fr* l0 = nullptr;             // ext (:340-345)
  cq::Fr* l_last = nullptr;         // ext (:357-363)
  uint32_t num_gate_polys = 0;
  uint32_t* gate_prog = nullptr;    // device: [len, words...] per polynomial
  cq::Fr* constants = nullptr;      // device
  uint32_t* lookup_prog = nullptr;  // device: [1-poly program] per expression-valued lookup input
  bool lookup_exprs = false;
  // legacy lookups: per lookup the number of (input, table) expression pairs and the word offsets of the two
  // program lists ([len, words...] x width each) in `legacy_prog`
  struct LegacyLookup { uint32_t width; size_t in_off, tab_off; };
  std::vector<LegacyLookup> legacy;
  uint32_t* legacy_prog = nullptr;
  std::vector<std::pair<uint32_t, uint32_t>> perm_columns;  // (CQ_COL_*, index) = cs.permutation.columns
  cq::Fr* perm_values = nullptr;    // columns x n   (permutation::ProvingKey::permutations)
  cq::Fr* perm_polys = nullptr;     // columns x n   (::polys)
  cq::Fr* perm_cosets = nullptr;    // columns x ext (::cosets)
  cq::Fr* omega_powers = nullptr;   // omega^i, i < n
  cq::Fr* ext_pow_lo = nullptr;     // extended_omega^t, t < 256
  cq::Fr* ext_pow_hi = nullptr;     // extended_omega^(256 b), b < max(ext / 256, 1)
  int opener = CQ_OPENER_GWC;
  cq_rng_fill_fn rng_fill = nullptr;     // caller's bulk form of its RNG (cq_pk_set_rng_fill)
  std::vector<uint8_t> advice_phase;     // phase of every advice column
  std::vector<uint8_t> challenge_phase;  // phase after which user challenge i is squeezed
  uint32_t num_phases = 1;
  bool general() const { return num_gate_polys || !perm_columns.empty() || !legacy.empty(); }
  size_t perm_sets() const {
    const size_t chunk = cs_degree - 2;
    return (perm_columns.size() + chunk - 1) / chunk;
  }
  cq::Fr vk_repr;
  std::vector<cq::G1Affine*> qs_concat;  // per lookup: [qs_0 | qs_1 | ...] (width*N points)
  // MSM sharding across ranks (one process per GPU): every rank commits its slice of each point range,
  // partial results are all-gathered through the caller's collective and summed locally
  uint32_t shard_rank = 0, shard_world = 1;
  // single-rank communicator: the collectives run over one rank (the path a 1-GPU box can exercise on hardware)
  bool shard_single = false;
  bool sharded() const { return shard_world > 1 || shard_single; }
  cq_allgather_fn allgather = nullptr;  // nullptr with shard_world > 1: the context's RCCL communicator
  void* allgather_user = nullptr;
  // column sharding: independent column transforms are split between the ranks by owner and the results
  // broadcast (comm.hpp); `bcast`: host-buffer transport for it (nullptr: RCCL)
  bool shard_columns = true;
  cq_bcast_fn bcast = nullptr;
  void* bcast_user = nullptr;
  // resident column sharding (cq_pk_set_resident_sharding): columns stay on their owner, slices travel point to point
  bool shard_resident = false;
  cq_exchange_fn exchange = nullptr;  // host-buffer transport for it (nullptr: ncclSend / ncclRecv)
  void* exchange_user = nullptr;
  // MSM window tables: the width this key's launches use (its SRS length decides; small arrays shared with keys of other
  // sizes get a second table of this width), the registry entries the key holds a reference of, and -- when sharded --
  // the tables of the rank's slices (capi_cq.hip: pk_shard_tables)
  uint32_t table_c = 0;
  struct TableRef { const void* bases; size_t n; uint32_t c; };
  std::vector<TableRef> held_tables, shard_tables;
  bool dropped_params_tables = false, dropped_cfg_tables = false;  // whole-array tables given up for slice tables
  bool counted_users = false;
};
