// Host orchestrator: `create_proof` (halo2_proofs/src/plonk/prover.rs:51-779) with the sub-arguments
// of plonk/static_lookup/prover.rs, plonk/permutation/prover.rs, plonk/vanishing/prover.rs,
// plonk/evaluation.rs:285-551 and poly/kzg/multiopen/gwc/prover.rs.  Everything O(n) runs on the GPU
// (polynomials never leave HBM); the host keeps the Fiat-Shamir transcript (transcript.rs:170-241),
// draws the blinding scalars from the caller's RNG in the reference's order, folds MSM window sums
// and normalises the handful of commitment points.  Call order, RNG order and transcript order are
// the contract: the proof bytes equal the reference's for
// the same (pk, witness, RNG stream).
//
// Scope: one circuit with advice / fixed / instance columns, custom gates (postfix programs, plonk.hip), the
// permutation argument, static (CQ) lookups whose inputs are arbitrary expressions, legacy plookup-style lookups
// (grand product and the sort of `permute_expression_pair` on the GPU, lksort.hip), multi-phase circuits with user
// challenges (cq_create_proof_phases), ProverGWC or ProverSHPLONK.  One deliberate omission:
// evaluation.rs:317-335 transforms every advice / instance polynomial to the extended coset even when no term
// reads them (a CQ-only circuit); that dead work is skipped.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "blake2b.hpp"
#include "comm.hpp"
#include "cq.hpp"
#include "ctx.hpp"
#include "msm.hpp"
#include "plonk.hpp"
#include "prover.hpp"
#include "glv.hpp"
#include "xoshiro.hpp"

using namespace cq;

namespace {

struct Transcript {
  Blake2b st;
  std::vector<uint8_t> proof;
  Transcript() { st.init(64, "Halo2-Transcript"); }
  void common_scalar(const Fr& s) {
    const uint8_t tag = 2;
    st.update(&tag, 1);
    U256 c = s.to_canonical();
    st.update((const uint8_t*)c.l, 32);
  }
  // transcript.rs:221-233; identity is an error
  bool common_point(const G1Affine& p) {
    if (p.is_identity()) return false;
    const uint8_t tag = 1;
    st.update(&tag, 1);
    U256 x = p.x.to_canonical(), y = p.y.to_canonical();
    st.update((const uint8_t*)x.l, 32);
    st.update((const uint8_t*)y.l, 32);
    return true;
  }
  bool write_point(const G1Affine& p) {
    if (!common_point(p)) return false;
    U256 x = p.x.to_canonical(), y = p.y.to_canonical();
    uint8_t b[32];
    memcpy(b, x.l, 32);
    b[31] |= (uint8_t)((y.l[0] & 1) << 7);  // derive/curve.rs:635-646
    proof.insert(proof.end(), b, b + 32);
    return true;
  }
  void write_scalar(const Fr& s) {
    common_scalar(s);
    U256 c = s.to_canonical();
    const uint8_t* b = (const uint8_t*)c.l;
    proof.insert(proof.end(), b, b + 32);
  }
  Fr squeeze() {
    const uint8_t tag = 0;
    st.update(&tag, 1);
    uint8_t out[64];
    st.finalize_clone(out);
    uint64_t w[8];
    memcpy(w, out, 64);
    return Fr::from_u512(w);  // from_bytes_wide (transcript.rs:300-309)
  }
};

// derive/curve.rs:362-397 `batch_normalize`
void batch_normalize(const std::vector<G1Jac>& in, std::vector<G1Affine>& out) {
  out.resize(in.size());
  std::vector<Fq> pref(in.size());
  Fq acc = Fq::one();
  for (size_t i = 0; i < in.size(); i++) {
    pref[i] = acc;
    if (!in[i].is_identity()) acc = acc * in[i].z;
  }
  acc = acc.inv();
  for (size_t i = in.size(); i-- > 0;) {
    if (in[i].is_identity()) {
      out[i] = G1Affine::identity();
      continue;
    }
    Fq zi = pref[i] * acc;
    acc = acc * in[i].z;
    Fq zi2 = zi.sqr();
    out[i] = {in[i].x * zi2, in[i].y * zi2 * zi};
  }
}

G1Jac jac_from_limbs(const uint64_t* l) {
  return {Fq::from_limbs64(l), Fq::from_limbs64(l + 4), Fq::from_limbs64(l + 8)};
}

struct Rng {
  cq_rng_next_u64 next;
  void* state;
  cq_rng_fill_fn bulk = nullptr;  // the caller's own bulk form (cq_pk_set_rng_fill)
  HostPool* pool = nullptr;       // the context's worker threads
  // Fr::random (bn256/fr.rs:159-170): eight next_u64, low limb first
  void words(uint64_t* w8) {
    for (int i = 0; i < 8; i++) w8[i] = next(state);
  }
  Fr fr() {
    uint64_t w[8];
    words(w);
    return Fr::from_u512(w);
  }
  // `count` consecutive draws; the library's own generators are recognised and run inline (the indirect
  // call per u64 costs more than the generator: 2^21 draws per k=18 proof)
  void fill(uint64_t* dst, size_t count) {
    if (bulk) {
      bulk(state, dst, count);
    } else if (next == cq_xoshiro256ss_next_u64) {
      xoshiro_fill((uint64_t*)state, dst, count, 8, pool);  // several threads for long runs, the same stream (xoshiro.hpp)
    } else if (next == cq_buffer_rng_next_u64) {
      cq_buffer_rng* b = (cq_buffer_rng*)state;
      const size_t avail = b->pos < b->len ? b->len - b->pos : 0, take = std::min(avail, count);
      memcpy(dst, b->words + b->pos, take * sizeof(uint64_t));
      memset(dst + take, 0, (count - take) * sizeof(uint64_t));
      b->pos += take;
      b->overrun += count - take;  // create_proof fails on an exhausted stream (capi_cq.hip)
    } else {
      for (size_t i = 0; i < count; i++) dst[i] = next(state);
    }
  }
};

// Commitments of one transcript round.  begin() enqueues the (possibly sharded) MSMs, end() waits,
// exchanges partial sums between ranks when sharded, and normalises (batch_normalize).
struct Commit {
  const cq_pk* pk;
  MsmPending pend;
  size_t count = 0;
  int begin(const cq_pk* pk_, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases,
            const std::vector<size_t>& lens) {
    pk = pk_;
    count = scalars.size();
    if (!pk->sharded()) return msm_multi_begin(pk->ctx, scalars.data(), bases.data(), lens.data(), count, pend);
    std::vector<const Fr*> sc(count);
    std::vector<const G1Affine*> bs(count);
    std::vector<size_t> ln(count);
    for (size_t j = 0; j < count; j++) {
      size_t lo, hi;
      shard_range(lens[j], pk->shard_rank, pk->shard_world, lo, hi);
      sc[j] = scalars[j] + lo;
      bs[j] = bases[j] + lo;
      ln[j] = hi - lo;
    }
    return msm_multi_begin(pk->ctx, sc.data(), bs.data(), ln.data(), count, pend);
  }
  int end(std::vector<G1Affine>& out) {
    cq_ctx* c = pk->ctx;
    std::vector<uint64_t> jac(count * 12);
    static const bool trace_host = getenv("CQ_TRACE_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (trace_host) hipStreamSynchronize(c->stream);
    const auto t1 = std::chrono::steady_clock::now();
    int rc = msm_multi_end(c, pend, jac.data());
    if (rc != CQ_OK) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    struct Report {
      bool on; std::chrono::steady_clock::time_point a, b, c_;
      ~Report() {
        if (on) fprintf(stderr, "[cq host]   commit end: wait %.1f us, fold %.1f us, exchange + normalise %.1f us\n",
                        std::chrono::duration<double, std::micro>(b - a).count(), std::chrono::duration<double, std::micro>(c_ - b).count(),
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c_).count());
      }
    } report{trace_host, t0, t1, t2};
    std::vector<G1Jac> j(count);
    if (pk->sharded()) {
      // all-gather of count x 96 B per rank, then the local EC sum (not an RCCL reduction op)
      std::vector<uint64_t> all((size_t)pk->shard_world * count * 12);
      int arc = shard_allgather_host(pk, jac.data(), all.data(), count * 12 * sizeof(uint64_t));
      if (arc != CQ_OK) return arc;
      for (size_t i = 0; i < count; i++) {
        G1Jac acc = G1Jac::identity();
        for (uint32_t r = 0; r < pk->shard_world; r++) acc = jac_add(acc, jac_from_limbs(all.data() + ((size_t)r * count + i) * 12));
        j[i] = acc;
      }
    } else {
      for (size_t i = 0; i < count; i++) j[i] = jac_from_limbs(jac.data() + 12 * i);
    }
    batch_normalize(j, out);
    return CQ_OK;
  }
};

// A decision that changes the STRUCTURE of a round (how many launches, hence how many collectives and of what size) must
// come out the same on every rank: true everywhere iff `flag` is set on any rank (one 8-byte all-gather).
int shard_any(const cq_pk* pk, bool flag, bool& out) {
  out = flag;
  if (!pk->sharded()) return CQ_OK;
  const uint64_t mine = flag ? 1 : 0;
  std::vector<uint64_t> all(pk->shard_world, 0);
  int arc = shard_allgather_host(pk, &mine, all.data(), sizeof(uint64_t));
  if (arc != CQ_OK) return arc;
  for (uint64_t v : all) out = out || v != 0;
  return CQ_OK;
}

// Work for the side stream (ctx.hpp): between begin() and end() every library call that enqueues on `c->stream`
// lands on the low-priority stream instead, ordered after the accumulate kernel of the MSM launch queued last (all
// earlier kernels of the main stream are complete by then, and what remains of the MSM only touches its own
// workspace).  join_aux() makes the main stream wait for it.  The destructor restores the context on an error path.
struct AuxFork {
  cq_ctx* c;
  hipStream_t main = nullptr;
  int saved_slot = 0;
  bool active = false;
  explicit AuxFork(cq_ctx* c_) : c(c_) {}
  // `seq_before`: c->msm_tail_seq sampled before the MSMs were queued
  int begin(uint64_t seq_before) {
    if (c->msm_tail_seq == seq_before) CQ_HIP(c, hipEventRecord(c->msm_tail_event, c->stream));  // no launch: plain ordering
    CQ_HIP(c, hipStreamWaitEvent(c->aux_stream, c->msm_tail_event, 0));
    main = c->stream;
    saved_slot = c->ntt_scratch_slot;
    c->stream = c->aux_stream;
    c->ntt_scratch_slot = 8;
    active = true;
    return CQ_OK;
  }
  int end() {
    if (!active) return CQ_OK;
    restore();
    c->aux_pending = true;
    CQ_HIP(c, hipEventRecord(c->aux_done, c->aux_stream));
    return CQ_OK;
  }
  void restore() {
    c->stream = main;
    c->ntt_scratch_slot = saved_slot;
    active = false;
  }
  ~AuxFork() {
    if (active) {
      restore();
      hipStreamSynchronize(c->aux_stream);
    }
  }
};
static int join_aux(cq_ctx* c) {
  if (!c->aux_pending) return CQ_OK;
  c->aux_pending = false;
  CQ_HIP(c, hipStreamWaitEvent(c->stream, c->aux_done, 0));
  return CQ_OK;
}

int commit_batch_v(const cq_pk* pk, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases,
                   const std::vector<size_t>& lens, std::vector<G1Affine>& out) {
  Commit cm;
  int rc = cm.begin(pk, scalars, bases, lens);
  if (rc != CQ_OK) return rc;
  return cm.end(out);
}

int commit_batch(const cq_pk* pk, const std::vector<const Fr*>& scalars, const std::vector<const G1Affine*>& bases, size_t len,
                 std::vector<G1Affine>& out) {
  std::vector<size_t> lens(scalars.size(), len);
  return commit_batch_v(pk, scalars, bases, lens, out);
}

}  // namespace

#define CQ_TRY(x)               \
  do {                          \
    int _rc = (x);              \
    if (_rc != CQ_OK) return _rc; \
  } while (0)

namespace cq {

namespace {

// Proof-lifetime device buffers, carved from one arena (scratch slot 6).  With base == nullptr the same
// code only counts elements.
struct Arena {
  Fr* base;
  size_t used = 0;
  Fr* take(size_t elems) {
    Fr* p = base ? base + used : nullptr;
    used += elems;
    return p;
  }
};
struct Buffers {
  Fr *adv, *adv_coeff, *inst_lag, *inst_coeff, *f_lag, *f_coeff, *bpoly, *random_poly, *z, *mv, *cosets, *adv_cosets, *inst_cosets,
      *z_cosets, *lk_inputs, *plk, *plk_cosets, *h_ext, *h_coeff, *gwc_batch, *gwc_wit, *shplonk, *t_comp, *den, *a_val, *m_fr, *a_scaled;
  Fr* challenges;  // user challenges (Expression::Challenge), uploaded as the phases complete
  Fr *tails, *gather;  // blinding rows of a phase's advice columns (staging); scalars on their way to the host
  uint64_t* rng_dev;
  uint32_t *m_counts, *err_dev;
};

std::vector<int32_t> opening_rotations(const cq_pk* pk) {
  std::vector<int32_t> rots{0};
  auto seen = [&](int32_t r) {
    if (std::find(rots.begin(), rots.end(), r) == rots.end()) rots.push_back(r);
  };
  for (auto& q : pk->advice_queries) seen(q.second);
  for (auto& q : pk->fixed_queries) seen(q.second);
  if (pk->perm_sets() || !pk->legacy.empty()) seen(1);
  if (!pk->legacy.empty()) seen(-1);
  if (pk->perm_sets() > 1) seen(-(int32_t)(pk->bf + 1));
  return rots;
}

bool early_advice_polys(const cq_pk* pk) {
  return !pk->general() && pk->perm_sets() == 0 && pk->num_phases == 1 && pk->num_advice > 0 && !pk->lookups.empty();
}

void carve(const cq_pk* pk, Arena& ar, Buffers& b) {
  const size_t n = (size_t)1 << pk->k, ext = pk->domain->ext();
  const size_t L = pk->lookups.size(), A = pk->num_advice, I = pk->num_instance, S = pk->perm_sets();
  const size_t N = pk->table_cfg ? pk->table_cfg->N : 0;
  const bool general = pk->general();
  size_t wsum = 0;
  for (auto& lk : pk->lookups) wsum += lk.cols.size();
  const size_t npts = opening_rotations(pk).size();
  b.adv = ar.take(A * n);            // advice (lagrange -> coeff in place)
  // CQ-only single-phase circuits: the coefficient forms go to a buffer of their own, so that the transform can start
  // under the advice launch's tail while the Lagrange values are still needed (round 1 folds them into f)
  b.adv_coeff = ar.take(early_advice_polys(pk) ? A * n : 0);
  b.inst_lag = ar.take(I * n);
  b.inst_coeff = ar.take(I * n);
  b.f_lag = ar.take(L * n);
  b.f_coeff = ar.take(L * n);
  b.bpoly = ar.take(L * n);
  b.den = ar.take(L * N + 8);        // directly after bpoly: both are inverted by one batch inversion
  b.random_poly = ar.take(n);
  b.z = ar.take(S * n);              // permutation products (lagrange -> coeff in place)
  b.mv = ar.take(S * n);
  b.cosets = ar.take(2 * L * ext);   // b, f on the extended coset
  b.adv_cosets = ar.take(general ? A * ext : 0);
  b.inst_cosets = ar.take(general ? I * ext : 0);
  b.z_cosets = ar.take(S * ext);
  b.lk_inputs = ar.take(pk->lookup_exprs ? wsum * n : 0);  // evaluated input expressions of the static lookups
  // legacy lookups, per lookup: compressed input / table, permuted input / table, product (n each; the last three
  // become coefficients in place) and the five extended-coset vectors of its quotient terms
  b.plk = ar.take(pk->legacy.size() * 5 * n);
  b.plk_cosets = ar.take(pk->legacy.size() * 5 * ext);
  b.h_ext = ar.take(ext);
  b.h_coeff = ar.take(ext);          // n * (degree - 1) coefficients
  const bool shplonk = pk->opener == CQ_OPENER_SHPLONK;
  b.gwc_batch = ar.take(shplonk ? 0 : npts * n);
  b.gwc_wit = ar.take(shplonk ? 0 : npts * n);
  b.shplonk = ar.take(shplonk ? 5 * n + 64 * 8 : 0);  // h, two division buffers, h_x, l_x, low-degree remainders
  b.t_comp = ar.take(0);             // (the compressed table is folded inside cq_round2_prep since round 3)
  b.a_val = ar.take(L * N);
  b.m_fr = ar.take(L * N);
  b.a_scaled = ar.take(wsum * N);
  b.challenges = ar.take(pk->challenge_phase.size() + 8);
  b.tails = ar.take(A * (pk->bf + 1) + 8);
  b.gather = ar.take(GATHER_MAX);
  b.rng_dev = (uint64_t*)ar.take(2 * n);  // 64 B per element = 2 Fr
  b.m_counts = (uint32_t*)ar.take(L * N / 8 + 64);
  b.err_dev = b.m_counts ? b.m_counts + L * N : nullptr;
}

// sum_i coeff_i * p_i - sub_const, any number of terms (LINCOMB_MAX per launch; later launches fold the
// partial result back in as one more term)
struct Term {
  const Fr* p;
  uint32_t len;
  Fr coeff;
};
int lincomb_many(cq_ctx* c, const std::vector<Term>& terms, const Fr& sub_const, uint32_t n, Fr* out) {
  size_t done = 0;
  bool first = true;
  do {
    LincombArgs la;
    la.count = 0;
    la.sub_const = Fr::zero();
    if (!first) {
      la.p[0] = out;
      la.len[0] = n;
      la.coeff[0] = Fr::one();
      la.count = 1;
    }
    while (done < terms.size() && la.count < LINCOMB_MAX) {
      la.p[la.count] = terms[done].p;
      la.len[la.count] = terms[done].len;
      la.coeff[la.count] = terms[done].coeff;
      la.count++;
      done++;
    }
    if (done == terms.size()) la.sub_const = sub_const;
    int rc = poly_lincomb(c, la, n, out);
    if (rc != CQ_OK) return rc;
    first = false;
  } while (done < terms.size());
  return CQ_OK;
}

int eval_many(cq_ctx* c, const std::vector<const Fr*>& ps, const std::vector<uint32_t>& ls, const Fr& z, Fr* out) {
  for (size_t off = 0; off < ps.size(); off += EVAL_MAX_BATCH) {
    const uint32_t cnt = (uint32_t)std::min((size_t)EVAL_MAX_BATCH, ps.size() - off);
    int rc = poly_eval_batch(c, ps.data() + off, ls.data() + off, cnt, z, out + off);
    if (rc != CQ_OK) return rc;
  }
  return CQ_OK;
}

}  // namespace

size_t prover_arena_elems(const cq_pk* pk) {
  Arena ar{nullptr};
  Buffers b;
  carve(pk, ar, b);
  return ar.used + 1024;
}

int create_proof_dev(cq_pk* pk, const uint64_t* const* advice_dev, const uint64_t* const* instances,
                     const size_t* instance_lens, cq_phase_fn phase_fn, void* phase_user, cq_rng_next_u64 rng_next,
                     void* rng_state, std::vector<uint8_t>& proof_out) {
  cq_ctx* c = pk->ctx;
  cq_domain* dom = pk->domain;
  const uint32_t k = pk->k;
  const size_t n = (size_t)1 << k, ext = dom->ext();
  const uint32_t bf = pk->bf, u = pk->u;
  const size_t L = pk->lookups.size(), A = pk->num_advice, I = pk->num_instance;
  const size_t N = pk->table_cfg ? pk->table_cfg->N : 0;
  const size_t S = pk->perm_sets(), PC = pk->perm_columns.size(), chunk_len = pk->cs_degree - 2;
  const bool general = pk->general();
  const size_t PL = pk->legacy.size();
  hipStream_t s = c->stream;
  Rng rng{rng_next, rng_state, pk->rng_fill, &c->pool()};
  Transcript tr;
  // CQ_TRACE_HOST=1: microseconds since the start of the proof at the host's milestones, on stderr (development aid)
  static const bool trace_host = getenv("CQ_TRACE_HOST") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace_host) fprintf(stderr, "[cq host] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(), what);
  };

  // Resident column sharding (cq_pk_set_resident_sharding; DESIGN.md, multi-GPU): for the CQ-shaped circuits of the
  // BASELINE configs -- advice columns, static lookups on plain advice inputs, one phase, ProverGWC -- a transformed
  // column stays on its owner and only slices travel.  Ownership: contiguous ranges (shard_range) of the lookups and of
  // the advice columns; consumers by point range, as the MSMs are cut.  Marked "resident:" below where the proof
  // departs from the replicated flow; every rank still derives the same transcript.
  bool resident = shard_resident_enabled(pk) && !general && PL == 0 && pk->num_phases == 1 && pk->challenge_phase.empty() && I == 0 &&
                  L > 0 && pk->opener == CQ_OPENER_GWC && n >= (size_t)64 * pk->shard_world;
  for (size_t l = 0; l < L && resident; l++)
    for (int64_t pj : pk->lookups[l].prog) resident = resident && pj < 0;
  const uint32_t SW = pk->shard_world, me = pk->shard_rank;
  size_t lk_lo = 0, lk_hi = L, adv_lo = 0, adv_hi = A;  // the lookups / advice columns this rank owns
  if (resident) {
    shard_range(L, me, SW, lk_lo, lk_hi);
    shard_range(A, me, SW, adv_lo, adv_hi);
  }
  auto owner_of = [&](size_t item, size_t count) -> uint32_t {
    for (uint32_t r = 0; r < SW; r++) {
      size_t lo, hi;
      shard_range(count, r, SW, lo, hi);
      if (item >= lo && item < hi) return r;
    }
    return 0;
  };
  const size_t lk_cnt = lk_hi - lk_lo;

  // ---- set-up that can fail on this rank alone (allocations: arena, pinned staging, streams, twiddle tables) comes first,
  //      and a sharded proof agrees on its outcome before anything else is exchanged: a rank that cannot start says so in
  //      a one-word all-gather and every rank returns an error, instead of the others waiting in the first collective of a
  //      proof one of them never joins.  (A failure later on one rank alone -- the MSM workspace, a HIP error -- aborts that
  //      rank's communicator, and its peers' waits time out: capi_cq.hip, comm.hip.)
  void *arena_v = nullptr, *small_v = nullptr, *pin = nullptr, *res_stage_v = nullptr;
  // resident: staging for the slices a rank receives -- per exchange at most one slice from every rank (or every owner
  // and piece), plus the rank's own range of the opening's batch polynomial with its two edge elements
  const size_t res_slice_max = (n + pk->shard_world - 1) / pk->shard_world + 1;
  const size_t res_stage_elems = (size_t)pk->shard_world * res_slice_max * (pk->cs_degree > 2 ? pk->cs_degree - 1 : 1) + res_slice_max + 64;
  {
    auto setup = [&]() -> int {
      CQ_TRY(c->ensure_scratch(6, prover_arena_elems(pk) * sizeof(Fr), &arena_v));
      CQ_TRY(c->ensure_pinned_small(&small_v));
      CQ_TRY(c->ensure_pinned((size_t)64 * n + A * (n - u) * sizeof(Fr) + 64, &pin));
      CQ_TRY(c->ensure_aux_stream());
      CQ_TRY(c->ensure_copy_stream());
      if (resident) CQ_TRY(c->ensure_scratch(7, res_stage_elems * sizeof(Fr), &res_stage_v));  // where received slices land
      int trc = CQ_OK;
      if (!c->tables_for(dom->k, dom->omega_inv, &trc) || !c->tables_for(dom->extended_k, dom->extended_omega, &trc)) return trc;
      return CQ_OK;
    };
    const int setup_rc = getenv("CQ_TEST_FAIL_SETUP") && atoi(getenv("CQ_TEST_FAIL_SETUP")) == (int)pk->shard_rank + 1
                             ? c->fail(CQ_ERR_HIP, "set-up failure injected by CQ_TEST_FAIL_SETUP") : setup();
    if (pk->sharded()) {
      bool any = false;
      const std::string mine = c->err;
      CQ_TRY(shard_any(pk, setup_rc != CQ_OK, any));
      if (any) return setup_rc != CQ_OK ? c->fail(setup_rc, mine) : c->fail(CQ_ERR_INTERNAL, "sharded proof: another rank could not set up its proof (allocation failure there)");
    } else if (setup_rc != CQ_OK) {
      return setup_rc;
    }
  }
  Arena ar{(Fr*)arena_v};
  Buffers B;
  carve(pk, ar, B);
  Fr *adv = B.adv, *f_lag = B.f_lag, *f_coeff = B.f_coeff, *bpoly = B.bpoly, *random_poly = B.random_poly, *cosets = B.cosets,
     *h_ext = B.h_ext, *h_coeff = B.h_coeff, *den = B.den, *a_val = B.a_val, *m_fr = B.m_fr,
     *a_scaled = B.a_scaled;
  uint64_t* rng_dev = B.rng_dev;
  uint32_t* m_counts = B.m_counts;
  const bool early_adv = early_advice_polys(pk);
  Fr* const adv_poly = early_adv ? B.adv_coeff : adv;  // where the advice polynomials (coefficient form) live
  bool adv_is_coeff = false;

  // Column sharding (cq_pk_set_column_sharding): a batch of independent column transforms is split
  // between the ranks by owner -- contiguous ranges of columns, cq::shard_range -- and every rank's output columns are
  // broadcast from their owner on the stream the transform ran on (one grouped RCCL launch).  Transforms chained on the
  // same batch (coefficients, then the coset of the same columns) read what their own rank wrote, so they need not
  // wait for the exchange.  Unsharded, or with fewer than two columns: the plain call.
  auto sharded_transform = [&](bool to_extended, const Fr* in, Fr* out, uint32_t batch) -> int {
    const size_t in_stride = n, out_stride = to_extended ? ext : n;
    auto run = [&](size_t first, size_t count) -> int {
      if (!count) return CQ_OK;
      return to_extended ? domain_coeff_to_extended(dom, in + first * in_stride, out + first * out_stride, (uint32_t)count, in_stride, out_stride)
                         : domain_lagrange_to_coeff(dom, in + first * in_stride, out + first * out_stride, (uint32_t)count, in_stride, out_stride);
    };
    if (!shard_columns_enabled(pk) || (batch < 2 && !pk->shard_single)) return run(0, batch);
    size_t lo, hi;
    shard_range(batch, pk->shard_rank, pk->shard_world, lo, hi);
    CQ_TRY(run(lo, hi - lo));
    std::vector<BcastPart> parts;
    for (uint32_t r = 0; r < pk->shard_world; r++) {
      shard_range(batch, r, pk->shard_world, lo, hi);
      if (hi > lo) parts.push_back({out + lo * out_stride, (hi - lo) * out_stride * sizeof(Fr), r});
    }
    return shard_bcast_parts(pk, parts.data(), parts.size(), c->stream);
  };
  auto lagrange_to_coeff_cols = [&](const Fr* in, Fr* out, size_t batch) { return sharded_transform(false, in, out, (uint32_t)batch); };
  auto coeff_to_extended_cols = [&](const Fr* in, Fr* out, size_t batch) { return sharded_transform(true, in, out, (uint32_t)batch); };

  // side stream: drain whatever an aborted proof may have left there (its NTTs find their twiddle tables built: set-up above)
  if (c->aux_pending) {
    CQ_HIP(c, hipStreamSynchronize(c->aux_stream));
    c->aux_pending = false;
  }

  // prover.rs:85 -- vk.hash_into(transcript)
  tr.common_scalar(pk->vk_repr);

  // ---- instance columns (prover.rs:100-131), absorbed as scalars in phase 0 (:305-312) ---------------
  if (I) {
    CQ_HIP(c, hipMemsetAsync(B.inst_lag, 0, I * n * sizeof(Fr), s));
    for (size_t i = 0; i < I; i++) {
      if (instance_lens[i] > u) return c->fail(CQ_ERR_ARG, "Error::InstanceTooLarge");  // :108-110
      if (instance_lens[i])
        CQ_HIP(c, hipMemcpyAsync(B.inst_lag + i * n, instances[i], instance_lens[i] * sizeof(Fr), hipMemcpyHostToDevice, s));
    }
    CQ_TRY(domain_lagrange_to_coeff(dom, B.inst_lag, B.inst_coeff, (uint32_t)I, n, n));
    for (size_t i = 0; i < I; i++)
      for (size_t r = 0; r < instance_lens[i]; r++) tr.common_scalar(Fr::from_limbs64(instances[i] + 4 * r));
  }

  std::vector<Fr> z_tails(S * bf), plk_tails(PL * 2 * (bf + 1)), plkz_tails(PL * bf);
  // The vanishing argument's random polynomial (vanishing/prover.rs:51-55: n field elements = 8n words, then one
  // blind) is the bulk of the RNG stream -- 2^21 words at k = 18, 2^25 (268 MB) at k = 22, more host time than the
  // GPU needs for the advice and round-1 commitments together.  Every draw that precedes it in stream order is made
  // up front by the main thread; the words themselves are drawn by a helper thread, chunk by chunk, each chunk
  // uploaded on the side stream as soon as it is drawn, while the main thread runs rounds 0 and 1.  The main thread
  // does not touch the RNG again before it has joined the helper (just before round 2 commits the polynomial).
  struct RandomPolyDrawer {
    std::thread th;
    bool running = false;
    std::atomic<bool> done{false};
    std::atomic<uint32_t> chunks_done{0};
    uint32_t chunks_total = 1;
    std::chrono::steady_clock::time_point t_start;
    hipError_t err = hipSuccess;
    // estimated time to completion in microseconds (from the progress so far); 0 when done
    double remaining_us() const {
      if (done.load()) return 0.0;
      const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count();
      const uint32_t d = chunks_done.load();
      return d ? el * (double)(chunks_total - d + 1) / d : 1e9;  // +1: the last upload's event
    }
    void start(cq_ctx* c, Rng* rng, uint64_t* pin, uint64_t* dev, size_t words) {
      const size_t chunk = std::max<size_t>(words / 4, (size_t)1 << 18);  // each drawn by up to eight threads, uploaded while the next is drawn
      chunks_total = (uint32_t)((words + chunk - 1) / chunk);
      t_start = std::chrono::steady_clock::now();
      auto work = [=]() {
        hipError_t e = hipSetDevice(c->device);
        for (size_t off = 0; off < words && e == hipSuccess; off += chunk) {
          const size_t cnt = std::min(chunk, words - off);
          rng->fill(pin + off, cnt);
          e = hipMemcpyAsync(dev + off, pin + off, cnt * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream);
          chunks_done.fetch_add(1);
        }
        (void)rng->fr();  // random_blind
        if (e == hipSuccess) e = hipEventRecord(c->copy_done, c->copy_stream);
        err = e;
        done.store(true);
      };
      try {
        th = std::thread(work);
        running = true;
      } catch (...) {  // no thread to be had: draw here (nothing may unwind across the C ABI)
        work();
      }
    }
    int join() {
      if (running) {
        th.join();
        running = false;
      }
      return err == hipSuccess ? 0 : -1;
    }
    ~RandomPolyDrawer() {
      if (running) th.join();
    }
  } drawer;
  // joins the helper, orders the main stream after the uploads, words -> field elements
  auto finish_random_poly = [&]() -> int {
    if (drawer.join() != 0) return c->fail(CQ_ERR_HIP, "random polynomial upload failed");
    CQ_HIP(c, hipStreamWaitEvent(s, c->copy_done, 0));
    return poly_from_u512(c, rng_dev, (uint32_t)n, random_poly);
  };
  // ---- CQ round 1, the part that does not depend on theta (static_lookup/prover.rs:91-107, 122-160): the lookup
  //      inputs on the Lagrange basis and the multiplicities m.  With a single phase and no user challenge the witness
  //      alone determines them, so they are computed right after the advice columns are in place and [m] is committed
  //      in the advice launch -- the round then has no launch of its own (unless it has legacy lookups or an f that is
  //      not a linear combination of advice columns).  The points are WRITTEN where the reference writes them.
  uint32_t *m_counts_ = B.m_counts, *err_dev_ = B.err_dev;
  std::vector<std::array<const Fr*, CQ_MAX_WIDTH>> lk_input(L);
  volatile uint32_t* herr = (volatile uint32_t*)small_v;             // lookup error flag
  Fr* small_fr = (Fr*)((char*)small_v + 64);                         // scalars read back (b(0), z values)
  auto count_multiplicities = [&]() -> int {
    if (!L) return CQ_OK;
    CQ_HIP(c, hipMemsetAsync(m_counts_, 0, (L * N + 16) * sizeof(uint32_t), s));
    size_t input_slot = 0;
    CqRound1Batch r1b;
    r1b.count = 0;
    for (size_t l = 0; l < L; l++) {
      const cq_lookup_desc& lk = pk->lookups[l];
      const uint32_t w = (uint32_t)lk.cols.size();
      // input expressions on the Lagrange basis: `evaluate(expr, n, 1, fixed, advice, instance)` (:91-107)
      for (uint32_t j = 0; j < w; j++, input_slot++) {
        if (lk.prog[j] < 0) {
          lk_input[l][j] = B.adv + (size_t)lk.cols[j] * n;
          continue;
        }
        GateEvalArgs ga;
        ga.prog = pk->lookup_prog + lk.prog[j];
        ga.num_polys = 1;
        ga.constants = pk->constants;
        ga.challenges = B.challenges;
        ga.advice = B.adv;
        ga.fixed = pk->fixed_values;
        ga.instance = B.inst_lag;
        ga.stride = n;
        ga.size = (uint32_t)n;
        ga.rot_scale = 1;
        ga.y = Fr::zero();
        Fr* dst = B.lk_inputs + input_slot * n;
        CQ_TRY(gate_eval(c, ga, dst));
        lk_input[l][j] = dst;
      }
      CqRound1Args& ra = r1b.a[r1b.count];
      ra.width = w;
      for (uint32_t j = 0; j < w; j++) {
        ra.cols[j] = lk_input[l][j];
        ra.values[j] = lk.tables[j]->values;
        ra.slots[j] = lk.tables[j]->slots;
        ra.nslots[j] = lk.tables[j]->nslots;
      }
      r1b.m_counts[r1b.count++] = m_counts_ + l * N;
      if (r1b.count == CQ_ROUND1_BATCH || l + 1 == L) {  // the lookups of a proof share launches
        CQ_TRY(cq_round1(c, r1b, u, err_dev_));
        r1b.count = 0;
      }
    }
    CQ_TRY(cq_m_to_fr(c, m_counts_, (uint32_t)(L * N), B.m_fr));  // the L vectors are adjacent
    *herr = 0;
    CQ_HIP(c, hipMemcpyAsync((void*)herr, err_dev_, 4, hipMemcpyDeviceToHost, s));  // read after the next synchronisation
    return CQ_OK;
  };
  auto lookup_error = [&]() -> int {
    if (*herr == 1) return c->fail(CQ_ERR_LOOKUP, "witness value not in table");
    if (*herr == 2) return c->fail(CQ_ERR_LOOKUP, "Vector lookup must be on the same table row");
    return CQ_OK;
  };
  const bool early_m = L > 0 && pk->num_phases == 1 && pk->challenge_phase.empty();
  std::vector<G1Affine> m_commitments(L, G1Affine::identity());

  // ---- advice, phase by phase (`next_phase`, prover.rs:299-391; the synthesis loop :436-463): copy the phase's
  //      columns in, blind rows u..n (:346-350), one unused blind per column (:352-355), commit, squeeze the phase's
  //      challenges ----------------------------------------------------------------------------------------------
  const size_t NC = pk->challenge_phase.size();
  std::vector<Fr> user_challenges(NC, Fr::zero());
  std::vector<G1Affine> advice_commitments(A, G1Affine::identity());
  auto phase_of = [&](size_t a) -> uint32_t { return pk->advice_phase.empty() ? 0u : pk->advice_phase[a]; };
  for (uint32_t phase = 0; phase < pk->num_phases; phase++) {
    const bool last_phase = phase + 1 == pk->num_phases;
    std::vector<size_t> cols;
    for (size_t a = 0; a < A; a++)
      if (phase_of(a) == phase) cols.push_back(a);
    if (phase > 0) {
      // the caller computes this phase's witness from the challenges of the earlier phases
      std::vector<uint64_t> ch(4 * std::max<size_t>(NC, 1), 0);
      for (size_t i = 0; i < NC; i++) user_challenges[i].to_limbs64(ch.data() + 4 * i);
      CQ_TRY(c->wait(s));
      if (!phase_fn || phase_fn(phase_user, phase, ch.data(), (uint64_t* const*)advice_dev) != 0)
        return c->fail(CQ_ERR_ARG, "create_proof: the phase callback failed");
    }
    const size_t AC = cols.size();
    std::vector<Fr> tails(AC * (n - u));
    for (size_t j = 0; j < AC; j++)
      for (size_t r = 0; r < n - u; r++) tails[j * (n - u) + r] = rng.fr();
    for (size_t j = 0; j < AC; j++) (void)rng.fr();
    // pinned staging: [0, 64 n) the random polynomial's words (filled by the helper thread from the last phase on),
    // behind it this phase's blinding rows -- no part is reused within a proof, so nothing waits for a copy to finish
    Fr* pin_tails = (Fr*)((char*)pin + (size_t)64 * n);
    if (phase > 0) CQ_TRY(c->wait(s));  // the previous phase's upload out of the same bytes has been consumed
    memcpy(pin_tails, tails.data(), tails.size() * sizeof(Fr));
    if (AC) CQ_HIP(c, hipMemcpyAsync(B.tails, pin_tails, tails.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    for (size_t off = 0; off < AC; off += ADVICE_FILL_MAX) {  // rows [0, u) from the caller's columns, [u, n) the blinding rows
      AdviceFillArgs fa;
      fa.count = (uint32_t)std::min<size_t>(ADVICE_FILL_MAX, AC - off);
      for (uint32_t j = 0; j < fa.count; j++) {
        fa.src[j] = (const Fr*)advice_dev[cols[off + j]];
        fa.dst[j] = adv + cols[off + j] * n;
      }
      CQ_TRY(poly_advice_fill(c, fa, B.tails + off * (n - u), (uint32_t)n, u));
    }
    // commit_lagrange per column (:356-360): enqueue now, collect after the host work below
    std::vector<const Fr*> sc(AC);
    std::vector<const G1Affine*> bs(AC, pk->params->g_lagrange);
    std::vector<size_t> ln(AC, n);
    for (size_t j = 0; j < AC; j++) sc[j] = adv + cols[j] * n;
    if (early_m) {  // m_cm (:167-172, as a dense MSM over the table SRS) rides along
      CQ_TRY(count_multiplicities());
      for (size_t l = 0; l < L; l++) { sc.push_back(B.m_fr + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N); }
    }
    Commit adv_cm;
    const uint64_t seq_adv = c->msm_tail_seq;
    if (!sc.empty()) CQ_TRY(adv_cm.begin(pk, sc, bs, ln));
    if (early_adv) {
      // advice -> coefficients (prover.rs:587-603) on the side stream, behind this launch's accumulate kernel: it runs
      // under the launch's tail and the host's round trip for theta, when the GPU has little else to do -- after theta
      // it would compete with the critical path from beta to round 2's launch (round 3: -0.3 ms at k = 18)
      AuxFork fork(c);
      CQ_TRY(fork.begin(seq_adv));
      if (resident) {
        if (adv_hi > adv_lo) CQ_TRY(domain_lagrange_to_coeff(dom, adv + adv_lo * n, adv_poly + adv_lo * n, (uint32_t)(adv_hi - adv_lo), n, n));
      } else {
        CQ_TRY(lagrange_to_coeff_cols(adv, adv_poly, A));
      }
      CQ_TRY(fork.end());
      adv_is_coeff = true;
    }
    if (last_phase) {
    // The next draws from the RNG are, per permutation set, `bf` blinding rows of z and one blind
    // (permutation/prover.rs:169-175), then the vanishing argument's n coefficients + 1 blind
    // (vanishing/prover.rs:51-55): the CQ rounds in between draw nothing, so taking them now keeps the
    // stream order, overlaps the host-side draws with the advice MSMs, and lets the random
    // polynomial's commitment ride along with round 2's launch.
    // legacy lookups, commit_permuted (lookup/prover.rs:491-494, 133-145): bf+1 rows of a', bf+1 rows of s', two blinds
    for (size_t l = 0; l < PL; l++) {
      for (uint32_t r = 0; r < 2 * (bf + 1); r++) plk_tails[l * 2 * (bf + 1) + r] = rng.fr();
      (void)rng.fr();
      (void)rng.fr();
    }
    for (size_t st = 0; st < S; st++) {
      for (uint32_t r = 0; r < bf; r++) z_tails[st * bf + r] = rng.fr();
      (void)rng.fr();  // permutation_product_blind
    }
    // legacy lookups, commit_product (:237, 283): bf rows of z, one blind
    for (size_t l = 0; l < PL; l++) {
      for (uint32_t r = 0; r < bf; r++) plkz_tails[l * bf + r] = rng.fr();
      (void)rng.fr();
    }
    // the random polynomial's words: drawn and uploaded by the helper thread from here on (see RandomPolyDrawer)
    drawer.start(c, &rng, (uint64_t*)pin, rng_dev, 8 * n);
    }
    // batch_normalize (:363-366), write (:370-374)
    mark("advice launch queued");
    if (!sc.empty()) {
      std::vector<G1Affine> pts;
      CQ_TRY(adv_cm.end(pts));
      mark("advice commitments on the host");
      if (early_m) CQ_TRY(lookup_error());
      for (size_t j = 0; j < AC; j++)
        if (!tr.write_point(pts[j])) return c->fail(CQ_ERR_TRANSCRIPT, "advice commitment is the identity");
      for (size_t j = 0; j < AC; j++) advice_commitments[cols[j]] = pts[j];
      if (early_m)
        for (size_t l = 0; l < L; l++) m_commitments[l] = pts[AC + l];
    } else {
      CQ_TRY(c->wait(s));
    }
    for (size_t i = 0; i < NC; i++)  // :383-389
      if (pk->challenge_phase[i] == phase) user_challenges[i] = tr.squeeze();
  }
  if (NC) CQ_HIP(c, hipMemcpyAsync(B.challenges, user_challenges.data(), NC * sizeof(Fr), hipMemcpyHostToDevice, s));
  const Fr theta = tr.squeeze();  // :472
  mark("theta");

  // ---- legacy lookups: commit_permuted (lookup/prover.rs:57-160) ---------------------------------------------
  auto plk_buf = [&](size_t l, int which) { return B.plk + (l * 5 + which) * n; };  // 0 A, 1 S, 2 a', 3 s', 4 z
  auto lagrange_compress = [&](const uint32_t* prog, uint32_t width, const Fr& chal, Fr* dst) {
    GateEvalArgs ga;  // `evaluate(expr, n, 1, ..)` of every expression, folded with theta (:98-117)
    ga.prog = prog;
    ga.num_polys = width;
    ga.constants = pk->constants;
    ga.challenges = B.challenges;
    ga.advice = adv;
    ga.fixed = pk->fixed_values;
    ga.instance = B.inst_lag;
    ga.stride = n;
    ga.size = (uint32_t)n;
    ga.rot_scale = 1;
    ga.y = chal;
    return gate_eval(c, ga, dst);
  };
  if (PL) {
    // permute_expression_pair (lookup/prover.rs:400-502) on the device: canonical values, sorted and matched there
    // (lksort.hip), back to Montgomery form; one status read-back per lookup
    void* stage_v;
    const size_t slot = n * 4;  // u64 words of one array of 2^k canonical values
    CQ_TRY(c->ensure_scratch(7, 3 * slot * sizeof(uint64_t) + lookup_permute_scratch_bytes(k) + 64, &stage_v));
    uint64_t* stage = (uint64_t*)stage_v;
    void* lk_scratch = stage + 3 * slot;
    uint32_t* lk_status = (uint32_t*)((char*)lk_scratch + lookup_permute_scratch_bytes(k));
    uint32_t* lk_status_host = (uint32_t*)((char*)c->pinned_small + 16);  // beside the lookup error flag
    for (size_t l = 0; l < PL; l++) {
      const auto& lk = pk->legacy[l];
      CQ_TRY(lagrange_compress(pk->legacy_prog + lk.in_off, lk.width, theta, plk_buf(l, 0)));
      CQ_TRY(lagrange_compress(pk->legacy_prog + lk.tab_off, lk.width, theta, plk_buf(l, 1)));
      CQ_TRY(fr_to_canonical(c, plk_buf(l, 0), u, stage));
      CQ_TRY(fr_to_canonical(c, plk_buf(l, 1), u, stage + slot));
      CQ_TRY(lookup_permute_dev(c, stage, stage + slot, u, k, stage + 2 * slot, lk_scratch, lk_status));
      CQ_HIP(c, hipMemcpyAsync(lk_status_host, lk_status, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      CQ_TRY(fr_from_canonical(c, stage, u, plk_buf(l, 2)));
      CQ_TRY(fr_from_canonical(c, stage + 2 * slot, u, plk_buf(l, 3)));
      CQ_HIP(c, hipMemcpyAsync(plk_buf(l, 2) + u, plk_tails.data() + l * 2 * (bf + 1), (bf + 1) * sizeof(Fr), hipMemcpyHostToDevice, s));
      CQ_HIP(c, hipMemcpyAsync(plk_buf(l, 3) + u, plk_tails.data() + l * 2 * (bf + 1) + (bf + 1), (bf + 1) * sizeof(Fr), hipMemcpyHostToDevice, s));
      CQ_TRY(c->wait(s));  // the staging arrays are reused by the next lookup
      if (lk_status_host[0] != 0 || lk_status_host[1] != lk_status_host[2])
        return c->fail(CQ_ERR_LOOKUP, "lookup input not in table (Error::ConstraintSystemFailure)");
    }
  }

  // ---- CQ round 1 (static_lookup/prover.rs:51-183) ------------------------------------------------
  if (L && !early_m) {
    CQ_TRY(count_multiplicities());
    CQ_TRY(c->wait(s));
    CQ_TRY(lookup_error());
  }
  // powers of theta for the folds of this proof (theta^0 .. theta^(CQ_MAX_WIDTH - 1))
  Fr theta_pow[CQ_MAX_WIDTH];
  theta_pow[0] = Fr::one();
  for (uint32_t j = 1; j < CQ_MAX_WIDTH; j++) theta_pow[j] = theta_pow[j - 1] * theta;
  for (size_t l0 = 0; l0 < L; l0 += CQ_FOLD_BATCH) {
    // f = sum_j theta^(w-1-j) * e_j   (:108-116, Horner with the first expression first), the lookups of a proof per launch
    CqFoldBatch fb;
    fb.count = (uint32_t)std::min<size_t>(CQ_FOLD_BATCH, L - l0);
    for (uint32_t j = 0; j < CQ_MAX_WIDTH; j++) fb.theta_pow[j] = theta_pow[j];
    for (uint32_t q = 0; q < fb.count; q++) {
      const size_t l = l0 + q;
      const uint32_t w = (uint32_t)pk->lookups[l].cols.size();
      fb.width[q] = w;
      for (uint32_t j = 0; j < w; j++) fb.src[q][j] = lk_input[l][j];
      // resident: f_l exists on the owner of lookup l only
      fb.out[q] = (resident && (l < lk_lo || l >= lk_hi)) ? nullptr : f_lag + l * n;
    }
    CQ_TRY(cq_fold_inputs(c, fb, (uint32_t)n));
  }
  if (L || PL) {
    // permuted input / table of every legacy lookup (lookup/prover.rs:136-151), then f_cm (:165) and, unless it went
    // out with the advice launch, m_cm (:167-172): one launch
    std::vector<const Fr*> sc;
    std::vector<const G1Affine*> bs;
    std::vector<size_t> ln;
    for (size_t l = 0; l < PL; l++)
      for (int which = 2; which <= 3; which++) { sc.push_back(plk_buf(l, which)); bs.push_back(pk->params->g_lagrange); ln.push_back(n); }
    // f_cm: when every input of a lookup is a plain advice column, f = sum_j theta^(w-1-j) e_j over whole columns
    // (blinding rows included), so [f] = sum_j theta^(w-1-j) [e_j] -- w - 1 scalar multiplications of commitments
    // round 0 already produced, done by host threads, instead of an n-term MSM.
    std::vector<int> f_linear(L, 0);
    std::vector<G1Jac> f_host(L);
    for (size_t l = 0; l < L; l++) {
      f_linear[l] = 1;
      for (int64_t pj : pk->lookups[l].prog) f_linear[l] &= pj < 0;
    }
    std::vector<size_t> f_slot(L, 0);
    for (size_t l = 0; l < L; l++)
      if (!f_linear[l]) { f_slot[l] = sc.size(); sc.push_back(f_lag + l * n); bs.push_back(pk->params->g_lagrange); ln.push_back(n); }
    const size_t m_first = sc.size();
    if (!early_m)
      for (size_t l = 0; l < L; l++) { sc.push_back(m_fr + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N); }
    std::vector<G1Affine> cm;
    Commit r1;
    const uint64_t seq = c->msm_tail_seq;
    if (!sc.empty()) CQ_TRY(r1.begin(pk, sc, bs, ln));
    // (on the context's worker threads: the main thread goes on queueing the work that needs theta only)
    std::vector<size_t> f_lin;
    for (size_t l = 0; l < L; l++)
      if (f_linear[l]) f_lin.push_back(l);
    struct PoolJoin {  // an error path must not leave workers with references into this frame
      HostPool& pool;
      HostPool::Ticket t;
      ~PoolJoin() { pool.wait(t); }
    } f_job{c->pool(), nullptr};
    f_job.t = c->pool().submit(f_lin.size(), [&](size_t i) {
      const size_t l = f_lin[i];
      const auto& lcols = pk->lookups[l].cols;
      G1Jac acc = jac_from_affine(advice_commitments[lcols[0]]);
      for (size_t j = 1; j < lcols.size(); j++) acc = jac_add(host_scalar_mul(acc, theta), jac_from_affine(advice_commitments[lcols[j]]));
      f_host[l] = acc;
    });
    {
      // on the side stream (under the launch's tail when there is one): f -> coefficients (:326-334) and onto the
      // extended coset (evaluation.rs:533-548 reads it), the instance cosets -- none of them depends on beta / gamma
      AuxFork fork(c);
      CQ_TRY(fork.begin(seq));
      if (resident) {
        // resident: the owner transforms its lookups' f and its advice columns; nobody else ever reads them
        if (lk_cnt) {
          CQ_TRY(domain_lagrange_to_coeff(dom, f_lag + lk_lo * n, f_coeff + lk_lo * n, (uint32_t)lk_cnt, n, n));
          CQ_TRY(domain_coeff_to_extended(dom, f_coeff + lk_lo * n, cosets + (L + lk_lo) * ext, (uint32_t)lk_cnt, n, ext));
        }
        if (!adv_is_coeff && adv_hi > adv_lo) CQ_TRY(domain_lagrange_to_coeff(dom, adv + adv_lo * n, adv + adv_lo * n, (uint32_t)(adv_hi - adv_lo), n, n));
        adv_is_coeff = true;
      } else {
      if (L) {
        CQ_TRY(lagrange_to_coeff_cols(f_lag, f_coeff, L));
        CQ_TRY(coeff_to_extended_cols(f_coeff, cosets + L * ext, L));
      }
      if (general && I) CQ_TRY(coeff_to_extended_cols(B.inst_coeff, B.inst_cosets, I));
      if (A && S == 0 && !adv_is_coeff) {
        // advice -> coefficients (prover.rs:587-603), in place: without a permutation argument nothing reads the
        // Lagrange values after round 1, and the round-2 inversions leave room for it
        CQ_TRY(lagrange_to_coeff_cols(adv, adv, A));
        adv_is_coeff = true;
      }
      }
      CQ_TRY(fork.end());
    }
    if (!sc.empty()) CQ_TRY(r1.end(cm));
    c->pool().wait(f_job.t);
    for (size_t q = 0; q < 2 * PL; q++)
      if (!tr.write_point(cm[q])) return c->fail(CQ_ERR_TRANSCRIPT, "permuted lookup commitment is the identity");
    for (size_t l = 0; l < L; l++) {
      const G1Affine f_cm = f_linear[l] ? jac_to_affine(f_host[l]) : cm[f_slot[l]];
      if (!tr.write_point(f_cm)) return c->fail(CQ_ERR_TRANSCRIPT, "f commitment is the identity");
      if (!tr.write_point(early_m ? m_commitments[l] : cm[m_first + l])) return c->fail(CQ_ERR_TRANSCRIPT, "m commitment is the identity");
    }
  }
  mark("round 1 written");
  const Fr beta = tr.squeeze();   // prover.rs:529
  const Fr gamma = tr.squeeze();  // :532
  const Fr beta_inv = beta.inv();

  // ---- permutation::Argument::commit (permutation/prover.rs:47-198): Lagrange values of every z ----------
  auto column_values = [&](const std::pair<uint32_t, uint32_t>& col) -> const Fr* {
    return col.first == CQ_COL_ADVICE ? adv + (size_t)col.second * n
         : col.first == CQ_COL_FIXED  ? pk->fixed_values + (size_t)col.second * n
                                      : B.inst_lag + (size_t)col.second * n;
  };
  if (S) {
    const Fr delta = fr_from_raw(FR_DELTA_RAW);
    std::vector<PermProductArgs> pa(S);
    Fr deltaomega = Fr::one();  // delta^(column position), :86-87,146
    for (size_t st = 0; st < S; st++) {
      PermProductArgs& a = pa[st];
      a.count = (uint32_t)std::min(chunk_len, PC - st * chunk_len);
      a.beta = beta;
      a.gamma = gamma;
      a.omega_powers = pk->omega_powers;
      for (uint32_t j = 0; j < a.count; j++) {
        const size_t ci = st * chunk_len + j;
        a.col[j] = column_values(pk->perm_columns[ci]);
        a.sigma[j] = pk->perm_values + ci * n;
        a.delta_beta[j] = deltaomega * beta;
        deltaomega = deltaomega * delta;
      }
      CQ_TRY(perm_denominators(c, a, (uint32_t)n, B.mv + st * n));  // :106-121
    }
    CQ_TRY(poly_batch_invert(c, B.mv, (uint32_t)(S * n)));  // :124
    for (size_t st = 0; st < S; st++) CQ_TRY(perm_numerators(c, pa[st], (uint32_t)n, B.mv + st * n));  // :128-147
    // z[0] = last_z, z[row] = z[row-1] * mv[row-1] (:160-166): a multiplicative scan per set, then the
    // chain last_z = z[n - (bf+1)] of the previous set (:173) as one scalar per set
    CQ_TRY(prefix_product(c, B.mv, B.z, (uint32_t)n, (uint32_t)S));
    std::vector<Fr> at_u(S);
    for (size_t st = 0; st < S; st++)
      CQ_HIP(c, hipMemcpyAsync(&at_u[st], B.z + st * n + u, sizeof(Fr), hipMemcpyDeviceToHost, s));
    CQ_TRY(c->wait(s));
    PermScaleArgs sa;
    Fr last_z = Fr::one();
    for (size_t st = 0; st < S; st++) {
      sa.mult[st] = last_z;
      last_z = last_z * at_u[st];
    }
    CQ_TRY(perm_scale(c, B.z, (uint32_t)n, u + 1, (uint32_t)S, sa));
    // blinding rows (:169-171)
    for (size_t st = 0; st < S; st++)
      CQ_HIP(c, hipMemcpyAsync(B.z + st * n + (n - bf), z_tails.data() + st * bf, bf * sizeof(Fr), hipMemcpyHostToDevice, s));
  }

  // ---- legacy lookups: commit_product (lookup/prover.rs:163-300): z = running product of
  //      (A + beta)(S + gamma) / ((a' + beta)(s' + gamma)), rows n-bf.. random ---------------------------------------
  if (PL) {
    for (size_t l = 0; l < PL; l++) CQ_TRY(lookup_denominators(c, plk_buf(l, 2), plk_buf(l, 3), beta, gamma, (uint32_t)n, plk_buf(l, 4)));
    for (size_t l = 0; l < PL; l++) {
      // the product vectors are not contiguous across lookups (stride 5n): one inversion launch each
      CQ_TRY(poly_batch_invert(c, plk_buf(l, 4), (uint32_t)n));
      CQ_TRY(lookup_numerators(c, plk_buf(l, 0), plk_buf(l, 1), beta, gamma, (uint32_t)n, plk_buf(l, 4)));
      CQ_TRY(prefix_product(c, plk_buf(l, 4), plk_buf(l, 4), (uint32_t)n, 1));
      CQ_HIP(c, hipMemcpyAsync(plk_buf(l, 4) + (n - bf), plkz_tails.data() + l * bf, bf * sizeof(Fr), hipMemcpyHostToDevice, s));
    }
  }

  // ---- CQ round 2 (static_lookup/prover.rs:187-342) -------------------------------------------------
  std::vector<Fr> a_at_zero(L);
  G1Affine random_cm = G1Affine::identity();
  {
    size_t woff = 0;
    for (size_t l0 = 0; l0 < L; l0 += CQ_FOLD_BATCH) {
      // per lookup: t_i = sum_j theta^(w-1-j) T_j[i] (compress_tables :224-240), den_i = m_i ? t_i + beta : 0 (:245-247),
      // B_r = f_r + beta for r < u, beta on the blinding rows (:261-269) -- one launch for the lookups of the proof
      CqFoldBatch fb;
      fb.count = (uint32_t)std::min<size_t>(CQ_FOLD_BATCH, L - l0);
      for (uint32_t j = 0; j < CQ_MAX_WIDTH; j++) fb.theta_pow[j] = theta_pow[j];
      for (uint32_t q = 0; q < fb.count; q++) {
        const size_t l = l0 + q;
        const cq_lookup_desc& lk = pk->lookups[l];
        fb.width[q] = (uint32_t)lk.cols.size();
        for (uint32_t j = 0; j < fb.width[q]; j++) fb.src[q][j] = lk.tables[j]->values;
        fb.out[q] = nullptr;
        const bool mine = !resident || (l >= lk_lo && l < lk_hi);  // resident: b_l on the owner of lookup l only
        fb.f[q] = mine ? f_lag + l * n : nullptr;
        fb.b[q] = bpoly + l * n;
        fb.m[q] = m_counts + l * N;
        fb.den[q] = den + l * N;
      }
      CQ_TRY(cq_round2_prep(c, fb, (uint32_t)n, (uint32_t)N, u, beta));
    }
    if (L && resident) {
      if (lk_cnt) CQ_TRY(poly_batch_invert(c, bpoly + lk_lo * n, (uint32_t)(lk_cnt * n)));
      CQ_TRY(poly_batch_invert(c, den, (uint32_t)(L * N)));  // the table side is small and stays replicated
    } else if (L) {
      // all inversions of the round in two launches (one Fermat inversion per lane dominates the latency)
      CQ_TRY(poly_batch_invert(c, bpoly, (uint32_t)(L * n + L * N)));  // bpoly and den are adjacent
    }
    woff = 0;
    for (size_t l0 = 0; l0 < L; l0 += CQ_FOLD_BATCH) {  // a_i = m_i / (t_i + beta) and its theta-scaled copies for q_a (:247-256)
      CqAValuesBatch ab;
      ab.count = (uint32_t)std::min<size_t>(CQ_FOLD_BATCH, L - l0);
      for (uint32_t j = 0; j < CQ_MAX_WIDTH; j++) ab.theta_pow[j] = theta_pow[j];
      for (uint32_t q = 0; q < ab.count; q++) {
        const size_t l = l0 + q;
        ab.width[q] = (uint32_t)pk->lookups[l].cols.size();
        ab.den_inv[q] = den + l * N;
        ab.m[q] = m_counts + l * N;
        ab.a[q] = a_val + l * N;
        ab.a_scaled[q] = a_scaled + woff * N;
        woff += ab.width[q];
      }
      CQ_TRY(cq_a_values_batch(c, ab, (uint32_t)N));
    }
    if (L && resident) {
      // resident: the owner turns its b_l into coefficients; every rank commits its point range of b_0 = (b - b(0)) / X
      // (:279, 299, 310: the same coefficients over two base arrays), so slice r of b_l[1..n) goes from the owner to rank r
      if (lk_cnt) CQ_TRY(domain_lagrange_to_coeff(dom, bpoly + lk_lo * n, bpoly + lk_lo * n, (uint32_t)lk_cnt, n, n));
      for (size_t l = 0; l < L; l++) small_fr[l] = Fr::zero();
      if (lk_cnt) {
        GatherArgs ga;
        ga.count = (uint32_t)lk_cnt;
        for (size_t l = lk_lo; l < lk_hi; l++) ga.src[l - lk_lo] = bpoly + l * n;
        CQ_TRY(poly_gather_scalars(c, ga, B.gather));
        CQ_HIP(c, hipMemcpyAsync(small_fr + lk_lo, B.gather, lk_cnt * sizeof(Fr), hipMemcpyDeviceToHost, s));  // summed over the ranks below
      }
      std::vector<Xfer> xf;
      for (size_t l = 0; l < L; l++) {
        const uint32_t own = owner_of(l, L);
        for (uint32_t r = 0; r < SW; r++) {
          size_t lo, hi;
          shard_range(n - 1, r, SW, lo, hi);
          Fr* p = bpoly + l * n + 1 + lo;
          if (hi > lo) xf.push_back({p, p, (hi - lo) * sizeof(Fr), own, r});
        }
      }
      CQ_TRY(shard_exchange(pk, xf.data(), xf.size(), s));
    } else if (L) {
      CQ_TRY(lagrange_to_coeff_cols(bpoly, bpoly, L));  // f: under round 1's launch
      // b(0) of every lookup (for a(0), :318-324): on its way to the host while the round's MSMs run
      GatherArgs ga;
      ga.count = (uint32_t)L;
      for (size_t l = 0; l < L; l++) ga.src[l] = bpoly + l * n;
      CQ_TRY(poly_gather_scalars(c, ga, B.gather));
      CQ_HIP(c, hipMemcpyAsync(small_fr, B.gather, L * sizeof(Fr), hipMemcpyDeviceToHost, s));
    }
    // The helper thread has had rounds 0 and 1 and this round's preparation to draw the random polynomial.  From
    // k = 20 on that is not enough (2^25 words take ~25 ms at k = 22): then the polynomial is committed in a launch
    // of its own after the round's other MSMs, which start now.
    // Not done yet?  A launch of its own for the polynomial costs ~0.7 ms of GPU time; waiting costs what is left of
    // the draws.  Wait while the estimate (from the chunks done so far) stays below that, give up otherwise.
    // The choice is timing-dependent, so sharded ranks agree on it first (late on any rank = late on all: a rank that
    // has the polynomial ready just commits it in the second launch too).  CQ_RANDOM_LATE=0/1 pins it (tests).
    bool random_late = false;
    {
      const char* force = getenv("CQ_RANDOM_LATE");
      const auto t0 = std::chrono::steady_clock::now();
      while (!drawer.done.load()) {
        if (force && force[0] == '1') break;
        const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (!(force && force[0] == '0') && waited > 100.0 && drawer.remaining_us() > 700.0) break;
        std::this_thread::yield();
      }
      CQ_TRY(shard_any(pk, (force && force[0] == '1') || !drawer.done.load(), random_late));
    }
    if (!random_late) CQ_TRY(finish_random_poly());
    // commitments, one batch of launches: the permutation products (permutation/prover.rs:177, written first),
    // then a, a0 (dense over the table SRS), q_a (over [qs_0|qs_1|...]), p, b0 (n-1 terms of b[1..]) and the
    // vanishing argument's random polynomial (vanishing/prover.rs:58)
    std::vector<G1Affine> r2;
    {
      std::vector<const Fr*> sc;
      std::vector<const G1Affine*> bs;
      std::vector<size_t> ln;
      for (size_t st = 0; st < S; st++) { sc.push_back(B.z + st * n); bs.push_back(pk->params->g_lagrange); ln.push_back(n); }
      for (size_t l = 0; l < PL; l++) { sc.push_back(plk_buf(l, 4)); bs.push_back(pk->params->g_lagrange); ln.push_back(n); }
      woff = 0;
      for (size_t l = 0; l < L; l++) {
        const uint32_t w = (uint32_t)pk->lookups[l].cols.size();
        sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g1_lagrange); ln.push_back(N);             // a
        sc.push_back(a_scaled + woff * N); bs.push_back(pk->qs_concat[l]); ln.push_back((size_t)w * N);      // q_a
        sc.push_back(a_val + l * N); bs.push_back(pk->table_cfg->g_lagrange_opening_at_0); ln.push_back(N);  // a0
        sc.push_back(bpoly + l * n + 1); bs.push_back(pk->params->g); ln.push_back(n - 1);                   // b0 (:310)
        sc.push_back(bpoly + l * n + 1); bs.push_back(pk->b0_g1_bound); ln.push_back(n - 1);                 // p (:299)
        woff += w;
      }
      if (!random_late) { sc.push_back(random_poly); bs.push_back(pk->params->g); ln.push_back(n); }
      Commit r2cm;
      const uint64_t seq = c->msm_tail_seq;
      CQ_TRY(r2cm.begin(pk, sc, bs, ln));
      {
        // under the launch's tail, none of it depending on y: advice -> coefficients (prover.rs:587-603, in place: the
        // Lagrange values were last read by the permutation / lookup products above) and onto the extended coset
        // (evaluation.rs:317-335), b onto the extended coset
        AuxFork fork(c);
        CQ_TRY(fork.begin(seq));
        if (A && !adv_is_coeff) CQ_TRY(lagrange_to_coeff_cols(adv, adv, A));
        if (L && resident) {  // resident: b_l's coset stays with the owner of lookup l (the quotient is folded there)
          if (lk_cnt) CQ_TRY(domain_coeff_to_extended(dom, bpoly + lk_lo * n, cosets + lk_lo * ext, (uint32_t)lk_cnt, n, ext));
        } else if (L) CQ_TRY(coeff_to_extended_cols(bpoly, cosets, L));
        if (general && A) CQ_TRY(coeff_to_extended_cols(adv, B.adv_cosets, A));
        CQ_TRY(fork.end());
      }
      if (random_late) CQ_TRY(finish_random_poly());  // queued behind the launch above
      mark("round 2 queued");
      CQ_TRY(r2cm.end(r2));
      mark("round 2 commitments on the host");
      if (random_late) {
        std::vector<G1Affine> rc;
        CQ_TRY(commit_batch(pk, {random_poly}, {pk->params->g}, n, rc));
        r2.push_back(rc[0]);
      }
    }
    for (size_t st = 0; st < S; st++)
      if (!tr.write_point(r2[st])) return c->fail(CQ_ERR_TRANSCRIPT, "permutation product commitment is the identity");
    // z -> coefficients (permutation/prover.rs:179), in place
    if (S) CQ_TRY(lagrange_to_coeff_cols(B.z, B.z, S));
    for (size_t l = 0; l < PL; l++) {
      if (!tr.write_point(r2[S + l])) return c->fail(CQ_ERR_TRANSCRIPT, "lookup product commitment is the identity");
      // a', s', z -> coefficients (lookup/prover.rs:133, 285), in place (three consecutive vectors)
      CQ_TRY(domain_lagrange_to_coeff(dom, plk_buf(l, 2), plk_buf(l, 2), 3, n, n));
    }
    for (size_t l = 0; l < L; l++) {
      // write order :306-313: a, q_a, a0, b0, p
      for (size_t q = 0; q < 5; q++)
        if (!tr.write_point(r2[S + PL + 5 * l + q])) return c->fail(CQ_ERR_TRANSCRIPT, "a CQ round-2 commitment is the identity");
    }
    random_cm = r2[S + PL + 5 * L];
    // a(0) = (n*b(0) - (bf+1)/beta) / N   (:318-324)
    if (L) {  // the copy of b(0) was queued before the launch; r2cm.end() has drained the stream
      if (resident) CQ_TRY(shard_sum_scalars(pk, small_fr, L));  // resident: every b_l(0) from its owner
      const Fr n_table_inv = Fr::from_u64(N).inv();
      for (size_t l = 0; l < L; l++)
        a_at_zero[l] = (small_fr[l] * Fr::from_u64(n) - Fr::from_u64(bf + 1) * beta_inv) * n_table_inv;
    }
  }

  // ---- vanishing::Argument::commit (vanishing/prover.rs:37-65): drawn and committed above ---------
  if (!tr.write_point(random_cm)) return c->fail(CQ_ERR_TRANSCRIPT, "random poly commitment is the identity");
  const Fr y = tr.squeeze();  // prover.rs:584
  mark("y");

  // advice polys (lagrange_to_coeff, :587-603) and the cosets of advice / instance / b / f were computed on the side
  // stream under the round-1 and round-2 launches
  CQ_TRY(join_aux(c));

  // ---- evaluate_h (evaluation.rs:285-551) + divide by the vanishing polynomial ------------------------
  {
    if (general) {
      // advice / instance cosets (:317-335), permutation product cosets (permutation/prover.rs:182)
      if (I && !(L || PL)) CQ_TRY(coeff_to_extended_cols(B.inst_coeff, B.inst_cosets, I));  // no round 1
      if (S) CQ_TRY(coeff_to_extended_cols(B.z, B.z_cosets, S));
      const uint32_t rot_scale = 1u << (dom->extended_k - dom->k);
      if (pk->num_gate_polys) {  // custom gates (:348-365)
        GateEvalArgs ga;
        ga.prog = pk->gate_prog;
        ga.num_polys = pk->num_gate_polys;
        ga.constants = pk->constants;
        ga.challenges = B.challenges;
        ga.advice = B.adv_cosets;
        ga.fixed = pk->fixed_cosets;
        ga.instance = B.inst_cosets;
        ga.stride = ext;
        ga.size = (uint32_t)ext;
        ga.rot_scale = rot_scale;
        ga.y = y;
        CQ_TRY(gate_eval(c, ga, h_ext));
      } else {
        CQ_HIP(c, hipMemsetAsync(h_ext, 0, ext * sizeof(Fr), s));
      }
      if (S) {  // permutation constraints (:367-459)
        PermHArgs ph;
        ph.z = B.z_cosets;
        ph.sigma = pk->perm_cosets;
        for (size_t ci = 0; ci < PC; ci++) {
          const auto& col = pk->perm_columns[ci];
          ph.col[ci] = col.first == CQ_COL_ADVICE ? B.adv_cosets + (size_t)col.second * ext
                     : col.first == CQ_COL_FIXED  ? pk->fixed_cosets + (size_t)col.second * ext
                                                  : B.inst_cosets + (size_t)col.second * ext;
        }
        ph.sets = (uint32_t)S;
        ph.chunk_len = (uint32_t)chunk_len;
        ph.ncols = (uint32_t)PC;
        ph.l0 = pk->l0;
        ph.l_last = pk->l_last;
        ph.l_active = pk->l_active_row;
        ph.beta = beta;
        ph.gamma = gamma;
        ph.y = y;
        ph.delta_start = beta * dom->g_coset;  // beta * ZETA (:375)
        ph.ext_pow_lo = pk->ext_pow_lo;
        ph.ext_pow_hi = pk->ext_pow_hi;
        ph.delta = fr_from_raw(FR_DELTA_RAW);
        ph.ext = (uint32_t)ext;
        ph.rot_scale = rot_scale;
        ph.last_rot = bf + 1;
        CQ_TRY(perm_h_terms(c, ph, h_ext));
      }
    }
    for (size_t l = 0; l < PL; l++) {  // legacy lookup constraints (:461-531)
      const auto& lk = pk->legacy[l];
      Fr* co = B.plk_cosets + l * 5 * ext;  // a', s', z cosets, then the compressed input / table on the coset
      CQ_TRY(domain_coeff_to_extended(dom, plk_buf(l, 2), co, 3, n, ext));
      const uint32_t rot_scale = 1u << (dom->extended_k - dom->k);
      for (int side = 0; side < 2; side++) {
        GateEvalArgs ga;
        ga.prog = pk->legacy_prog + (side ? lk.tab_off : lk.in_off);
        ga.num_polys = lk.width;
        ga.constants = pk->constants;
        ga.challenges = B.challenges;
        ga.advice = B.adv_cosets;
        ga.fixed = pk->fixed_cosets;
        ga.instance = B.inst_cosets;
        ga.stride = ext;
        ga.size = (uint32_t)ext;
        ga.rot_scale = rot_scale;
        ga.y = theta;
        CQ_TRY(gate_eval(c, ga, co + (3 + side) * ext));
      }
      LookupHArgs la;
      la.a = co;
      la.s = co + ext;
      la.z = co + 2 * ext;
      la.cin = co + 3 * ext;
      la.ctab = co + 4 * ext;
      la.l0 = pk->l0;
      la.l_last = pk->l_last;
      la.l_active = pk->l_active_row;
      la.beta = beta;
      la.gamma = gamma;
      la.y = y;
      la.ext = (uint32_t)ext;
      la.rot_scale = rot_scale;
      CQ_TRY(lookup_h_terms(c, la, h_ext));
    }
    // CQ terms (:533-548), then the division by X^n - 1 (vanishing/prover.rs:84, domain.rs:319-338)
    CqQuotientArgs qa;
    qa.count = (uint32_t)(resident ? lk_cnt : L);
    for (size_t l = 0; l < qa.count; l++) {
      qa.b[l] = cosets + ((resident ? lk_lo : 0) + l) * ext;
      qa.f[l] = cosets + (L + (resident ? lk_lo : 0) + l) * ext;
    }
    qa.h_in = general ? h_ext : nullptr;
    qa.l_active = pk->l_active_row;
    qa.t_evals = dom->t_evaluations_dev;
    qa.t_len = (uint32_t)dom->t_evaluations.size();
    qa.y = y;
    qa.beta = beta;
    if (resident) {
      // resident: the numerator is a sum over lookups (evaluation.rs:533-548 folds them with powers of y), each owner
      // folds its own range and multiplies by the power of y the lookups after its range contribute
      qa.scale = y.pow_u64(L - lk_hi);
      qa.has_scale = 1;
    }
    if (!resident || lk_cnt) CQ_TRY(poly_cq_quotient(c, qa, (uint32_t)ext, h_ext));
  }
  // vanishing.construct (vanishing/prover.rs:69-120): coefficients, n-sized pieces, blinds, commitments
  const size_t pieces = dom->quotient_poly_degree;
  if (!resident || lk_cnt) CQ_TRY(domain_extended_to_coeff(dom, h_ext, h_coeff));
  if (resident) {
    // resident: division by X^n - 1, extended_to_coeff and the commitments are all linear, so each owner transformed its
    // PARTIAL quotient; rank r now collects slice r of every piece from every owner and adds them up -- it holds the
    // coefficients of h on its own point range only (which is all its share of the commitments, evaluations and the
    // opening reads).  Staging: the buffer reserved at set-up.
    const size_t slice_max = res_slice_max;
    Fr* const stage = (Fr*)res_stage_v;
    std::vector<uint32_t> owners;  // ranks that own a lookup
    for (uint32_t q = 0; q < SW; q++) {
      size_t lo, hi;
      shard_range(L, q, SW, lo, hi);
      if (hi > lo) owners.push_back(q);
    }
    if (owners.size() * pieces * slice_max > res_stage_elems) return c->fail(CQ_ERR_INTERNAL, "resident sharding: staging too small");
    std::vector<Xfer> xf;
    for (size_t oi = 0; oi < owners.size(); oi++)
      for (size_t i = 0; i < pieces; i++)
        for (uint32_t r = 0; r < SW; r++) {
          size_t lo, hi;
          shard_range(n, r, SW, lo, hi);
          if (hi > lo) xf.push_back({h_coeff + i * n + lo, stage + (oi * pieces + i) * slice_max, (hi - lo) * sizeof(Fr), owners[oi], r});
        }
    CQ_TRY(shard_exchange(pk, xf.data(), xf.size(), s));
    size_t lo, hi;
    shard_range(n, me, SW, lo, hi);
    for (size_t i = 0; i < pieces && hi > lo; i++) {
      std::vector<Term> terms;
      for (size_t oi = 0; oi < owners.size(); oi++) terms.push_back({stage + (oi * pieces + i) * slice_max, (uint32_t)(hi - lo), Fr::one()});
      CQ_TRY(lincomb_many(c, terms, Fr::zero(), (uint32_t)(hi - lo), h_coeff + i * n + lo));
    }
  }
  for (size_t i = 0; i < pieces; i++) (void)rng.fr();  // h_blinds (:95-98)
  {
    std::vector<const Fr*> sc(pieces);
    std::vector<const G1Affine*> bs(pieces, pk->params->g);
    for (size_t i = 0; i < pieces; i++) sc[i] = h_coeff + i * n;
    std::vector<G1Affine> o;
    CQ_TRY(commit_batch(pk, sc, bs, n, o));
    for (auto& p : o)
      if (!tr.write_point(p)) return c->fail(CQ_ERR_TRANSCRIPT, "h piece commitment is the identity");
  }
  const Fr x = tr.squeeze();  // prover.rs:629
  mark("x");
  const Fr xn = x.pow_u64(n);

  // ---- evaluations (prover.rs:654-719) and opening queries (:721-773) ------------------------------------
  // Every polynomial opened by the proof, in the order ProverGWC receives the queries; `written` marks the
  // evaluations the transcript carries (h is opened but its value is derived, vanishing/prover.rs:131-153).
  struct Query {
    const Fr* p;
    uint32_t len;
    int32_t rot;
    int h_piece;  // -1, or the index of an h piece (folded into one query with coefficient xn^i)
    Fr eval;
  };
  std::vector<Query> qs;
  auto add_query = [&](const Fr* p, size_t len, int32_t rot) {
    qs.push_back({p, (uint32_t)len, rot, -1, Fr::zero()});
    return qs.size() - 1;
  };
  const int32_t rot_last = -(int32_t)(bf + 1);
  std::vector<size_t> q_advice, q_fixed, q_sigma, q_z, q_z_next, q_z_last(S, (size_t)-1), q_b0, q_f, q_h;
  for (auto& q : pk->advice_queries) q_advice.push_back(add_query(adv_poly + (size_t)q.first * n, n, q.second));
  for (size_t st = 0; st < S; st++) {  // permutation::Evaluated::open (permutation/prover.rs:294-344)
    q_z.push_back(add_query(B.z + st * n, n, 0));
    q_z_next.push_back(add_query(B.z + st * n, n, 1));
  }
  for (size_t st = S > 0 ? S - 1 : 0; st-- > 0;) q_z_last[st] = add_query(B.z + st * n, n, rot_last);
  std::vector<std::array<size_t, 5>> q_plk(PL);  // lookup::Evaluated::open (lookup/prover.rs:343-392): z, a', s' @ x, a' @ x/w, z @ wx
  for (size_t l = 0; l < PL; l++)
    q_plk[l] = {add_query(plk_buf(l, 4), n, 0), add_query(plk_buf(l, 2), n, 0), add_query(plk_buf(l, 3), n, 0),
                add_query(plk_buf(l, 2), n, -1), add_query(plk_buf(l, 4), n, 1)};
  for (size_t l = 0; l < L; l++) {  // static_lookup::Evaluated::open
    q_b0.push_back(add_query(bpoly + l * n + 1, n - 1, 0));  // b0 = (b - b(0))/X
    q_f.push_back(add_query(f_coeff + l * n, n, 0));
  }
  for (auto& q : pk->fixed_queries) q_fixed.push_back(add_query(pk->fixed_polys + (size_t)q.first * n, n, q.second));
  for (size_t ci = 0; ci < PC; ci++) q_sigma.push_back(add_query(pk->perm_polys + ci * n, n, 0));  // ProvingKey::open (:215-225)
  for (size_t i = 0; i < pieces; i++) {
    q_h.push_back(add_query(h_coeff + i * n, n, 0));
    qs.back().h_piece = (int)i;
  }
  const size_t q_random = add_query(random_poly, n, 0);
  // distinct points in first-seen order (gwc.rs:36-61); rotate_omega (domain.rs:414-424)
  std::vector<int32_t> rots;
  for (auto& q : qs)
    if (std::find(rots.begin(), rots.end(), q.rot) == rots.end()) rots.push_back(q.rot);
  auto point_of = [&](int32_t rot) { return rot >= 0 ? x * dom->omega.pow_u64((uint64_t)rot) : x * dom->omega_inv.pow_u64((uint64_t)(-(int64_t)rot)); };
  // resident: who holds what -- an advice polynomial on the owner of its column, b_0 / f of a lookup on its owner (whole
  // polynomials, evaluated there), the h pieces and the random polynomial on every rank by point range (each rank
  // evaluates its range and multiplies by x^lo).  qowner[i] = owning rank, or -1 for "by point range".
  std::vector<int> qowner(qs.size(), -1);
  size_t my_lo = 0, my_hi = n;  // this rank's point range of a length-n vector
  if (resident) {
    shard_range(n, me, SW, my_lo, my_hi);
    for (size_t j = 0; j < q_advice.size(); j++) qowner[q_advice[j]] = (int)owner_of(pk->advice_queries[j].first, A);
    for (size_t l = 0; l < L; l++) qowner[q_b0[l]] = qowner[q_f[l]] = (int)owner_of(l, L);
    std::vector<const Fr*> ps;
    std::vector<uint32_t> ls;
    std::vector<size_t> idx;
    for (size_t i = 0; i < qs.size(); i++) {
      if (qowner[i] >= 0 && qowner[i] != (int)me) continue;
      const bool whole = qowner[i] >= 0;
      if (!whole && my_hi <= my_lo) continue;
      ps.push_back(whole ? qs[i].p : qs[i].p + my_lo);
      ls.push_back(whole ? qs[i].len : (uint32_t)(my_hi - my_lo));
      idx.push_back(i);
    }
    std::vector<Fr> ev(ps.size()), contrib(qs.size(), Fr::zero());
    if (!ps.empty()) CQ_TRY(eval_many(c, ps, ls, x, ev.data()));
    const Fr x_lo = x.pow_u64(my_lo);
    for (size_t j = 0; j < idx.size(); j++) contrib[idx[j]] = qowner[idx[j]] >= 0 ? ev[j] : ev[j] * x_lo;
    CQ_TRY(shard_sum_scalars(pk, contrib.data(), contrib.size()));  // one all-gather: every evaluation is a sum over the ranks
    for (size_t i = 0; i < qs.size(); i++) qs[i].eval = contrib[i];
  } else
  for (int32_t rot : rots) {
    std::vector<const Fr*> ps;
    std::vector<uint32_t> ls;
    std::vector<size_t> idx;
    for (size_t i = 0; i < qs.size(); i++)
      if (qs[i].rot == rot) {
        ps.push_back(qs[i].p);
        ls.push_back(qs[i].len);
        idx.push_back(i);
      }
    std::vector<Fr> ev(ps.size());
    CQ_TRY(eval_many(c, ps, ls, point_of(rot), ev.data()));
    for (size_t j = 0; j < idx.size(); j++) qs[idx[j]].eval = ev[j];
  }
  for (size_t i : q_advice) tr.write_scalar(qs[i].eval);  // :654-672
  for (size_t i : q_fixed) tr.write_scalar(qs[i].eval);   // :674-687
  tr.write_scalar(qs[q_random].eval);                     // vanishing/prover.rs:145-146
  for (size_t i : q_sigma) tr.write_scalar(qs[i].eval);   // permutation/prover.rs:227-239
  for (size_t st = 0; st < S; st++) {                     // :243-290
    tr.write_scalar(qs[q_z[st]].eval);
    tr.write_scalar(qs[q_z_next[st]].eval);
    if (st + 1 < S) tr.write_scalar(qs[q_z_last[st]].eval);
  }
  for (size_t l = 0; l < PL; l++)  // lookup::Committed::evaluate (lookup/prover.rs:303-340): z, z(wx), a', a'(x/w), s'
    for (size_t which : {(size_t)0, (size_t)4, (size_t)1, (size_t)3, (size_t)2}) tr.write_scalar(qs[q_plk[l][which]].eval);
  for (size_t l = 0; l < L; l++) {  // static_lookup/prover.rs:360-370
    tr.write_scalar(qs[q_b0[l]].eval);
    tr.write_scalar(qs[q_f[l]].eval);
    tr.write_scalar(a_at_zero[l]);
  }

  // ---- multiopen, SHPLONK (shplonk/prover.rs:120-286): two commitments whatever the number of points ---------
  if (pk->opener == CQ_OPENER_SHPLONK) {
    Fr* sh_h = B.shplonk;           // h(X) = sum_i xn^i h_i as one polynomial (vanishing/prover.rs:131-135)
    Fr* sh_div[2] = {B.shplonk + n, B.shplonk + 2 * n};
    Fr* sh_hx = B.shplonk + 3 * n;
    Fr* sh_lx = B.shplonk + 4 * n;
    Fr* sh_rem = B.shplonk + 5 * n;  // per rotation set: sum_j y^j r_j (at most 8 coefficients)
    Fr h_eval = Fr::zero();
    {
      std::vector<Term> terms;
      Fr xp = Fr::one();
      for (size_t i = 0; i < pieces; i++) {
        terms.push_back({h_coeff + i * n, (uint32_t)n, xp});
        xp = xp * xn;
      }
      CQ_TRY(lincomb_many(c, terms, Fr::zero(), (uint32_t)n, sh_h));
      for (size_t i = pieces; i-- > 0;) h_eval = h_eval * xn + qs[q_h[i]].eval;
    }
    const Fr y_ = tr.squeeze();  // :136
    // construct_intermediate_sets (shplonk.rs:56-133): commitments by polynomial identity, grouped by point set
    struct Com { const Fr* p; uint32_t len; std::vector<int32_t> rots; std::vector<Fr> evals; };
    std::vector<Com> coms;
    for (auto& q : qs) {
      const Fr* key = q.h_piece >= 0 ? sh_h : q.p;
      if (q.h_piece > 0) continue;
      const Fr ev = q.h_piece == 0 ? h_eval : q.eval;
      auto it = std::find_if(coms.begin(), coms.end(), [&](const Com& cm) { return cm.p == key; });
      if (it == coms.end()) {
        coms.push_back({key, q.h_piece == 0 ? (uint32_t)n : q.len, {q.rot}, {ev}});
      } else if (std::find(it->rots.begin(), it->rots.end(), q.rot) == it->rots.end()) {
        it->rots.push_back(q.rot);
        it->evals.push_back(ev);
      }
    }
    struct RotSet { std::vector<int32_t> rots; std::vector<size_t> members; };
    std::vector<RotSet> sets;
    auto same_set = [](const std::vector<int32_t>& a, const std::vector<int32_t>& b) {
      return a.size() == b.size() && std::all_of(a.begin(), a.end(), [&](int32_t r) { return std::find(b.begin(), b.end(), r) != b.end(); });
    };
    for (size_t ci = 0; ci < coms.size(); ci++) {
      auto it = std::find_if(sets.begin(), sets.end(), [&](const RotSet& rs) { return same_set(rs.rots, coms[ci].rots); });
      if (it == sets.end()) sets.push_back({coms[ci].rots, {ci}});
      else it->members.push_back(ci);
    }
    if (sets.size() > 64) return c->fail(CQ_ERR_ARG, "too many rotation sets");
    const Fr v_ = tr.squeeze();  // :196
    // lagrange_interpolate (arithmetic.rs:425-478) of a commitment's evaluations over its set's points
    auto interpolate = [&](const std::vector<Fr>& pts, const std::vector<Fr>& evals) {
      const size_t m = pts.size();
      if (m == 1) return std::vector<Fr>{evals[0]};
      std::vector<Fr> fin(m, Fr::zero());
      for (size_t j = 0; j < m; j++) {
        std::vector<Fr> tmp{Fr::one()};
        for (size_t kk = 0; kk < m; kk++) {
          if (kk == j) continue;
          const Fr denom = (pts[j] - pts[kk]).inv();
          std::vector<Fr> nxt(tmp.size() + 1, Fr::zero());
          for (size_t i = 0; i <= tmp.size(); i++) {
            const Fr a_ = i < tmp.size() ? tmp[i] : Fr::zero();
            const Fr b_ = i > 0 ? tmp[i - 1] : Fr::zero();
            nxt[i] = a_ * (Fr::zero() - denom * pts[kk]) + b_ * denom;
          }
          tmp.swap(nxt);
        }
        for (size_t i = 0; i < m; i++) fin[i] = fin[i] + tmp[i] * evals[j];
      }
      return fin;
    };
    auto eval_small = [](const std::vector<Fr>& poly, const Fr& at) {
      Fr acc = Fr::zero();
      for (size_t i = poly.size(); i-- > 0;) acc = acc * at + poly[i];
      return acc;
    };
    // per set: low-degree equivalents r_j and the y-combined remainder (CommitmentExtension, :36-76)
    std::vector<std::vector<Fr>> set_points(sets.size());
    std::vector<std::vector<std::vector<Fr>>> low(sets.size());
    std::vector<Fr> rem_host(sets.size() * 8, Fr::zero());
    for (size_t si = 0; si < sets.size(); si++) {
      for (int32_t r : sets[si].rots) set_points[si].push_back(point_of(r));
      if (set_points[si].size() > 8) return c->fail(CQ_ERR_ARG, "rotation set too large");
      Fr py = Fr::one();
      for (size_t ci : sets[si].members) {
        // the commitment's evaluations, aligned with the set's point order
        std::vector<Fr> evals;
        for (int32_t r : sets[si].rots) {
          const size_t pos = std::find(coms[ci].rots.begin(), coms[ci].rots.end(), r) - coms[ci].rots.begin();
          evals.push_back(coms[ci].evals[pos]);
        }
        low[si].push_back(interpolate(set_points[si], evals));
        for (size_t i = 0; i < low[si].back().size(); i++) rem_host[si * 8 + i] = rem_host[si * 8 + i] + py * low[si].back()[i];
        py = py * y_;
      }
    }
    CQ_HIP(c, hipMemcpyAsync(sh_rem, rem_host.data(), rem_host.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    // h_x = sum_i v^i * (sum_j y^j (p_j - r_j)) / prod_{pt in set_i} (X - pt)   (quotient_contribution, :138-168)
    Fr pv = Fr::one();
    for (size_t si = 0; si < sets.size(); si++) {
      std::vector<Term> terms;
      Fr py = Fr::one();
      for (size_t ci : sets[si].members) {
        terms.push_back({coms[ci].p, coms[ci].len, py});
        py = py * y_;
      }
      terms.push_back({sh_rem + si * 8, (uint32_t)set_points[si].size(), Fr::zero() - Fr::one()});
      CQ_TRY(lincomb_many(c, terms, Fr::zero(), (uint32_t)n, sh_div[0]));
      uint32_t len = (uint32_t)n;
      int cur = 0;
      for (const Fr& pt : set_points[si]) {  // div_by_vanishing (:28-34)
        CQ_TRY(poly_kate_division(c, sh_div[cur], len, pt, sh_div[cur ^ 1]));
        cur ^= 1;
        len--;
      }
      std::vector<Term> acc;
      if (si) acc.push_back({sh_hx, (uint32_t)n, Fr::one()});
      acc.push_back({sh_div[cur], len, pv});
      CQ_TRY(lincomb_many(c, acc, Fr::zero(), (uint32_t)n, sh_hx));
      pv = pv * v_;
    }
    {
      std::vector<const Fr*> sc{sh_hx};
      std::vector<const G1Affine*> bs{pk->params->g};
      std::vector<G1Affine> o;
      CQ_TRY(commit_batch(pk, sc, bs, n, o));  // :204
      if (!tr.write_point(o[0])) return c->fail(CQ_ERR_TRANSCRIPT, "shplonk quotient commitment is the identity");
    }
    const Fr u_ = tr.squeeze();  // :206
    // l_x = sum_i v^i z_i sum_j y^j (p_j - r_j(u)) - Z_T(u) h_x, then / (X - u) and / z_0   (:208-283); the
    // scaling by 1/z_0 is folded into the coefficients (the division is linear)
    std::vector<Fr> all_points;
    for (int32_t r : rots) all_points.push_back(point_of(r));  // super_point_set: every distinct point
    auto vanish = [&](const std::vector<Fr>& roots) {
      Fr acc = Fr::one();
      for (auto& r : roots) acc = (u_ - r) * acc;
      return acc;
    };
    std::vector<Fr> z_diff(sets.size());
    for (size_t si = 0; si < sets.size(); si++) {
      std::vector<Fr> diffs;
      for (size_t pi = 0; pi < rots.size(); pi++)
        if (std::find(sets[si].rots.begin(), sets[si].rots.end(), rots[pi]) == sets[si].rots.end()) diffs.push_back(all_points[pi]);
      z_diff[si] = vanish(diffs);
    }
    const Fr z0_inv = z_diff[0].inv();
    const Fr zt_eval = vanish(all_points);
    std::vector<Term> terms;
    Fr sub = Fr::zero();
    pv = Fr::one();
    for (size_t si = 0; si < sets.size(); si++) {
      const Fr scale = pv * z_diff[si] * z0_inv;
      Fr py = Fr::one();
      for (size_t j = 0; j < sets[si].members.size(); j++) {
        const Com& cm = coms[sets[si].members[j]];
        terms.push_back({cm.p, cm.len, scale * py});
        sub = sub + scale * py * eval_small(low[si][j], u_);
        py = py * y_;
      }
      pv = pv * v_;
    }
    terms.push_back({sh_hx, (uint32_t)n, Fr::zero() - zt_eval * z0_inv});
    CQ_TRY(lincomb_many(c, terms, sub, (uint32_t)n, sh_lx));
    CQ_TRY(poly_kate_division(c, sh_lx, (uint32_t)n, u_, sh_div[0]));  // :257
    std::vector<const Fr*> sc{sh_div[0]};
    std::vector<const G1Affine*> bs{pk->params->g};
    std::vector<G1Affine> o;
    CQ_TRY(commit_batch(pk, sc, bs, n - 1, o));  // :270
    if (!tr.write_point(o[0])) return c->fail(CQ_ERR_TRANSCRIPT, "shplonk opening commitment is the identity");
    proof_out.swap(tr.proof);
    return CQ_OK;
  }

  // ---- multiopen, GWC (gwc/prover.rs:42-91): one witness polynomial per distinct point ---------------------
  {
    const Fr v = tr.squeeze();
    // h(X) = sum_i xn^i h_i (vanishing/prover.rs:131-135); its value follows from the piece evaluations
    Fr h_eval = Fr::zero();
    for (size_t i = pieces; i-- > 0;) h_eval = h_eval * xn + qs[q_h[i]].eval;
    std::vector<const Fr*> sc;
    for (size_t g = 0; g < rots.size(); g++) {
      std::vector<Term> terms;
      Fr pv = Fr::one(), eval_batch = Fr::zero(), xp = Fr::one();
      for (auto& q : qs) {
        if (q.rot != rots[g]) continue;
        if (q.h_piece >= 0) {  // the pieces of h share one power of v
          terms.push_back({q.p, q.len, pv * xp});
          xp = xp * xn;
          if ((size_t)q.h_piece + 1 < pieces) continue;
          eval_batch = eval_batch + h_eval * pv;
        } else {
          terms.push_back({q.p, q.len, pv});
          eval_batch = eval_batch + q.eval * pv;
        }
        pv = pv * v;
      }
      Fr* batch = B.gwc_batch + g * n;
      Fr* wit = B.gwc_wit + g * n;
      if (resident) {
        // resident: poly_batch = sum_j v^j p_j is assembled where the p_j live -- whole polynomials on their owner, the
        // by-range ones on each rank's range -- as a partial sum P_r over all n coefficients; rank r then collects the
        // slice of every P_q that its share of the witness polynomial needs and adds them up.  kate_division
        // (q_i = a_(i+1) + z q_(i+1)) runs per range: the carry into a range from above is the Horner value of the
        // slices above it, S_q = sum_t slice_q[t] z^t, exchanged as one field element per rank.
        const Fr z = point_of(rots[g]);
        std::vector<Term> whole, ranged;
        size_t ti = 0;
        for (size_t i = 0; i < qs.size(); i++) {
          if (qs[i].rot != rots[g]) continue;
          const Term& t = terms[ti++];
          if (qowner[i] < 0) ranged.push_back({t.p + my_lo, (uint32_t)(my_hi - my_lo), t.coeff});
          else if (qowner[i] == (int)me) whole.push_back(t);
        }
        if (whole.empty()) CQ_HIP(c, hipMemsetAsync(batch, 0, n * sizeof(Fr), s));
        else CQ_TRY(lincomb_many(c, whole, Fr::zero(), (uint32_t)n, batch));
        if (my_hi > my_lo) {  // the by-range terms on this rank's range; "- eval_batch" touches coefficient 0 only
          std::vector<Term> tt{{batch + my_lo, (uint32_t)(my_hi - my_lo), Fr::one()}};
          tt.insert(tt.end(), ranged.begin(), ranged.end());
          CQ_TRY(lincomb_many(c, tt, my_lo == 0 ? eval_batch : Fr::zero(), (uint32_t)(my_hi - my_lo), batch + my_lo));
        }
        // rank r commits witness coefficients [wlo, whi) (its point range of the n - 1 term MSM) and needs a[wlo+1 .. whi]
        const size_t slice_max = res_slice_max;
        Fr* const stage = (Fr*)res_stage_v;
        std::vector<Xfer> xf;
        for (uint32_t q = 0; q < SW; q++)
          for (uint32_t r = 0; r < SW; r++) {
            size_t wlo, whi;
            shard_range(n - 1, r, SW, wlo, whi);
            if (whi > wlo) xf.push_back({batch + 1 + wlo, stage + (size_t)q * slice_max, (whi - wlo) * sizeof(Fr), q, r});
          }
        CQ_TRY(shard_exchange(pk, xf.data(), xf.size(), s));
        size_t wlo, whi;
        shard_range(n - 1, me, SW, wlo, whi);
        const size_t len = whi - wlo;
        // [0] unused, [1 .. len] the summed slice, [len + 1] the carry -- behind the received slices in the staging area
        Fr* a_loc = stage + (size_t)SW * slice_max + 8;
        if ((size_t)SW * slice_max + 8 + len + 2 > res_stage_elems) return c->fail(CQ_ERR_INTERNAL, "resident sharding: staging too small");
        Fr slice_eval = Fr::zero();
        if (len) {
          std::vector<Term> tt;
          for (uint32_t q = 0; q < SW; q++) tt.push_back({stage + (size_t)q * slice_max, (uint32_t)len, Fr::one()});
          CQ_TRY(lincomb_many(c, tt, Fr::zero(), (uint32_t)len, a_loc + 1));
          const Fr* pp = a_loc + 1;
          const uint32_t ll = (uint32_t)len;
          CQ_TRY(poly_eval_batch(c, &pp, &ll, 1, z, &slice_eval));
        }
        std::vector<Fr> all_s(SW);
        CQ_TRY(shard_allgather_host(pk, &slice_eval, all_s.data(), sizeof(Fr)));
        Fr carry = Fr::zero();  // q_(whi) = S_(r+1) + z^len_(r+1) (S_(r+2) + ...)
        for (uint32_t q = SW; q-- > me + 1;) {
          size_t qlo, qhi;
          shard_range(n - 1, q, SW, qlo, qhi);
          carry = all_s[q] + z.pow_u64(qhi - qlo) * carry;
        }
        if (len) {
          Fr edge[2] = {Fr::zero(), carry};
          CQ_HIP(c, hipMemcpyAsync(a_loc, &edge[0], sizeof(Fr), hipMemcpyHostToDevice, s));
          CQ_HIP(c, hipMemcpyAsync(a_loc + len + 1, &edge[1], sizeof(Fr), hipMemcpyHostToDevice, s));
          CQ_TRY(c->wait(s));  // `edge` is on this frame
          CQ_TRY(poly_kate_division(c, a_loc, (uint32_t)(len + 2), z, wit + wlo));  // wit[wlo .. whi) (and the carry at [whi])
        }
        sc.push_back(wit);
        continue;
      }
      CQ_TRY(lincomb_many(c, terms, eval_batch, (uint32_t)n, batch));  // poly_batch - eval_batch (gwc/prover.rs:62-78)
      CQ_TRY(poly_kate_division(c, batch, (uint32_t)n, point_of(rots[g]), wit));  // :80
      sc.push_back(wit);
    }
    std::vector<const G1Affine*> bs(sc.size(), pk->params->g);
    std::vector<G1Affine> o;
    CQ_TRY(commit_batch(pk, sc, bs, n - 1, o));  // :85
    for (auto& w : o)
      if (!tr.write_point(w)) return c->fail(CQ_ERR_TRANSCRIPT, "opening witness commitment is the identity");
  }
  mark("done");
  proof_out.swap(tr.proof);
  return CQ_OK;
}

}  // namespace cq
