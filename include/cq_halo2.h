/*
 * cq_halo2.h -- C ABI of the MI355X (gfx950) backend for the CQ-lookup / KZG
 * polynomial-commitment hot path of halo2_proofs::plonk::create_proof.
 *
 * The reference (aleph-zero-foundation/sha2-on-cq-halo2) is 100 % Rust and has no FFI
 * seam; each entry point below replaces one Rust function that `create_proof` and its
 * sub-arguments call (cited as file:line under halo2_proofs/src unless noted).  The Rust
 * binding a maintainer would add is sketched in INTEGRATION.md.
 *
 * Conventions
 *  - Field elements: 4 x uint64_t little-endian limbs of a*2^256 mod p (Montgomery form),
 *    byte-identical to halo2curves `Fr([u64;4])` / `Fq([u64;4])` (arithmetic/curves/src/bn256/fr.rs:25).
 *  - G1 affine: 8 x uint64_t = x || y, identity = all zero (derive/curve.rs:453-463).
 *  - G1 Jacobian: 12 x uint64_t = x || y || z, identity z = 0 (derive/curve.rs:696-705).
 *  - Every function returns CQ_OK (0) or a negative status; nothing unwinds or aborts across
 *    the boundary (the Rust side panics on these contract violations: arithmetic.rs:133,184).
 *  - `_dev` entry points take DEVICE pointers (hipMalloc / cq_dev_alloc / a torch tensor's
 *    data_ptr) and run asynchronously on the context's stream; the others take host slices,
 *    like the Rust functions they replace, and return after the result is in host memory.
 *  - A context is bound to one GPU and one HIP stream and may be used by one thread at a
 *    time (create_proof calls these serially, plonk/prover.rs:51-779); use one context per
 *    thread / per rank for concurrency.
 */
#ifndef CQ_HALO2_H
#define CQ_HALO2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CQ_OK 0
#define CQ_ERR_ARG (-1)      /* bad argument (length mismatch, not a power of two, null pointer) */
#define CQ_ERR_HIP (-2)      /* HIP runtime failure; see cq_last_error */
#define CQ_ERR_NO_DEVICE (-3)
#define CQ_ERR_LOOKUP (-4)   /* CQ: witness value not in table / vector lookup on different rows */
#define CQ_ERR_INTERNAL (-5)
#define CQ_ERR_TRANSCRIPT (-6) /* a commitment was the identity (transcript.rs:221-227) */

typedef struct cq_ctx cq_ctx;
typedef struct cq_domain cq_domain;   /* EvaluationDomain<Fr>, poly/domain.rs:19-34 */
typedef struct cq_params cq_params;   /* ParamsKZG<Bn256> G1 part, poly/kzg/commitment.rs:31-39 */
typedef struct cq_table_config cq_table_config; /* StaticTableConfig, plonk/static_lookup.rs:47-66 */
typedef struct cq_static_table cq_static_table; /* StaticTableValues, plonk/static_lookup.rs:68-75 */
typedef struct cq_pk cq_pk;           /* ProvingKey slice read by the CQ-only prover, plonk.rs:291-308 */
/* `R: RngCore` of create_proof (plonk/prover.rs:65): the library calls next_u64 exactly as
 * `Fr::random` does (8 calls per scalar, low limb first, bn256/fr.rs:159-170). */
typedef uint64_t (*cq_rng_next_u64)(void* state);
/* Optional bulk form of the same generator (`RngCore::fill_bytes`): writes the next `count` outputs of next_u64 to
 * dst and advances the state.  See cq_pk_set_rng_fill. */
typedef void (*cq_rng_fill_fn)(void* state, uint64_t* dst, size_t count);
/* Collective hook for MSM sharding (one process per GPU): gathers `bytes_per_rank` bytes from every rank
 * into `recv` (world x bytes_per_rank, rank order).  The host application implements it with RCCL /
 * torch.distributed; returns 0 on success. */
typedef int (*cq_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
/* Broadcast hook for column sharding: `buf` (host memory, `bytes` long) holds the payload on rank `root` and receives it
 * on the others; returns 0 on success. */
typedef int (*cq_bcast_fn)(void* user, void* buf, size_t bytes, uint32_t root);

/* Point-to-point hook for resident column sharding (cq_pk_set_resident_sharding): `count` transfers of HOST buffers
 * between this rank and `peer`, `send` != 0 for outgoing ones.  Every rank is called with its part of one global list, in
 * the same relative order, so the k-th send from a to b meets the k-th receive of b from a.  Returns 0 on success. */
typedef struct { uint32_t peer; uint32_t send; void* buf; size_t bytes; } cq_xfer;
typedef int (*cq_exchange_fn)(void* user, const cq_xfer* xfers, size_t count);

/* ---- context ------------------------------------------------------------------------- */
/* `hip_stream` may be NULL (the library creates its own stream) or an existing hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream) that all work is then enqueued on. */
int cq_ctx_create(int device, void* hip_stream, cq_ctx** out);
void cq_ctx_destroy(cq_ctx* ctx);
const char* cq_last_error(const cq_ctx* ctx);
int cq_ctx_sync(cq_ctx* ctx);
/* hipGraph replay of the MSM pipeline (BASELINE configs[4]: "hipGraph-captured rounds"): with `on` != 0 the kernel
 * sequences before and after the accumulate kernel of every MSM launch (sort + plan: ~12 launches; combine + bucket
 * reduction: ~6) are captured once per launch shape and replayed with one hipGraphLaunch each.  Same results; off by
 * default because on ROCm 7.2 it measured no faster than the plain launches.  This is SEGMENT capture: a whole
 * Fiat-Shamir round cannot be one graph here, because a round ends in a host step the next round's kernels depend on --
 * the commitments are normalised and absorbed by the Blake2b transcript on the host and the challenge comes back as a
 * kernel argument -- and the measured idle time between rounds is that host step (fold, normalise, hash: 60-100 us per
 * round, profiles/r03_host_trace_k18.txt), not launch overhead. */
int cq_ctx_set_hip_graphs(cq_ctx* ctx, int on);
/* RCCL communicator of the context (one process per GPU, xGMI within a node): rank 0 draws an id with
 * cq_rccl_unique_id and hands it to the other ranks by whatever means the application has (a torch.distributed
 * broadcast, MPI, a file); every rank then calls cq_ctx_comm_init_rccl -- collectively, it blocks until all have.  The
 * library loads librccl at run time (the copy already in the process, e.g. PyTorch's, else ROCm's) and issues its
 * collectives on the context's stream, on device buffers.  CQ_ERR_NO_DEVICE when librccl cannot be loaded. */
#define CQ_RCCL_UNIQUE_ID_BYTES 128
int cq_rccl_unique_id(uint8_t id[CQ_RCCL_UNIQUE_ID_BYTES]);
int cq_ctx_comm_init_rccl(cq_ctx* ctx, uint32_t rank, uint32_t world, const uint8_t id[CQ_RCCL_UNIQUE_ID_BYTES]);
int cq_ctx_comm_destroy(cq_ctx* ctx);
/* Collective self-check of the communicator, issued the way the sharded prover issues its exchanges: an ncclAllGather
 * on the context's stream; a grouped launch of ncclBroadcasts (one root per rank, unequal lengths of about 1 MiB) on the
 * context's SIDE stream, ordered behind an event of the main stream; a second ncclAllGather on the main stream while the
 * broadcasts are in flight; device buffers, patterns every rank verifies.  Every rank calls it.
 *
 * Failure behaviour of the RCCL transport: the staging its small exchanges need is allocated by cq_ctx_comm_init_rccl, not
 * inside proofs; a sharded proof first agrees that every rank could set itself up (one status word per rank) and returns an
 * error everywhere if one could not; host waits inside a proof poll the stream with a time-out (CQ_COMM_TIMEOUT_S, default
 * 300 s, 0 = none) and ncclCommGetAsyncError; a rank that fails alone in the middle of a proof (CQ_ERR_HIP / _INTERNAL)
 * aborts its communicator (ncclCommAbort) so that its peers fail -- asynchronous error or time-out -- instead of hanging.
 * After an abort the context needs a new communicator (cq_ctx_comm_init_rccl). */
int cq_ctx_comm_selftest(cq_ctx* ctx);
void* cq_ctx_stream(cq_ctx* ctx);
const char* cq_version(void);

/* ---- device memory (plumbing) ------------------------------------------------------------ */
int cq_dev_alloc(cq_ctx* ctx, size_t bytes, void** dptr);
int cq_dev_free(cq_ctx* ctx, void* dptr);
int cq_dev_upload(cq_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int cq_dev_download(cq_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int cq_dev_memset(cq_ctx* ctx, void* dptr, int value, size_t bytes);

/* ---- arithmetic.rs ------------------------------------------------------------------- */
/* best_fft(a, omega, log_n)  arithmetic.rs:171-234.  In place on a host slice of 2^log_n Fr,
 * natural order in and out. */
int cq_best_fft(cq_ctx* ctx, uint64_t* a, uint32_t log_n, const uint64_t omega[4]);
/* Same on device memory; both 2^log_n elements; `out` may alias `in`. */
int cq_best_fft_dev(cq_ctx* ctx, const uint64_t* in_dev, uint64_t* out_dev, uint32_t log_n,
                    const uint64_t omega[4]);

/* best_multiexp(coeffs, bases)  arithmetic.rs:132-159.  sum_i coeffs[i] * bases[i] as a Jacobian
 * point (the caller normalises, e.g. batch_normalize at plonk/prover.rs:363).  `len` is the common
 * length of both slices (the Rust function asserts equality, arithmetic.rs:133). */
int cq_best_multiexp(cq_ctx* ctx, const uint64_t* coeffs, const uint64_t* bases, size_t len,
                     uint64_t out_jac[12]);
/* Same with scalars and bases resident in device memory; the result is written to HOST memory
 * (96 bytes) after the stream has drained. */
int cq_best_multiexp_dev(cq_ctx* ctx, const uint64_t* coeffs_dev, const uint64_t* bases_dev, size_t len,
                         uint64_t out_jac[12]);
/* `count` multiexps over the same bases (e.g. all advice columns of a phase,
 * plonk/prover.rs:356-360): `coeffs_dev` is a HOST array of `count` device pointers; one host
 * synchronisation for the whole batch; out_jac = count x 12 limbs on the host. */
int cq_msm_batch_dev(cq_ctx* ctx, const uint64_t* const* coeffs_dev, const uint64_t* bases_dev, size_t len,
                     size_t count, uint64_t* out_jac);
/* Host helpers for multi-GPU MSM sharding: sum of `count` Jacobian points (the per-rank partial
 * results after an all-gather; EC addition is not an RCCL reduction op), and Curve::to_affine
 * (derive/curve.rs:399-412). */
int cq_g1_sum(const uint64_t* jac_points, size_t count, uint64_t out_jac[12]);
int cq_g1_to_affine(const uint64_t jac[12], uint64_t out_affine[8]);
/* Fixed-base acceleration: builds per-window tables T[w][i] = 2^(c*w) * bases[i] (ceil(255/c) x n x 64 B of
 * HBM) for a device-resident base array and registers them with the context; later multiexps over
 * that array (or a prefix of it) then need a single bucket set and no window folding.  cq_params_*
 * does this for g and g_lagrange unless disabled with cq_msm_set_precompute(ctx, 0).  Results are
 * identical either way.  The tables are a snapshot of the array: call cq_msm_forget_dev before the array is
 * freed or overwritten (cq_dev_free does it for arrays it frees). */
int cq_msm_precompute_dev(cq_ctx* ctx, const uint64_t* bases_dev, size_t n);
int cq_msm_forget_dev(cq_ctx* ctx, const uint64_t* bases_dev);
int cq_msm_set_precompute(cq_ctx* ctx, int on);
/* Pippenger window width in bits (2..15), 0 = automatic.  Tuning knob; results do not depend on it. */
int cq_msm_set_window(cq_ctx* ctx, uint32_t bits);
/* Window width (8..20) of the per-window tables built from now on by cq_msm_precompute_dev, the params and key
 * constructors; 0 = automatic: every array's own length decides (15 bits up to 2^19 points, 17 from 2^20 on), and a
 * proving key brings the small arrays it multiplies over (table SRS, cached quotients) to the width of its SRS tables --
 * MSMs over different base arrays share a launch only when their tables agree -- whatever was built first on the
 * context.  Tuning knob; results do not depend on it. */
int cq_msm_set_table_window(cq_ctx* ctx, uint32_t bits);
/* Which tables would a multiexp over [bases_dev, bases_dev + n) use?  *bits = their window width, 0 = none (plain
 * Pippenger over the array itself); `preferred_bits` != 0 asks for that width first, as a launch does for the width of
 * its longest MSM.  Introspection for callers and tests: results never depend on it. */
int cq_msm_table_width_dev(cq_ctx* ctx, const uint64_t* bases_dev, size_t n, uint32_t preferred_bits, uint32_t* bits);

/* eval_polynomial(poly, point)  arithmetic.rs:304-329 */
int cq_eval_polynomial(cq_ctx* ctx, const uint64_t* poly, size_t n, const uint64_t point[4], uint64_t out[4]);
int cq_eval_polynomial_dev(cq_ctx* ctx, const uint64_t* poly_dev, size_t n, const uint64_t point[4], uint64_t out[4]);
/* kate_division(a, b)  arithmetic.rs:351-387: (a(X) - a(b)) / (X - b), n-1 coefficients into q.
 * (The reference's always-on re-multiplication check is a debug assertion and is not reproduced.) */
int cq_kate_division(cq_ctx* ctx, const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* q);
int cq_kate_division_dev(cq_ctx* ctx, const uint64_t* a_dev, size_t n, const uint64_t b[4], uint64_t* q_dev);
/* ff::BatchInvert::batch_invert (ff 0.12; call sites poly.rs:192,232, domain.rs:124,468): in place,
 * zeros stay zero. */
int cq_batch_invert(cq_ctx* ctx, uint64_t* a, size_t n);
int cq_batch_invert_dev(cq_ctx* ctx, uint64_t* a_dev, size_t n);

/* ---- poly/domain.rs ------------------------------------------------------------------------ */
/* EvaluationDomain::new(j, k)  domain.rs:39-142 */
int cq_domain_create(cq_ctx* ctx, uint32_t j, uint32_t k, cq_domain** out);
void cq_domain_destroy(cq_domain* domain);
uint32_t cq_domain_k(const cq_domain* domain);
uint32_t cq_domain_extended_k(const cq_domain* domain);
/* get_omega / get_omega_inv / get_extended_omega / ifft_divisor (domain.rs:391-410); any pointer may be NULL */
int cq_domain_constants(const cq_domain* domain, uint64_t omega[4], uint64_t omega_inv[4],
                        uint64_t extended_omega[4], uint64_t ifft_divisor[4]);
/* lagrange_to_coeff  domain.rs:238-248 (host slice of n, in place) */
int cq_lagrange_to_coeff(cq_domain* domain, uint64_t* a);
/* coeff_to_extended  domain.rs:252-266 (n coefficients in, 2^extended_k coset evaluations out) */
int cq_coeff_to_extended(cq_domain* domain, const uint64_t* a, uint64_t* out);
/* extended_to_coeff  domain.rs:293-315 (2^extended_k in, n*(j-1) coefficients out) */
int cq_extended_to_coeff(cq_domain* domain, const uint64_t* a, uint64_t* out);
/* Device variants; `batch` columns stored back to back (stride n in, n resp. 2^extended_k out). */
int cq_lagrange_to_coeff_dev(cq_domain* domain, const uint64_t* in_dev, uint64_t* out_dev, uint32_t batch);
int cq_coeff_to_extended_dev(cq_domain* domain, const uint64_t* in_dev, uint64_t* out_dev, uint32_t batch);
int cq_extended_to_coeff_dev(cq_domain* domain, const uint64_t* in_dev, uint64_t* out_dev);

/* ---- poly/kzg/commitment.rs ---------------------------------------------------------------- */
/* ParamsKZG (commitment.rs:31-39): uploads g = [s^i]_1 and g_lagrange = [L_i(s)]_1 (2^k affine
 * points each, host memory, `RawBytes` layout) once; they stay resident in HBM. */
int cq_params_create(cq_ctx* ctx, uint32_t k, const uint64_t* g, const uint64_t* g_lagrange, cq_params** out);
/* ParamsKZG::setup_from_toxic_waste (commitment.rs:209-276; "MUST NOT be used in production"):
 * builds g and g_lagrange on the GPU from the toxic waste `s` -- for tests and benches. */
int cq_params_setup_from_toxic_waste(cq_ctx* ctx, uint32_t k, const uint64_t s[4], cq_params** out);
/* out[i] = scalars[i] * G1::generator(), device in / device out (affine). */
int cq_fixed_base_mul_dev(cq_ctx* ctx, const uint64_t* scalars_dev, size_t n, uint64_t* out_affine_dev);
/* ParamsKZG::read_custom / write_custom in the RawBytes layouts (commitment.rs:366-459):
 * k:u32 LE | n x 64 B g | n x 64 B g_lagrange | 128 B g2 | 128 B s_g2.  read: `checked` != 0 validates
 * every point (SerdeFormat::RawBytes), 0 = RawBytesUnchecked; the G2 tail is ignored.  write: emits the
 * 4 + 128 n byte G1 part. */
int cq_params_read_raw(cq_ctx* ctx, const uint8_t* buf, size_t len, int checked, cq_params** out);
int cq_params_write_raw(cq_params* params, uint8_t* buf, size_t cap, size_t* written);
/* g_to_lagrange(g, k) (arithmetic.rs:277-301): the Lagrange-basis SRS from the monomial one by an inverse FFT
 * over G1 (device arrays of 2^k affine points). */
int cq_g_to_lagrange_dev(cq_ctx* ctx, const uint64_t* g_dev, uint32_t k, uint64_t* g_lagrange_dev);
/* ParamsKZG::downsize(k) (kzg/commitment.rs:480-492): the first 2^k powers and their Lagrange basis (recomputed with
 * g_to_lagrange), as a new object. */
int cq_params_downsize(cq_params* params, uint32_t k, cq_params** out);
void cq_params_destroy(cq_params* params);
const uint64_t* cq_params_g_dev(const cq_params* params);
const uint64_t* cq_params_g_lagrange_dev(const cq_params* params);
/* ParamsKZG::commit (commitment.rs:539-543) / commit_lagrange (:496-504); the blind is ignored by
 * the reference and has no parameter here.  `len` <= 2^k. */
int cq_commit(cq_params* params, const uint64_t* poly, size_t len, uint64_t out_jac[12]);
int cq_commit_lagrange(cq_params* params, const uint64_t* poly, size_t len, uint64_t out_jac[12]);
int cq_commit_dev(cq_params* params, const uint64_t* poly_dev, size_t len, uint64_t out_jac[12]);
int cq_commit_lagrange_dev(cq_params* params, const uint64_t* poly_dev, size_t len, uint64_t out_jac[12]);

/* ---- plonk/static_lookup.rs, plonk/keygen.rs, plonk/prover.rs ------------------------------- */
/* StaticTableConfig::new(size, g1_lagrange, g_lagrange_opening_at_0)  static_lookup.rs:55-65
 * (host arrays of `size` affine points; `size` a power of two). */
int cq_table_config_create(cq_ctx* ctx, size_t size, const uint64_t* g1_lagrange,
                           const uint64_t* g_lagrange_opening_at_0, cq_table_config** out);
/* G1 part of TableSRS::setup_from_toxic_waste (kzg/commitment.rs:73-178), built on the GPU (tests/benches). */
int cq_table_config_setup_from_toxic_waste(cq_ctx* ctx, size_t size, const uint64_t s[4], cq_table_config** out);
void cq_table_config_destroy(cq_table_config* cfg);
int cq_table_config_download(cq_table_config* cfg, uint64_t* g1_lagrange, uint64_t* g_lagrange_opening_at_0);
/* StaticTableValues {size, value_index_mapping, qs}  static_lookup.rs:68-75.  `values`: `size` unique
 * field elements (CQ_ERR_ARG if not unique, as the assert at :84-85); `qs`: the cached quotient
 * commitments as affine points (normalise the reference's Vec<G1> first). */
int cq_static_table_create(cq_ctx* ctx, size_t size, const uint64_t* values, const uint64_t* qs_affine,
                           cq_static_table** out);
/* Same, with qs computed in closed form from the toxic waste: Q_i = [(T(s)-T(w^i))/(s-w^i) * w^i/N]_1
 * (equal to StaticTableValues::new's O(N^2) result, static_lookup.rs:108-119; tests/benches). */
int cq_static_table_setup_from_toxic_waste(cq_ctx* ctx, size_t size, const uint64_t* values, const uint64_t s[4],
                                           cq_static_table** out);
/* StaticTableValues::new(values, srs_g1) (static_lookup.rs:78-126): the reference's own construction --
 * iNTT of the values, one kate_division + (N-1)-term multiexp per root -- run on the GPU.
 * `srs_g1`: `size` affine powers [s^i]_1 (host). */
int cq_static_table_new(cq_ctx* ctx, size_t size, const uint64_t* values, const uint64_t* srs_g1, cq_static_table** out);
/* Same result as cq_static_table_new (bit-identical qs), computed FK-style ("fast amortized KZG proofs"): one cyclic
 * convolution of size 2N over G1 and one G1 DFT instead of N multiexps -- O(N log N) group operations. */
int cq_static_table_new_fk(cq_ctx* ctx, size_t size, const uint64_t* values, const uint64_t* srs_g1, cq_static_table** out);
void cq_static_table_destroy(cq_static_table* table);
int cq_static_table_download_qs(cq_static_table* table, uint64_t* qs_affine);

/* Gate polynomials (`Expression`, plonk/circuit.rs:780-1100, as stored in vk.cs.gates after selector
 * compression) cross the boundary as postfix programs of u32 words: word = op | arg << 8.  A column query
 * (arg = column index) is followed by one word holding the rotation as int32.  CONST pushes
 * constants[arg], SCALE multiplies the top of the stack by constants[arg] (Expression::Scaled),
 * NEG/ADD/MUL are Expression::Negated/Sum/Product.  Each program must leave exactly one value. */
#define CQ_GATE_CONST 0u
#define CQ_GATE_ADVICE 1u
#define CQ_GATE_FIXED 2u
#define CQ_GATE_INSTANCE 3u
#define CQ_GATE_NEG 4u
#define CQ_GATE_ADD 5u
#define CQ_GATE_MUL 6u
#define CQ_GATE_SCALE 7u
#define CQ_GATE_CHALLENGE 8u /* pushes user challenge `arg` (Expression::Challenge, circuit.rs:793-794) */
/* column kinds (`Any`, plonk/circuit.rs:141-160) */
#define CQ_COL_ADVICE 0u
#define CQ_COL_FIXED 1u
#define CQ_COL_INSTANCE 2u

/* The part of ConstraintSystem / ProvingKey a general circuit adds to the CQ-only shape: fixed and
 * instance columns, custom gates, the permutation argument.  All host pointers, read during
 * cq_pk_create only. */
typedef struct {
  uint32_t num_fixed;               /* cs.num_fixed_columns (+ compressed selector columns) */
  uint32_t num_instance;            /* cs.num_instance_columns */
  const uint64_t* const* fixed;     /* pk.fixed_values: num_fixed columns of 2^k elements (keygen.rs:320-326) */
  uint32_t cs_degree;               /* vk.cs_degree = cs.degree() (circuit.rs:1979-2018); 0 = 3 */
  uint32_t blinding_factors;        /* cs.blinding_factors() (circuit.rs:2022-2047); 0 = derive from advice queries */
  /* cs.advice_queries / cs.fixed_queries in registration order; evaluations are written in this order
   * (prover.rs:654-687).  When num_advice_queries == 0 the advice queries are derived as for a CQ-only
   * circuit (one query at Rotation::cur() per lookup input, first-seen order). */
  uint32_t num_advice_queries;
  const uint32_t* advice_query_columns;
  const int32_t* advice_query_rotations;
  uint32_t num_fixed_queries;
  const uint32_t* fixed_query_columns;
  const int32_t* fixed_query_rotations;
  /* gate polynomials in cs.gates order (evaluation.rs:226-235) */
  uint32_t num_gate_polys;
  const uint32_t* gate_program_lens;   /* words per polynomial */
  const uint32_t* gate_programs;       /* concatenated */
  uint32_t num_constants;
  const uint64_t* constants;           /* num_constants field elements */
  /* cs.permutation.columns (permutation.rs:20-38) and permutation::keygen::Assembly.mapping
   * (permutation/keygen.rs:14-20): for column position c and row r, mapping[(c * 2^k + r) * 2] =
   * permuted column position, [.. + 1] = permuted row.  NULL mapping = identity (no copy constraints). */
  uint32_t num_perm_columns;
  const uint32_t* perm_column_kinds;   /* CQ_COL_* */
  const uint32_t* perm_column_indices;
  const uint32_t* perm_mapping;
  /* Static lookup inputs as expressions (`Argument::input: Vec<Expression<F>>`, static_lookup.rs:171-178):
   * one postfix program per (lookup, table column) pair in the order of cq_circuit.lookup_columns, sharing
   * `constants`; evaluated over the Lagrange basis as `evaluate(expr, n, 1, ..)` does
   * (static_lookup/prover.rs:91-107).  NULL = every input is advice[lookup_columns[i]] @ Rotation::cur(). */
  const uint32_t* lookup_input_program_lens;
  const uint32_t* lookup_input_programs;
  /* Legacy (plookup-style) lookups, `cs.lookups` (plonk/lookup.rs:9-36): lookup l has legacy_lookup_widths[l] input
   * expressions and as many table expressions; programs are flattened lookup by lookup, inputs first, then tables,
   * and share `constants`.  The grand product, its quotient terms, all commitments and the sort of
   * `permute_expression_pair` (lookup/prover.rs:400-502; cq_permute_expression_pair_dev) run on the GPU. */
  uint32_t num_legacy_lookups;
  const uint32_t* legacy_lookup_widths;
  const uint32_t* legacy_program_lens;
  const uint32_t* legacy_programs;
  /* Phases (circuit.rs `advice_column_phase`, `challenge_phase`; prover.rs:392-464): advice column c is committed in
   * phase advice_column_phases[c] (NULL = everything in the first phase); user challenge i is squeezed after the
   * commitments of phase challenge_phases[i].  Circuits with more than one phase are proven with
   * cq_create_proof_phases. */
  const uint8_t* advice_column_phases;
  uint32_t num_challenges;
  const uint8_t* challenge_phases;
} cq_plonk;

/* Shape of the constraint system (stands in for ConstraintSystem, plonk/circuit.rs):
 * `num_advice` advice columns; lookup l has `lookup_widths[l]` (input, table) pairs; inputs are
 * advice[column] @ Rotation::cur() (lookup_static, circuit.rs:1579-1602), flattened in
 * lookup_columns / lookup_tables.  vk_repr = VerifyingKey::transcript_repr (plonk.rs:221-232).
 * `plonk` = NULL for a CQ-only circuit (advice columns + static lookups). */
typedef struct {
  uint32_t k;
  uint32_t num_advice;
  uint32_t num_lookups;
  const uint32_t* lookup_widths;
  const uint32_t* lookup_columns;
  cq_static_table* const* lookup_tables;
  uint64_t vk_repr[4];
  const cq_plonk* plonk;
} cq_circuit;

/* permutation::keygen::Assembly (permutation/keygen.rs:14-113): the copy-constraint bookkeeping keygen
 * runs while synthesizing; `mapping` (columns * n * 2 words, layout above) starts as the identity from
 * cq_permutation_assembly_init.  cq_permutation_assembly_copy merges the two cells' cycles exactly as
 * Assembly::copy (:43-112); CQ_ERR_ARG on an out-of-range column / row (Error::BoundsFailure).
 * `aux` and `sizes`: caller-provided scratch of columns*n*2 and columns*n words. */
void cq_permutation_assembly_init(uint32_t columns, uint32_t n, uint32_t* mapping, uint32_t* aux, uint32_t* sizes);
int cq_permutation_assembly_copy(uint32_t columns, uint32_t n, uint32_t* mapping, uint32_t* aux, uint32_t* sizes,
                                 uint32_t left_column, uint32_t left_row, uint32_t right_column, uint32_t right_row);
/* keygen_pk (plonk/keygen.rs:278-397): the domain, l0 / l_last / l_active_row on the extended coset
 * (:338-373), fixed polys and cosets (:328-336), the permutation proving key (permutation/keygen.rs:151-208),
 * the table config and `b0_g1_bound` (n-1 affine points; device pointer if b0_on_device != 0, else
 * host; may be NULL when the circuit has no static lookup, as may `cfg`).
 * Lifetimes: `params`, `cfg` and the static tables must outlive the key (as `ProvingKey<'params>` borrows them in the
 * reference); destroy keys first.  (cq_params_destroy / cq_table_config_destroy called while a key still uses the object
 * only mark it released -- the last such key frees it -- so a wrong order leaks nothing and touches no freed memory; the
 * static tables have no such grace.) */
int cq_pk_create(cq_ctx* ctx, cq_params* params, const cq_circuit* circuit, cq_table_config* cfg,
                 const uint64_t* b0_g1_bound, int b0_on_device, cq_pk** out);
/* ProvingKey::write / ProvingKey::read, SerdeFormat::RawBytes / RawBytesUnchecked (plonk.rs:349-403).  Layout:
 *   VerifyingKey::write (:92-113): k:u32 BE | #fixed:u32 BE | fixed commitments (64 B each, raw Montgomery x||y) |
 *     permutation commitments (64 B each) | per selector 2^k/8 bytes of packed bits (helpers.rs:98-105)
 *   then l0 | l_last | l_active_row | fixed_values | fixed_polys | fixed_cosets | permutation {permutations, polys,
 *     cosets} where a polynomial is len:u32 BE + len x 32 B raw limbs (poly.rs:163-170) and a slice of
 *     polynomials starts with its count:u32 BE (helpers.rs:129-140).
 * cq_pk_read_raw = cq_pk_create with those polynomials uploaded straight into HBM instead of recomputed (cosets
 * included: no NTT runs); `circuit` gives the shape (its plonk->fixed / perm_mapping are not read), `num_selectors`
 * the number of selector bit vectors to skip, `checked` != 0 validates every element (RawBytes).  The Rust reader
 * leaves static tables / b0_g1_bound empty (:396-401, "FIXME"); here they are passed as for cq_pk_create.
 * cq_pk_write_raw emits the same stream (`selector_bits` is copied through: the prover does not keep selectors). */
int cq_pk_read_raw(cq_ctx* ctx, cq_params* params, const cq_circuit* circuit, cq_table_config* cfg,
                   const uint64_t* b0_g1_bound, int b0_on_device, const uint8_t* buf, size_t len, uint32_t num_selectors,
                   int checked, cq_pk** out);
size_t cq_pk_raw_size(const cq_pk* pk, uint32_t num_selectors);
int cq_pk_write_raw(cq_pk* pk, const uint8_t* selector_bits, uint32_t num_selectors, uint8_t* buf, size_t cap, size_t* written);
/* Shards every commitment of cq_create_proof across `world` ranks by point range: rank r
 * multiplies the slice shard(len, r, world) of each (scalars, bases) pair, the 96-byte Jacobian partials of a round are
 * all-gathered and summed locally (EC addition is not an RCCL reduction op), so every rank derives the same
 * transcript.  Every rank must hold the same witness and RNG stream.  `fn` = NULL: the exchange is an ncclAllGather
 * on the context's RCCL communicator (cq_ctx_comm_init_rccl with the same rank / world); otherwise the caller's
 * collective on host buffers.  The key's MSM window tables are rebuilt for the rank's slices only (1 / world of the
 * 17 x SRS); world = 1 restores the unsharded key -- except over a one-rank RCCL communicator (fn = NULL), where the
 * prover still issues every collective, over the one rank: the path a single-GPU machine can exercise. */
int cq_pk_set_sharding(cq_pk* pk, uint32_t rank, uint32_t world, cq_allgather_fn fn, void* user);
/* Column sharding, on by default when sharded and a transport for whole columns exists: the independent
 * column transforms of a proof (advice / f / b -> coefficients, -> extended coset: plonk/prover.rs:587-603,
 * evaluation.rs:317-335, static_lookup/prover.rs:271,327) are computed by their owner rank only and broadcast, instead
 * of by every rank.  Transport: RCCL broadcasts on device buffers (`fn` = NULL, needs the context communicator), or the
 * caller's broadcast on host buffers.  `on` = 0 replicates the transforms on every rank. */
int cq_pk_set_column_sharding(cq_pk* pk, int on, cq_bcast_fn fn, void* user);
/* Resident column sharding: the next step after column sharding for the CQ-shaped circuits of the BASELINE configs (advice
 * columns + static lookups on plain advice inputs, one phase, GWC): a transformed column STAYS on its owner -- lookup l's f
 * and b polynomials and their extended cosets on the owner of lookup l, an advice polynomial on the owner of its column --
 * and only what another rank consumes travels, by point range: the slices of b_0 each rank commits (static_lookup/prover.rs:
 * 299,310), the per-owner partial quotients summed slice-wise into the h pieces each rank commits (evaluation.rs:533-548 is
 * a sum over lookups, extended_to_coeff and commit are linear), partial evaluations (a 32-byte sum per query) and the
 * slices of the GWC batch polynomial, whose kate_division runs per range with a carried Horner value.  Per proof and rank
 * that is ~(3 + L / world) x 32 B x 2^k / world x (world - 1) received instead of ~(5 L + A) x 32 B x 2^k broadcast.
 * Same proof bytes.  Transport: grouped ncclSend / ncclRecv on the context's communicator (`fn` = NULL) or the caller's
 * point-to-point hook on host buffers.  Other circuits keep the column-sharding mode set above. */
int cq_pk_set_resident_sharding(cq_pk* pk, int on, cq_exchange_fn fn, void* user);
/* Multi-open scheme of cq_create_proof*: `P: Prover` of create_proof (prover.rs:55).
 * CQ_OPENER_GWC = ProverGWC (poly/kzg/multiopen/gwc/prover.rs:42-91, one witness commitment per distinct
 * point; the default, as in tests/my_test.rs), CQ_OPENER_SHPLONK = ProverSHPLONK
 * (poly/kzg/multiopen/shplonk/prover.rs:120-286, two commitments). */
#define CQ_OPENER_GWC 0
#define CQ_OPENER_SHPLONK 1
int cq_pk_set_opener(cq_pk* pk, int opener);
/* The vanishing argument's random polynomial takes 8 * 2^k words of the caller's RNG (vanishing/prover.rs:51-55): one
 * indirect next_u64 call per word, made on a helper thread of the library while the GPU runs the first rounds.  A
 * caller whose generator can produce words in bulk registers `fill` here (NULL = back to per-word calls); the
 * `rng_state` given to cq_create_proof* is handed to it, and it must yield exactly the words `count` next_u64 calls
 * would.  Either way the callbacks may run on a thread other than the caller's, never concurrently. */
int cq_pk_set_rng_fill(cq_pk* pk, cq_rng_fill_fn fill);
void cq_pk_destroy(cq_pk* pk);
uint32_t cq_pk_usable_rows(const cq_pk* pk);
size_t cq_pk_proof_size(const cq_pk* pk);
/* create_proof (plonk/prover.rs:51-779) with ProverGWC + Blake2bWrite<Challenge255>.
 * advice_dev: `num_advice` DEVICE pointers to 2^k field elements each; rows [0, usable_rows) are the
 * assigned witness (unassigned cells zero), the rest is ignored (blinding rows are drawn from rng).
 * The proof (cq_pk_proof_size bytes) is written to `proof`. */
int cq_create_proof(cq_pk* pk, const uint64_t* const* advice_dev, cq_rng_next_u64 rng, void* rng_state,
                    uint8_t* proof, size_t proof_cap, size_t* proof_len);
/* `count` independent proofs of the circuit (BASELINE configs[4]: batches of instances): advice_dev[i] = the num_advice
 * device columns of instance i, rng_states[i] its RNG state (one `rng` function for all), proofs[i] / proof_lens[i] its
 * output (proof_cap bytes each).  The proofs are the ones cq_create_proof gives for the same (witness, RNG) one at a
 * time; here up to `lanes` (0 = 3) of them are in flight on the GPU at once, each on a library-owned stream and host
 * thread, so that one proof's latency-bound stretches are filled by another's kernels.  For circuits without instance
 * columns and with a single phase.  (Set GPU_MAX_HW_QUEUES >= 8 in the environment: HIP multiplexes a process's
 * streams onto 4 hardware queues by default and every lane uses three.) */
int cq_create_proof_batch(cq_pk* pk, size_t count, const uint64_t* const* const* advice_dev, cq_rng_next_u64 rng,
                          void* const* rng_states, uint8_t* const* proofs, size_t proof_cap, size_t* proof_lens, uint32_t lanes);
/* Same with host-resident advice columns (uploaded first). */
int cq_create_proof_host(cq_pk* pk, const uint64_t* const* advice, cq_rng_next_u64 rng, void* rng_state,
                         uint8_t* proof, size_t proof_cap, size_t* proof_len);
/* create_proof with public inputs (`instances: &[&[Fr]]`, prover.rs:64): `instances[c]` = HOST pointer to
 * `instance_lens[c]` elements of instance column c (CQ_ERR_ARG if longer than usable_rows: "InstanceTooLarge",
 * :108-110).  ProverGWC has QUERY_INSTANCE = false: the values are absorbed into the transcript (:305-312)
 * and no instance commitment / evaluation is written.  `advice_on_device` selects the two variants above. */
int cq_create_proof_instances(cq_pk* pk, const uint64_t* const* advice, int advice_on_device,
                              const uint64_t* const* instances, const size_t* instance_lens, cq_rng_next_u64 rng,
                              void* rng_state, uint8_t* proof, size_t proof_cap, size_t* proof_len);
/* Multi-phase circuits: the witness of phase p > 0 depends on the challenges squeezed after the commitments of the
 * earlier phases (`WitnessCollection::next_phase`, prover.rs:299-391; the synthesis loop :436-463).  Before it
 * commits the columns of phase p the library calls `phase_fn(user, p, challenges, advice_dev)`: `challenges` holds
 * num_challenges x 4 limbs (those of later phases are zero), and the callback fills the phase-p advice columns
 * (device memory, rows [0, usable_rows)) -- it stands in for re-running `FloorPlanner::synthesize` with
 * `current_phase = p`.  A non-zero return aborts the proof (CQ_ERR_ARG). */
typedef int (*cq_phase_fn)(void* user, uint32_t phase, const uint64_t* challenges, uint64_t* const* advice_dev);
int cq_create_proof_phases(cq_pk* pk, uint64_t* const* advice_dev, const uint64_t* const* instances,
                           const size_t* instance_lens, cq_phase_fn phase_fn, void* phase_user, cq_rng_next_u64 rng,
                           void* rng_state, uint8_t* proof, size_t proof_cap, size_t* proof_len);
/* ---- the CQ sub-arguments on their own (for a host that keeps its own create_proof and swaps in only these) ---------
 * Shapes: L = lookups of `pk`, n = 2^k, N = table size, ext = 2^extended_k; lookups in cs.static_lookups order.
 *
 * static_lookup::Argument::commit (plonk/static_lookup/prover.rs:51-183), called at plonk/prover.rs:512-522: evaluates
 * the lookup inputs over the Lagrange basis, compresses them with theta into f (:108-116), maps every usable row to its
 * table index and counts the multiplicities m (:122-160; CQ_ERR_LOOKUP as the reference's errors), commits both.
 *   advice_dev:   num_advice DEVICE columns of n elements, blinding rows included (f is committed over all n rows)
 *   instance_dev: num_instance device columns (NULL without instance columns); challenges: num_challenges x 4 limbs, HOST
 *   f_dev:  L x n elements out (Lagrange values of f);  m_dev: L x N uint32 out (m_sparse, dense)
 *   commitments: HOST, L x 2 affine points (8 limbs each): f_cm, m_cm per lookup -- the order they are written (:175-176) */
int cq_cq_round1_dev(cq_pk* pk, const uint64_t* const* advice_dev, const uint64_t* const* instance_dev, const uint64_t* challenges,
                     const uint64_t theta[4], uint64_t* f_dev, uint32_t* m_dev, uint64_t* commitments);
/* static_lookup::Committed::commit_log_derivatives (:187-342), called at plonk/prover.rs:572-575: A_i = m_i / (t_i + beta)
 * and its three commitments over the table SRS / cached quotients (:224-257), B = 1 / (f + beta), b = iNTT(B) (:261-276),
 * b_0 = (b - b(0)) / X, its commitment and the degree-bound commitment p (:279-313), a(0) (:318-324), f in coefficient
 * form (:326-334).
 *   f_dev, m_dev: as left by cq_cq_round1_dev;  b_coeff_dev, f_coeff_dev: L x n elements out (b_0 is b_coeff + 1, n - 1 long)
 *   commitments: HOST, L x 5 affine points: a, q_a, a_0, b_0, p per lookup (write order :306-313); a_at_zero: HOST, L x 4 limbs */
int cq_cq_round2_dev(cq_pk* pk, const uint64_t* f_dev, const uint32_t* m_dev, const uint64_t theta[4], const uint64_t beta[4],
                     uint64_t* b_coeff_dev, uint64_t* f_coeff_dev, uint64_t* commitments, uint64_t* a_at_zero);
/* The static-lookup part of Evaluator::evaluate_h (plonk/evaluation.rs:533-548; called at plonk/prover.rs:606-624):
 * h <- h * y + (b (f l_active_row + beta) - 1) per lookup on the extended coset, starting from h_in_dev (the gate /
 * permutation / legacy-lookup terms folded so far; NULL = zero); b and f are taken in coefficient form and extended here
 * (:535-536).  divide_by_vanishing != 0 also applies EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:319-338,
 * what vanishing::Argument::construct does next, vanishing/prover.rs:84).  h_out_dev: ext elements (may alias h_in_dev). */
int cq_quotient_dev(cq_pk* pk, const uint64_t* b_coeff_dev, const uint64_t* f_coeff_dev, const uint64_t y[4], const uint64_t beta[4],
                    const uint64_t* h_in_dev, int divide_by_vanishing, uint64_t* h_out_dev);
/* permute_expression_pair of the legacy (halo2) lookup argument, plonk/lookup/prover.rs:400-502, without the blinding rows
 * it appends (:486-497, the caller's RNG): the first `usable` rows of the compressed input expression sorted by canonical
 * value, and the compressed table expression rearranged so that every first occurrence of an input value faces the same
 * table value and the remaining table values fill the repeated rows the way the reference does (ascending values into
 * descending rows).  All arrays: device, Montgomery-form field elements; the outputs hold `usable` elements and may not
 * alias the inputs.  2^k >= usable is the domain size (sizes the sort).  CQ_ERR_LOOKUP when an input value is missing from
 * the table (Error::ConstraintSystemFailure, :456-460). */
int cq_permute_expression_pair_dev(cq_ctx* ctx, uint32_t k, uint32_t usable, const uint64_t* input_dev, const uint64_t* table_dev,
                                   uint64_t* permuted_input_dev, uint64_t* permuted_table_dev);
/* The verifying-key commitments the prover's key implies: commit_lagrange of every fixed column
 * (keygen.rs:247-250) and of every permutation polynomial (permutation/keygen.rs:115-149), as affine
 * points (num_fixed x 8 and num_perm_columns x 8 words). */
int cq_pk_vk_commitments(cq_pk* pk, uint64_t* fixed_commitments, uint64_t* permutation_commitments);

/* ---- sha/src/tables.rs: SHA-256 word -> limb witness fill ----------------------------------------- */
/* Decomposes `nwords` 32-bit words (device) into 12/10/10-bit limbs (LongLimbs, tables.rs:70-75,
 * 135-154) and writes, for limb t, its dense value to column 2*(t % pairs) and its bit-spread value
 * (bit i -> bit 2i) to column 2*(t % pairs)+1 at row t / pairs, Montgomery-encoded.  `cols_dev`:
 * 2*pairs device columns of `n` elements, zero-filled by the caller. */
int cq_sha_witness_fill_dev(cq_ctx* ctx, const uint32_t* words_dev, size_t nwords, uint32_t pairs, size_t n,
                            uint64_t* const* cols_dev);
/* sha/src/tables.rs generators, rows of four u64 written to device memory:
 * create_{rot0,rot1,maj,ch}_table::<L> (tables.rs:105-133; kind 0..3), 2^(first+2*second) rows (x,y,z,f);
 * create_decomposition_table::<L,K> (tables.rs:135-154), 2^k_bits rows (a,x,y,z). */
int cq_sha_synthesis_table_dev(cq_ctx* ctx, int kind, uint32_t first_limb_len, uint32_t second_limb_len, uint64_t* out_dev);
int cq_sha_decomposition_table_dev(cq_ctx* ctx, uint32_t first_limb_len, uint32_t second_limb_len, uint32_t k_bits,
                                   uint64_t* out_dev);
/* dense[i] = i, spread[i] = bit-spread(i) for i < size (device arrays of `size` elements). */
int cq_sha_spread_table_dev(cq_ctx* ctx, size_t size, uint64_t* dense_dev, uint64_t* spread_dev);

/* ---- harness RNG (not in the reference: its test draws from OsRng) ---------------------------------- */
void cq_xoshiro256ss_seed(uint64_t seed, uint64_t state[4]);
uint64_t cq_xoshiro256ss_next_u64(void* state /* uint64_t[4] */);
/* `count` consecutive outputs of the generator above into dst, advancing the state -- the same words and final state
 * as `count` calls of cq_xoshiro256ss_next_u64, produced by up to `threads` host threads (jump-ahead on the
 * GF(2)-linear state transition).  cq_create_proof* draws the vanishing argument's random polynomial this way
 * (8 * 2^k words: the RNG, not the GPU, bounds the first rounds of a large proof otherwise). */
void cq_xoshiro256ss_fill(uint64_t state[4], uint64_t* dst, size_t count, uint32_t threads);
/* replays a pre-drawn stream: state = {const uint64_t* words; size_t pos; size_t len; size_t overrun}.  Words asked
 * for beyond `len` read as zero and are counted in `overrun` (reset by cq_create_proof* when a proof starts: the
 * caller need not initialise it); cq_create_proof* fails with CQ_ERR_ARG when the stream it was given ran out (a proof
 * blinded with zeros is not zero-knowledge); an unrelated failure (e.g. CQ_ERR_LOOKUP) keeps its own code.  Like the xoshiro generator below this is
 * for tests and benches; production blinding comes from the caller's CSPRNG through cq_rng_next_u64. */
typedef struct { const uint64_t* words; size_t pos; size_t len; size_t overrun; } cq_buffer_rng;
uint64_t cq_buffer_rng_next_u64(void* state /* cq_buffer_rng* */);
/* The xoshiro generator behind a function the library does NOT recognise (state: uint64_t[4]): stands in for a
 * caller's opaque `RngCore` in tests and in bench.py's `generic_rng` leg (same words as cq_xoshiro256ss_next_u64). */
uint64_t cq_opaque_rng_next_u64(void* state);
void cq_opaque_rng_fill(void* state, uint64_t* dst, size_t count);

/* ---- measurement support ---------------------------------------------------------------------- */
/* When enabled, the library brackets its dominant kernels with HIP events on the context's stream.
 * cq_profile_read drains the stream and returns the summed duration and launch count for `id`
 * (and forgets those spans). */
#define CQ_PROF_MSM_ACCUMULATE 1 /* msm_accumulate_kernel: bucket accumulation (mixed additions) */
#define CQ_PROF_NTT_PASS 2       /* ntt_pass_kernel: one radix-2^deg Stockham pass */
#define CQ_PROF_MSM_ENTRIES 3    /* no timing: `calls` = (point, non-zero digit) pairs = mixed additions executed by
                                  * msm_accumulate_kernel since the last read */
int cq_profile_enable(cq_ctx* ctx, int on);
int cq_profile_read(cq_ctx* ctx, int id, double* total_ms, uint64_t* calls);

/* ---- microbenchmarks (measurement support, not part of the drop-in surface) -------------- */
/* Runs `iters` dependent Montgomery multiplications per lane over `lanes` lanes and writes one
 * folded element per lane; used to measure the chip's 256-bit modmul rate. which: 0 = Fr, 1 = Fq (the 8 x u32
 * memory-format type), 2 = Fq in the lazy 9 x 29-bit form the MSM kernels compute in */
int cq_bench_modmul_dev(cq_ctx* ctx, uint64_t* out_dev, uint32_t lanes, uint32_t iters, int which);

#ifdef __cplusplus
}
#endif
#endif /* CQ_HALO2_H */
