/* A plain C11 consumer of include/cq_halo2.h, linked against libcq_halo2.so: what a maintainer's FFI layer (the Rust
 * `extern "C"` block of INTEGRATION.md, cgo, ...) sees.  Compiled with -std=c11 -Wall -Wextra -Werror -pedantic by
 * tests/test_abi_consumer.py, which compares what it prints with the oracle.
 *
 *   consumer layout          sizeof / offsetof of the structs that cross the ABI (no GPU needed)
 *   consumer run <k> <seed>  best_fft, best_multiexp, commit_lagrange and a create_proof with HOST advice columns on cuda:0
 */
#include <inttypes.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cq_halo2.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != CQ_OK) {                                                                      \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, ctx ? cq_last_error(ctx) : "no ctx"); \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

static void print_limbs(const char* tag, const uint64_t* w, size_t words) {
  printf("%s", tag);
  for (size_t i = 0; i < words; i++) printf(" %016" PRIx64, w[i]);
  printf("\n");
}

/* FNV-1a over the bytes of an array: lets the test compare a long output without printing it */
static uint64_t fnv(const void* p, size_t bytes) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

static int layout(void) {
#define FIELD(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
  printf("sizeof.cq_plonk %zu\n", sizeof(cq_plonk));
  FIELD(cq_plonk, num_fixed);
  FIELD(cq_plonk, num_instance);
  FIELD(cq_plonk, fixed);
  FIELD(cq_plonk, cs_degree);
  FIELD(cq_plonk, blinding_factors);
  FIELD(cq_plonk, num_advice_queries);
  FIELD(cq_plonk, advice_query_columns);
  FIELD(cq_plonk, advice_query_rotations);
  FIELD(cq_plonk, num_fixed_queries);
  FIELD(cq_plonk, fixed_query_columns);
  FIELD(cq_plonk, fixed_query_rotations);
  FIELD(cq_plonk, num_gate_polys);
  FIELD(cq_plonk, gate_program_lens);
  FIELD(cq_plonk, gate_programs);
  FIELD(cq_plonk, num_constants);
  FIELD(cq_plonk, constants);
  FIELD(cq_plonk, num_perm_columns);
  FIELD(cq_plonk, perm_column_kinds);
  FIELD(cq_plonk, perm_column_indices);
  FIELD(cq_plonk, perm_mapping);
  FIELD(cq_plonk, lookup_input_program_lens);
  FIELD(cq_plonk, lookup_input_programs);
  FIELD(cq_plonk, num_legacy_lookups);
  FIELD(cq_plonk, legacy_lookup_widths);
  FIELD(cq_plonk, legacy_program_lens);
  FIELD(cq_plonk, legacy_programs);
  FIELD(cq_plonk, advice_column_phases);
  FIELD(cq_plonk, num_challenges);
  FIELD(cq_plonk, challenge_phases);
  printf("sizeof.cq_circuit %zu\n", sizeof(cq_circuit));
  FIELD(cq_circuit, k);
  FIELD(cq_circuit, num_advice);
  FIELD(cq_circuit, num_lookups);
  FIELD(cq_circuit, lookup_widths);
  FIELD(cq_circuit, lookup_columns);
  FIELD(cq_circuit, lookup_tables);
  FIELD(cq_circuit, vk_repr);
  FIELD(cq_circuit, plonk);
  printf("sizeof.cq_buffer_rng %zu\n", sizeof(cq_buffer_rng));
  FIELD(cq_buffer_rng, words);
  FIELD(cq_buffer_rng, pos);
  FIELD(cq_buffer_rng, len);
  FIELD(cq_buffer_rng, overrun);
#undef FIELD
  printf("version %s\n", cq_version());
  return 0;
}

/* `count` field elements below 2^252 (valid Montgomery residues whatever they mean) from the harness generator */
static void draw(uint64_t st[4], uint64_t* dst, size_t count) {
  for (size_t i = 0; i < count; i++) {
    for (int l = 0; l < 4; l++) dst[4 * i + l] = cq_xoshiro256ss_next_u64(st);
    dst[4 * i + 3] &= (1ull << 60) - 1;
  }
}

static int run(uint32_t k, uint64_t seed) {
  cq_ctx* ctx = NULL;
  const size_t n = (size_t)1 << k, N = 64;
  uint64_t st[4];
  cq_xoshiro256ss_seed(seed, st);
  CHECK(cq_ctx_create(0, NULL, &ctx));

  /* toxic waste, SRS */
  uint64_t s[4];
  draw(st, s, 1);
  print_limbs("s", s, 4);
  cq_params* params = NULL;
  CHECK(cq_params_setup_from_toxic_waste(ctx, k, s, &params));
  uint64_t* g_lagrange = malloc(n * 64);
  uint64_t* scalars = malloc(n * 32);
  uint64_t* fft = malloc(n * 32);
  if (!g_lagrange || !scalars || !fft) return 2;
  CHECK(cq_dev_download(ctx, g_lagrange, cq_params_g_lagrange_dev(params), n * 64));
  draw(st, scalars, n);

  /* best_fft (arithmetic.rs:171-234) with the domain's omega */
  cq_domain* dom = NULL;
  uint64_t omega[4];
  CHECK(cq_domain_create(ctx, 3, k, &dom));
  CHECK(cq_domain_constants(dom, omega, NULL, NULL, NULL));
  memcpy(fft, scalars, n * 32);
  CHECK(cq_best_fft(ctx, fft, k, omega));
  printf("best_fft_fnv %016" PRIx64 "\n", fnv(fft, n * 32));
  print_limbs("best_fft_first", fft, 4);

  /* best_multiexp (arithmetic.rs:132-159) over host slices == commit_lagrange (kzg/commitment.rs:496-504) */
  uint64_t jac[12], jac2[12], aff[8], aff2[8];
  CHECK(cq_best_multiexp(ctx, scalars, g_lagrange, n, jac));
  CHECK(cq_commit_lagrange(params, scalars, n, jac2));
  CHECK(cq_g1_to_affine(jac, aff));
  CHECK(cq_g1_to_affine(jac2, aff2));
  print_limbs("best_multiexp", aff, 8);
  print_limbs("commit_lagrange", aff2, 8);

  /* a proof: 2 advice columns, one width-2 static lookup into (dense, spread) tables of 64 rows */
  cq_table_config* cfg = NULL;
  cq_static_table *dense = NULL, *spread = NULL;
  void *dense_dev = NULL, *spread_dev = NULL;
  uint64_t* dense_h = malloc(N * 32);
  uint64_t* spread_h = malloc(N * 32);
  if (!dense_h || !spread_h) return 2;
  CHECK(cq_table_config_setup_from_toxic_waste(ctx, N, s, &cfg));
  CHECK(cq_dev_alloc(ctx, N * 32, &dense_dev));
  CHECK(cq_dev_alloc(ctx, N * 32, &spread_dev));
  CHECK(cq_sha_spread_table_dev(ctx, N, dense_dev, spread_dev));
  CHECK(cq_dev_download(ctx, dense_h, dense_dev, N * 32));
  CHECK(cq_dev_download(ctx, spread_h, spread_dev, N * 32));
  CHECK(cq_static_table_setup_from_toxic_waste(ctx, N, dense_h, s, &dense));
  CHECK(cq_static_table_setup_from_toxic_waste(ctx, N, spread_h, s, &spread));
  const uint32_t widths[1] = {2}, columns[2] = {0, 1};
  cq_static_table* tables[2];
  tables[0] = dense;
  tables[1] = spread;
  cq_circuit circuit;
  memset(&circuit, 0, sizeof circuit);
  circuit.k = k;
  circuit.num_advice = 2;
  circuit.num_lookups = 1;
  circuit.lookup_widths = widths;
  circuit.lookup_columns = columns;
  circuit.lookup_tables = tables;
  circuit.vk_repr[0] = 0xC0FFEE;  /* any residue: opaque to the backend (plonk.rs:221-232) */
  circuit.plonk = NULL;
  cq_pk* pk = NULL;
  /* b0_g1_bound = [s^1 .. s^(n-1)]_1 = g[1..] on the device */
  CHECK(cq_pk_create(ctx, params, &circuit, cfg, cq_params_g_dev(params) + 8, 1, &pk));
  const uint32_t usable = cq_pk_usable_rows(pk);
  uint64_t* a0 = calloc(n, 32);
  uint64_t* a1 = calloc(n, 32);
  if (!a0 || !a1) return 2;
  for (uint32_t r = 0; r < usable; r++) {
    const size_t idx = ((size_t)r * 7 + 3) % N;
    memcpy(a0 + 4 * (size_t)r, dense_h + 4 * idx, 32);
    memcpy(a1 + 4 * (size_t)r, spread_h + 4 * idx, 32);
  }
  const uint64_t* advice[2];
  advice[0] = a0;
  advice[1] = a1;
  uint64_t rng[4];
  cq_xoshiro256ss_seed(seed + 1, rng);
  const size_t cap = cq_pk_proof_size(pk);
  uint8_t* proof = malloc(cap);
  size_t len = 0;
  if (!proof) return 2;
  CHECK(cq_create_proof_host(pk, advice, cq_xoshiro256ss_next_u64, rng, proof, cap, &len));
  printf("proof");
  for (size_t i = 0; i < len; i++) printf("%s%02x", i ? "" : " ", proof[i]);
  printf("\n");
  /* contract violations come back as a status, never as an abort: a short buffer ... */
  cq_xoshiro256ss_seed(seed + 1, rng);
  printf("short_buffer_rc %d\n", cq_create_proof_host(pk, advice, cq_xoshiro256ss_next_u64, rng, proof, len - 1, &len));
  /* ... and a witness value that is not in the table (static_lookup/prover.rs:141 panics there) */
  a0[0] = 12345;
  cq_xoshiro256ss_seed(seed + 1, rng);
  printf("lookup_miss_rc %d\n", cq_create_proof_host(pk, advice, cq_xoshiro256ss_next_u64, rng, proof, cap, &len));

  cq_pk_destroy(pk);
  cq_static_table_destroy(dense);
  cq_static_table_destroy(spread);
  cq_table_config_destroy(cfg);
  CHECK(cq_dev_free(ctx, dense_dev));
  CHECK(cq_dev_free(ctx, spread_dev));
  cq_domain_destroy(dom);
  cq_params_destroy(params);
  cq_ctx_destroy(ctx);
  free(g_lagrange); free(scalars); free(fft); free(dense_h); free(spread_h); free(a0); free(a1); free(proof);
  printf("done\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "layout") == 0) return layout();
  if (argc >= 4 && strcmp(argv[1], "run") == 0) return run((uint32_t)strtoul(argv[2], NULL, 10), strtoull(argv[3], NULL, 10));
  fprintf(stderr, "usage: consumer layout | consumer run <k> <seed>\n");
  return 64;
}
