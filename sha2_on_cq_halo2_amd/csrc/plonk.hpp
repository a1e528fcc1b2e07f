// Internal interface of the general-PLONK device kernels (plonk.hip): custom-gate evaluation on the
// extended coset, the permutation argument's grand product and its quotient terms.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cq_halo2.h"
#include "field.hpp"

struct cq_ctx;

namespace cq {

constexpr uint32_t GATE_STACK = 16;        // operand stack depth of the gate interpreter
constexpr uint32_t PERM_MAX_COLUMNS = 64;  // columns in the permutation argument
constexpr uint32_t PERM_MAX_CHUNK = 16;    // columns per grand-product set (cs_degree - 2)

// Gate programs: one postfix program per gate polynomial, u32 words.  word = op | arg << 8; column
// queries are followed by one word holding the rotation (int32).
enum GateOp : uint32_t {
  GATE_CONST = CQ_GATE_CONST,
  GATE_ADVICE = CQ_GATE_ADVICE,
  GATE_FIXED = CQ_GATE_FIXED,
  GATE_INSTANCE = CQ_GATE_INSTANCE,
  GATE_NEG = CQ_GATE_NEG,
  GATE_ADD = CQ_GATE_ADD,
  GATE_MUL = CQ_GATE_MUL,
  GATE_SCALE = CQ_GATE_SCALE,
  GATE_CHALLENGE = CQ_GATE_CHALLENGE,
};

struct GateEvalArgs {
  const uint32_t* prog;      // device: [len_0, words..., len_1, words..., ...]
  uint32_t num_polys;
  const Fr* constants;       // device
  const Fr* challenges;      // device: user challenges (Expression::Challenge)
  const Fr* advice;          // column c at advice + c * stride
  const Fr* fixed;
  const Fr* instance;
  size_t stride;             // elements between columns (= size)
  uint32_t size;             // domain size (power of two)
  uint32_t rot_scale;        // 2^(extended_k - k) on the coset, 1 on the Lagrange basis
  Fr y;
};

struct PermProductArgs {
  const Fr* col[PERM_MAX_CHUNK];    // Lagrange values of the set's columns
  const Fr* sigma[PERM_MAX_CHUNK];  // pk.permutation.permutations
  Fr delta_beta[PERM_MAX_CHUNK];    // delta^(global column position) * beta
  uint32_t count;
  Fr beta, gamma;
  const Fr* omega_powers;           // omega^i, i < n
};

struct PermScaleArgs {
  Fr mult[PERM_MAX_COLUMNS];  // chain multiplier per set (last_z of the previous set)
};

struct PermHArgs {
  const Fr* z;        // sets x ext, contiguous
  const Fr* sigma;    // columns x ext, contiguous (pk.permutation.cosets)
  const Fr* col[PERM_MAX_COLUMNS];  // value cosets per permutation column
  uint32_t sets, chunk_len, ncols;
  const Fr *l0, *l_last, *l_active;
  Fr beta, gamma, y, delta_start, delta;
  const Fr *ext_pow_lo, *ext_pow_hi;  // extended_omega^t (t < 256) and extended_omega^(256 b): omega_ext^i from two loads
  uint32_t ext, rot_scale, last_rot;  // last_rot = blinding_factors + 1
};

struct LookupHArgs {
  const Fr *z, *a, *s;            // product, permuted input, permuted table on the extended coset
  const Fr *cin, *ctab;           // theta-compressed input / table expressions on the extended coset
  const Fr *l0, *l_last, *l_active;
  Fr beta, gamma, y;
  uint32_t ext, rot_scale;
};

// checks a gate program blob on the host; returns false with `why` set if malformed
bool gate_program_check(const uint32_t* lens, const uint32_t* words, uint32_t num_polys, uint32_t num_constants,
                        uint32_t num_advice, uint32_t num_fixed, uint32_t num_instance, uint32_t num_challenges,
                        const char** why, size_t* total_words);

int gate_eval(cq_ctx* c, const GateEvalArgs& a, Fr* h);  // h[i] = Horner_y(gate polynomials)(i)
int perm_sigma(cq_ctx* c, const uint32_t* mapping_dev, uint32_t ncols, uint32_t n, const Fr* omega_powers,
               const Fr* delta_powers_dev, Fr* out);
int fr_powers(cq_ctx* c, const Fr& base, uint32_t n, Fr* out);  // out[i] = base^i
int perm_denominators(cq_ctx* c, const PermProductArgs& a, uint32_t n, Fr* den);
int perm_numerators(cq_ctx* c, const PermProductArgs& a, uint32_t n, Fr* den_inv_inout);
// exclusive prefix product of `batch` rows of n elements: out[0] = 1, out[i] = prod_{r<i} in[r]
int prefix_product(cq_ctx* c, const Fr* in, Fr* out, uint32_t n, uint32_t batch);
int perm_scale(cq_ctx* c, Fr* z, uint32_t n, uint32_t rows, uint32_t sets, const PermScaleArgs& a);
int perm_h_terms(cq_ctx* c, const PermHArgs& a, Fr* h);  // h <- fold of the permutation constraints, in place
// legacy lookup argument (plonk/lookup/prover.rs:163-300, plonk/evaluation.rs:461-531)
int lookup_denominators(cq_ctx* c, const Fr* a, const Fr* s, const Fr& beta, const Fr& gamma, uint32_t n, Fr* out);
int lookup_numerators(cq_ctx* c, const Fr* cin, const Fr* ctab, const Fr& beta, const Fr& gamma, uint32_t n, Fr* inout);
int lookup_h_terms(cq_ctx* c, const LookupHArgs& a, Fr* h);
// Montgomery <-> canonical limbs (the host sorts canonical values, derive/field.rs `Ord`)
int fr_to_canonical(cq_ctx* c, const Fr* in, uint32_t n, uint64_t* out);
// permute_expression_pair on the device (lksort.hip)
size_t lookup_permute_scratch_bytes(uint32_t k);
int lookup_permute_dev(cq_ctx* c, uint64_t* in_canon, uint64_t* tab_canon, uint32_t u, uint32_t k, uint64_t* out_tab, void* scratch,
                       uint32_t* status_dev);
int fr_from_canonical(cq_ctx* c, const uint64_t* in, uint32_t n, Fr* out);

}  // namespace cq
