// General-PLONK device kernels for gfx950: what a circuit with custom gates, fixed/instance columns
// and copy constraints adds to the CQ-only proving path.
//   * gate_eval_kernel      -- `GraphEvaluator::evaluate` over the custom gates (plonk/evaluation.rs:226-235,
//                              729-774): h(i) = Horner_y(gate polynomials), one lane per extended-coset row,
//                              a small postfix interpreter whose program is wave-uniform (scalar loads);
//   * perm_* / prefix_product -- `permutation::Argument::commit` (plonk/permutation/prover.rs:47-198): the
//                              reference's serial running product z (:160-166) becomes a three-phase
//                              multiplicative scan;
//   * perm_h_kernel         -- the permutation terms of `evaluate_h` (plonk/evaluation.rs:367-459);
//   * perm_sigma_kernel     -- `Assembly::build_pk` (plonk/permutation/keygen.rs:151-208).
// All are streaming, HBM-bound passes: coalesced 32-byte-per-lane loads, every vector touched once.
#include "plonk.hpp"
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
static inline uint32_t blocks_for(uint32_t n) { return (n + 255) / 256; }

// ---- custom gates -----------------------------------------------------------------------------------
// get_rotation_idx (evaluation.rs:37-39): (idx + rot * rot_scale) mod size, size a power of two
static __device__ __forceinline__ uint32_t rot_idx(uint32_t i, int32_t rot, uint32_t rot_scale, uint32_t size) {
  return (uint32_t)((int64_t)i + (int64_t)rot * (int64_t)rot_scale) & (size - 1);
}

__global__ __launch_bounds__(256) void gate_eval_kernel(GateEvalArgs a, Fr* __restrict__ h) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.size) return;
  Fr stack[GATE_STACK];
  Fr acc = Fr::zero();
  uint32_t pc = 0;
  for (uint32_t poly = 0; poly < a.num_polys; poly++) {
    const uint32_t end = pc + 1 + a.prog[pc];
    pc++;
    uint32_t sp = 0;
    while (pc < end) {
      const uint32_t w = a.prog[pc++];
      const uint32_t op = w & 0xffu, arg = w >> 8;
      switch (op) {
        case GATE_CONST:
          stack[sp++] = ld(a.constants + arg);
          break;
        case GATE_ADVICE:
        case GATE_FIXED:
        case GATE_INSTANCE: {
          const int32_t rot = (int32_t)a.prog[pc++];
          const Fr* base = op == GATE_ADVICE ? a.advice : (op == GATE_FIXED ? a.fixed : a.instance);
          stack[sp++] = ld(base + (size_t)arg * a.stride + rot_idx(i, rot, a.rot_scale, a.size));
          break;
        }
        case GATE_CHALLENGE:
          stack[sp++] = ld(a.challenges + arg);
          break;
        case GATE_NEG:
          stack[sp - 1] = stack[sp - 1].neg();
          break;
        case GATE_ADD:
          stack[sp - 2] = stack[sp - 2] + stack[sp - 1];
          sp--;
          break;
        case GATE_MUL:
          stack[sp - 2] = stack[sp - 2] * stack[sp - 1];
          sp--;
          break;
        default:  // GATE_SCALE
          stack[sp - 1] = stack[sp - 1] * ld(a.constants + arg);
          break;
      }
    }
    acc = acc * a.y + stack[0];
  }
  st(h + i, acc);
}

bool gate_program_check(const uint32_t* lens, const uint32_t* words, uint32_t num_polys, uint32_t num_constants,
                        uint32_t num_advice, uint32_t num_fixed, uint32_t num_instance, uint32_t num_challenges,
                        const char** why, size_t* total_words) {
  size_t off = 0;
  for (uint32_t p = 0; p < num_polys; p++) {
    const uint32_t len = lens[p];
    uint32_t sp = 0;
    for (uint32_t q = 0; q < len; q++) {
      const uint32_t w = words[off + q], op = w & 0xffu, arg = w >> 8;
      switch (op) {
        case CQ_GATE_CONST:
          if (arg >= num_constants) { *why = "gate program: constant index out of range"; return false; }
          sp++;
          break;
        case CQ_GATE_ADVICE:
        case CQ_GATE_FIXED:
        case CQ_GATE_INSTANCE: {
          const uint32_t lim = op == CQ_GATE_ADVICE ? num_advice : (op == CQ_GATE_FIXED ? num_fixed : num_instance);
          if (arg >= lim) { *why = "gate program: column index out of range"; return false; }
          if (++q >= len) { *why = "gate program: query without rotation word"; return false; }
          sp++;
          break;
        }
        case CQ_GATE_CHALLENGE:
          if (arg >= num_challenges) { *why = "gate program: challenge index out of range"; return false; }
          sp++;
          break;
        case CQ_GATE_NEG:
          if (sp < 1) { *why = "gate program: stack underflow"; return false; }
          break;
        case CQ_GATE_ADD:
        case CQ_GATE_MUL:
          if (sp < 2) { *why = "gate program: stack underflow"; return false; }
          sp--;
          break;
        case CQ_GATE_SCALE:
          if (sp < 1) { *why = "gate program: stack underflow"; return false; }
          if (arg >= num_constants) { *why = "gate program: constant index out of range"; return false; }
          break;
        default:
          *why = "gate program: unknown opcode";
          return false;
      }
      if (sp > GATE_STACK) { *why = "gate program: expression too deep for the operand stack"; return false; }
    }
    if (sp != 1) { *why = "gate program: must leave exactly one value"; return false; }
    off += len;
  }
  *total_words = off;
  return true;
}

int gate_eval(cq_ctx* c, const GateEvalArgs& a, Fr* h) {
  gate_eval_kernel<<<blocks_for(a.size), 256, 0, c->stream>>>(a, h);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "gate_eval launch failed");
}

// ---- permutation keygen --------------------------------------------------------------------------------
__global__ void fr_powers_kernel(Fr base, uint32_t n, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(out + i, base.pow_u64(i));
}

int fr_powers(cq_ctx* c, const Fr& base, uint32_t n, Fr* out) {
  fr_powers_kernel<<<blocks_for(n), 256, 0, c->stream>>>(base, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "fr_powers launch failed");
}

// sigma_c[i] = delta^(mapped column) * omega^(mapped row)   (permutation/keygen.rs:186-194)
__global__ void perm_sigma_kernel(const uint32_t* __restrict__ mapping, uint32_t n, const Fr* __restrict__ omega_powers,
                                  const Fr* __restrict__ delta_powers, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t cell = (size_t)blockIdx.y * n + i;
  const uint32_t mc = mapping[2 * cell], mr = mapping[2 * cell + 1];
  st(out + cell, ld(delta_powers + mc) * ld(omega_powers + mr));
}

int perm_sigma(cq_ctx* c, const uint32_t* mapping_dev, uint32_t ncols, uint32_t n, const Fr* omega_powers,
               const Fr* delta_powers_dev, Fr* out) {
  perm_sigma_kernel<<<dim3(blocks_for(n), ncols), 256, 0, c->stream>>>(mapping_dev, n, omega_powers, delta_powers_dev, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "perm_sigma launch failed");
}

// ---- permutation grand product ---------------------------------------------------------------------------
// den[i] = prod_j (beta * sigma_j[i] + gamma + v_j[i])   (permutation/prover.rs:106-121)
__global__ __launch_bounds__(256) void perm_den_kernel(PermProductArgs a, uint32_t n, Fr* __restrict__ den) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr acc = Fr::one();
  for (uint32_t j = 0; j < a.count; j++) acc = acc * (a.beta * ld(a.sigma[j] + i) + a.gamma + ld(a.col[j] + i));
  st(den + i, acc);
}
// mv[i] = den_inv[i] * prod_j (delta^j' omega^i beta + gamma + v_j[i])   (:128-147)
__global__ __launch_bounds__(256) void perm_num_kernel(PermProductArgs a, uint32_t n, Fr* __restrict__ mv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr acc = ld(mv + i);
  const Fr w = ld(a.omega_powers + i);
  for (uint32_t j = 0; j < a.count; j++) acc = acc * (a.delta_beta[j] * w + a.gamma + ld(a.col[j] + i));
  st(mv + i, acc);
}

int perm_denominators(cq_ctx* c, const PermProductArgs& a, uint32_t n, Fr* den) {
  perm_den_kernel<<<blocks_for(n), 256, 0, c->stream>>>(a, n, den);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "perm_den launch failed");
}
int perm_numerators(cq_ctx* c, const PermProductArgs& a, uint32_t n, Fr* den_inv_inout) {
  perm_num_kernel<<<blocks_for(n), 256, 0, c->stream>>>(a, n, den_inv_inout);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "perm_num launch failed");
}

// ---- exclusive multiplicative scan ---------------------------------------------------------------------
constexpr uint32_t SCAN_PER_LANE = 4;
constexpr uint32_t SCAN_TILE = 256 * SCAN_PER_LANE;

static __device__ __forceinline__ void sh_put(uint4* lo, uint4* hi, uint32_t t, const Fr& v) {
  lo[t] = make_uint4(v.v.l[0], v.v.l[1], v.v.l[2], v.v.l[3]);
  hi[t] = make_uint4(v.v.l[4], v.v.l[5], v.v.l[6], v.v.l[7]);
}
static __device__ __forceinline__ Fr sh_get(const uint4* lo, const uint4* hi, uint32_t t) {
  const uint4 a = lo[t], b = hi[t];
  Fr r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  return r;
}
// inclusive product scan of one value per lane across the 256-lane block (Hillis-Steele in LDS)
static __device__ __forceinline__ Fr block_scan_inclusive(Fr v, uint4* lo, uint4* hi) {
  const uint32_t t = threadIdx.x;
#pragma unroll 1
  for (uint32_t d = 1; d < 256; d <<= 1) {
    sh_put(lo, hi, t, v);
    __syncthreads();
    if (t >= d) v = sh_get(lo, hi, t - d) * v;
    __syncthreads();
  }
  return v;
}

// phase 1: product of each tile
__global__ __launch_bounds__(256) void scan_tile_kernel(const Fr* __restrict__ in, uint32_t n, uint32_t ntiles,
                                                        Fr* __restrict__ partial) {
  __shared__ uint4 lo[256], hi[256];
  const uint32_t t = threadIdx.x;
  const Fr* row = in + (size_t)blockIdx.y * n;
  const uint32_t base = blockIdx.x * SCAN_TILE + t * SCAN_PER_LANE;
  Fr v = Fr::one();
  for (uint32_t q = 0; q < SCAN_PER_LANE; q++)
    if (base + q < n) v = v * ld(row + base + q);
  v = block_scan_inclusive(v, lo, hi);
  if (t == 255) st(partial + (size_t)blockIdx.y * ntiles + blockIdx.x, v);
}
// phase 2: exclusive scan of the tile products of one row, 256 at a time with a running carry
__global__ __launch_bounds__(256) void scan_spine_kernel(Fr* __restrict__ partial, uint32_t ntiles) {
  __shared__ uint4 lo[256], hi[256];
  const uint32_t t = threadIdx.x;
  Fr* row = partial + (size_t)blockIdx.x * ntiles;
  Fr carry = Fr::one();
  for (uint32_t base = 0; base < ntiles; base += 256) {
    Fr v = base + t < ntiles ? ld(row + base + t) : Fr::one();
    const Fr inc = block_scan_inclusive(v, lo, hi);
    sh_put(lo, hi, t, inc);
    __syncthreads();
    const Fr exc = t ? sh_get(lo, hi, t - 1) : Fr::one();
    const Fr total = sh_get(lo, hi, 255);
    __syncthreads();
    if (base + t < ntiles) st(row + base + t, carry * exc);
    carry = carry * total;
  }
}
// phase 3: out[i] = tile prefix * lane prefix * running product inside the lane
__global__ __launch_bounds__(256) void scan_apply_kernel(const Fr* __restrict__ in, uint32_t n, uint32_t ntiles,
                                                         const Fr* __restrict__ partial, Fr* __restrict__ out) {
  __shared__ uint4 lo[256], hi[256];
  const uint32_t t = threadIdx.x;
  const Fr* row = in + (size_t)blockIdx.y * n;
  Fr* orow = out + (size_t)blockIdx.y * n;
  const uint32_t base = blockIdx.x * SCAN_TILE + t * SCAN_PER_LANE;
  Fr e[SCAN_PER_LANE];
  Fr v = Fr::one();
  for (uint32_t q = 0; q < SCAN_PER_LANE; q++) {
    e[q] = base + q < n ? ld(row + base + q) : Fr::one();
    v = v * e[q];
  }
  const Fr inc = block_scan_inclusive(v, lo, hi);
  sh_put(lo, hi, t, inc);
  __syncthreads();
  Fr run = t ? sh_get(lo, hi, t - 1) : Fr::one();
  run = run * ld(partial + (size_t)blockIdx.y * ntiles + blockIdx.x);
  for (uint32_t q = 0; q < SCAN_PER_LANE; q++) {
    if (base + q < n) st(orow + base + q, run);
    run = run * e[q];
  }
}

int prefix_product(cq_ctx* c, const Fr* in, Fr* out, uint32_t n, uint32_t batch) {
  if (batch == 0 || n == 0) return CQ_OK;
  const uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  void* scr;
  int rc;
  if ((rc = c->ensure_scratch(5, (size_t)batch * ntiles * sizeof(Fr), &scr)) != CQ_OK) return rc;
  Fr* partial = (Fr*)scr;
  scan_tile_kernel<<<dim3(ntiles, batch), 256, 0, c->stream>>>(in, n, ntiles, partial);
  scan_spine_kernel<<<batch, 256, 0, c->stream>>>(partial, ntiles);
  scan_apply_kernel<<<dim3(ntiles, batch), 256, 0, c->stream>>>(in, n, ntiles, partial, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "prefix_product launch failed");
}

// z_s[i] *= mult[s] for i < rows (the chain of last_z values across sets, permutation/prover.rs:89-90,173)
__global__ void perm_scale_kernel(Fr* __restrict__ z, uint32_t n, uint32_t rows, PermScaleArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  Fr* p = z + (size_t)blockIdx.y * n + i;
  st(p, ld(p) * a.mult[blockIdx.y]);
}

int perm_scale(cq_ctx* c, Fr* z, uint32_t n, uint32_t rows, uint32_t sets, const PermScaleArgs& a) {
  perm_scale_kernel<<<dim3(blocks_for(rows), sets), 256, 0, c->stream>>>(z, n, rows, a);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "perm_scale launch failed");
}

// ---- permutation terms of the quotient numerator (evaluation.rs:367-459) -----------------------------------
__global__ __launch_bounds__(256) void perm_h_kernel(PermHArgs a, Fr* __restrict__ h) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ext) return;
  const uint32_t r_next = rot_idx(i, 1, a.rot_scale, a.ext);
  const uint32_t r_last = rot_idx(i, -(int32_t)a.last_rot, a.rot_scale, a.ext);
  const Fr one = Fr::one();
  const Fr l0 = ld(a.l0 + i), ll = ld(a.l_last + i), la = ld(a.l_active + i);
  Fr v = ld(h + i);
  {
    const Fr z0 = ld(a.z + i);
    v = v * a.y + (one - z0) * l0;  // l_0(X) (1 - z_0(X))
    const Fr zl = ld(a.z + (size_t)(a.sets - 1) * a.ext + i);
    v = v * a.y + (zl * zl - zl) * ll;  // l_last(X) (z_l(X)^2 - z_l(X))
  }
  for (uint32_t s = 1; s < a.sets; s++)  // l_0(X) (z_i(X) - z_{i-1}(omega^last X))
    v = v * a.y + (ld(a.z + (size_t)s * a.ext + i) - ld(a.z + (size_t)(s - 1) * a.ext + r_last)) * l0;
  Fr current_delta = a.delta_start * (ld(a.ext_pow_hi + (i >> 8)) * ld(a.ext_pow_lo + (i & 255)));  // beta zeta omega_ext^i
  uint32_t ci = 0;
  for (uint32_t s = 0; s < a.sets; s++) {
    const Fr* zs = a.z + (size_t)s * a.ext;
    Fr left = ld(zs + r_next), right = ld(zs + i);
    const uint32_t cnt = min(a.chunk_len, a.ncols - ci);
    for (uint32_t j = 0; j < cnt; j++, ci++) {
      const Fr cv = ld(a.col[ci] + i);
      left = left * (cv + a.beta * ld(a.sigma + (size_t)ci * a.ext + i) + a.gamma);
      right = right * (cv + current_delta + a.gamma);
      current_delta = current_delta * a.delta;
    }
    v = v * a.y + (left - right) * la;
  }
  st(h + i, v);
}

// ---- legacy (plookup-style) lookup argument --------------------------------------------------------------------
// den[i] = (beta + a'[i]) (gamma + s'[i])   (lookup/prover.rs:200-209)
__global__ __launch_bounds__(256) void lookup_den_kernel(const Fr* __restrict__ a, const Fr* __restrict__ s, Fr beta, Fr gamma, uint32_t n,
                                                         Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(out + i, (beta + ld(a + i)) * (gamma + ld(s + i)));
}
// v[i] *= (A[i] + beta) (S[i] + gamma), A / S the compressed input / table expressions   (:213-221)
__global__ __launch_bounds__(256) void lookup_num_kernel(const Fr* __restrict__ cin, const Fr* __restrict__ ctab, Fr beta, Fr gamma, uint32_t n,
                                                         Fr* __restrict__ v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st(v + i, ld(v + i) * (ld(cin + i) + beta) * (ld(ctab + i) + gamma));
}
int lookup_denominators(cq_ctx* c, const Fr* a, const Fr* s, const Fr& beta, const Fr& gamma, uint32_t n, Fr* out) {
  lookup_den_kernel<<<blocks_for(n), 256, 0, c->stream>>>(a, s, beta, gamma, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "lookup_den launch failed");
}
int lookup_numerators(cq_ctx* c, const Fr* cin, const Fr* ctab, const Fr& beta, const Fr& gamma, uint32_t n, Fr* inout) {
  lookup_num_kernel<<<blocks_for(n), 256, 0, c->stream>>>(cin, ctab, beta, gamma, n, inout);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "lookup_num launch failed");
}

// the five lookup constraints of evaluate_h (evaluation.rs:461-531)
__global__ __launch_bounds__(256) void lookup_h_kernel(LookupHArgs a, Fr* __restrict__ h) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ext) return;
  const uint32_t r_next = rot_idx(i, 1, a.rot_scale, a.ext), r_prev = rot_idx(i, -1, a.rot_scale, a.ext);
  const Fr one = Fr::one();
  const Fr l0 = ld(a.l0 + i), ll = ld(a.l_last + i), la = ld(a.l_active + i);
  const Fr z = ld(a.z + i), pa = ld(a.a + i), ps = ld(a.s + i);
  const Fr table_value = (ld(a.cin + i) + a.beta) * (ld(a.ctab + i) + a.gamma);
  const Fr a_minus_s = pa - ps;
  Fr v = ld(h + i);
  v = v * a.y + (one - z) * l0;                                                      // l_0 (1 - z)
  v = v * a.y + (z * z - z) * ll;                                                    // l_last (z^2 - z)
  v = v * a.y + (ld(a.z + r_next) * (pa + a.beta) * (ps + a.gamma) - z * table_value) * la;
  v = v * a.y + a_minus_s * l0;                                                      // l_0 (a' - s')
  v = v * a.y + a_minus_s * (pa - ld(a.a + r_prev)) * la;                           // (a' - s')(a' - a'(w^-1 X))
  st(h + i, v);
}
int lookup_h_terms(cq_ctx* c, const LookupHArgs& a, Fr* h) {
  lookup_h_kernel<<<blocks_for(a.ext), 256, 0, c->stream>>>(a, h);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "lookup_h launch failed");
}

__global__ void fr_to_canonical_kernel(const Fr* __restrict__ in, uint32_t n, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U256 c = ld(in + i).to_canonical();
  uint4* q = reinterpret_cast<uint4*>(out + 4 * (size_t)i);
  q[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
  q[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
}
__global__ void fr_from_canonical_kernel(const uint64_t* __restrict__ in, uint32_t n, Fr* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(in + 4 * (size_t)i);
  const uint4 a = q[0], b = q[1];
  U256 c;
  c.l[0] = a.x; c.l[1] = a.y; c.l[2] = a.z; c.l[3] = a.w; c.l[4] = b.x; c.l[5] = b.y; c.l[6] = b.z; c.l[7] = b.w;
  st(out + i, Fr::from_canonical(c));
}
int fr_to_canonical(cq_ctx* c, const Fr* in, uint32_t n, uint64_t* out) {
  if (n) fr_to_canonical_kernel<<<blocks_for(n), 256, 0, c->stream>>>(in, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "to_canonical launch failed");
}
int fr_from_canonical(cq_ctx* c, const uint64_t* in, uint32_t n, Fr* out) {
  if (n) fr_from_canonical_kernel<<<blocks_for(n), 256, 0, c->stream>>>(in, n, out);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "from_canonical launch failed");
}

int perm_h_terms(cq_ctx* c, const PermHArgs& a, Fr* h) {
  perm_h_kernel<<<blocks_for(a.ext), 256, 0, c->stream>>>(a, h);
  return hipGetLastError() == hipSuccess ? CQ_OK : c->fail(CQ_ERR_HIP, "perm_h launch failed");
}

}  // namespace cq
