// Internal interface of the host orchestrator (prover.hip).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/cq_halo2.h"

struct cq_pk;

namespace cq {
size_t prover_arena_elems(const cq_pk* pk);
int create_proof_dev(cq_pk* pk, const uint64_t* const* advice_dev, const uint64_t* const* instances,
                     const size_t* instance_lens, cq_phase_fn phase_fn, void* phase_user, cq_rng_next_u64 rng_next,
                     void* rng_state, std::vector<uint8_t>& proof_out);
}  // namespace cq
