#!/bin/bash
# Tuning sweep (run on the GPU box): rebuild the library with another MSM sub-list length and time a k=18/k=20 proof.
set -e
for s1 in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="-DCQ_MSM_S1=$s1" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== MSM_S1=$s1"
  python3 tools/prove_large.py 18 | tail -3 | head -2
  python3 tools/prove_large.py 20 | tail -2 | head -1
done
