#!/bin/bash
# Tuning sweep (run on the GPU box): rebuild the library with other hipcc -D flags and time k=18 / k=20 / k=22 proofs.
#   bash tools/s1_sweep.sh "-DCQ_MSM_S1_BIG=32" "-DCQ_MSM_S1_BIG=64 -DCQ_MSM_S1_BIG_ENTRIES=8000000ull"
set -e
for flags in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/prove_large.py 18 | tail -3 | head -2
  python3 tools/prove_large.py 20 | tail -2 | head -1
  python3 tools/prove_large.py 22 | tail -2 | head -1
done
