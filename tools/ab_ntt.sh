#!/bin/bash
# A/B on one GPU box: rebuild with each set of hipcc -D flags and time stand-alone NTTs of several sizes.
set -e
for flags in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/ntt_perf.py 18 8 10
  python3 tools/ntt_perf.py 22 4 3
  python3 tools/ntt_perf.py 24 1 3
done
