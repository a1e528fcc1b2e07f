#!/bin/bash
# A/B of the NTT's radix-4 steps against level-by-level butterflies, same box (rebuilds the library twice)
set -e
for flags in "${@:-"-DCQ_NTT_RADIX2" "-DCQ_NTT_RADIX4_DEFAULT"}"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  for rep in 1 2; do python3 tools/ntt_perf.py 18 8 40; python3 tools/ntt_perf.py 18 4 40; python3 tools/ntt_perf.py 20 4 20; python3 tools/ntt_perf.py 22 2 8; done
  python3 tools/prove_large.py 20 | grep prove | tail -2
  python3 tools/prove_large.py 22 | grep prove | tail -2
done
