#!/bin/bash
# Where does a stand-alone NTT's time go?  Rebuilds the library with parts of the pass kernel switched off (WRONG results,
# timing only) and times batches.  Measured (2^18 x 8, 0.237 ms): load + store skeleton 0.076 ms (HBM-bound: 384 MB), the
# store phase's twiddle products +0.04, the butterflies +0.12 -- their plain sum: a 79 us pass is two rounds of four
# workgroups per CU that start in step, so memory phases and arithmetic barely overlap.
set -e
for flags in "-DCQ_DEFAULT" "-DCQ_NTT_EXP_NOSTAGES" "-DCQ_NTT_EXP_NOTW" "-DCQ_NTT_EXP_NOSTAGES -DCQ_NTT_EXP_NOTW"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/ntt_perf.py 18 8 40; python3 tools/ntt_perf.py 18 8 40; python3 tools/ntt_perf.py 20 4 20
done
