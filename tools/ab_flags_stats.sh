#!/bin/bash
# A/B of hipcc -D flags with per-kernel times (run on the GPU box): rebuilds the library for every flag set, proves at
# k = 18 and k = 22 under rocprofv3 --stats and prints the MSM kernels' average durations.
#   bash tools/ab_flags_stats.sh "-DCQ_NO_SQR" "-DCQ_SQR_ON"
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for flags in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  for k in 18 22; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/$k -o run -- python3 tools/prove_large.py $k > gpurun_out/ab/log.txt 2>&1
    grep "prove" gpurun_out/ab/log.txt | tail -2
    python3 tools/kstats.py gpurun_out/ab/$k | grep -E "accumulate|combine_level_kernel|rowcol_kernel|ntt_pass"
    rm -rf gpurun_out/ab/$k
  done
done
