#!/bin/bash
# A/B of the paired column-wise products (Fp29::mul_pair) on one GPU box: rebuilds with -DCQ_MUL_NO_PAIRS and without,
# NTT stand-alone rate, proofs at k = 18 / 20 / 22 (wall clock, no profiler).
#   bash tools/ab_pairs.sh > gpurun_out/pairs_ab.txt
set -e
cd $GRAFT_REPO_ROOT
for flags in "-DCQ_MUL_NO_PAIRS" "-DCQ_PAIRS_ON" "-DCQ_MUL_NO_PAIRS" "-DCQ_PAIRS_ON"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/ntt_perf.py 18 8 30
  python3 tools/ntt_perf.py 20 8 10
  for k in 18 20 22; do python3 tools/prove_large.py $k | grep prove | tail -2; done
done
