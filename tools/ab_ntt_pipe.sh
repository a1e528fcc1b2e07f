#!/bin/bash
# A/B of the pipelined NTT pass kernel on one GPU box: each argument is a set of hipcc -D flags; ntt.hip is rebuilt with
# them, parity is checked (tests/test_ntt_gpu.py) and stand-alone transforms are timed; CQ_NTT_PIPE=0 times the plain kernel.
set -e
export CQ_BUILD_JOBS=12
for flags in "$@"; do
  touch sha2_on_cq_halo2_amd/csrc/ntt.hip
  CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py > /dev/null
  echo "== flags: $flags"
  CQ_NTT_PIPE=1 python3 -m pytest tests/test_ntt_gpu.py -m gpu -x -q 2>&1 | tail -1
  for wg in 4 3; do
    echo "-- workgroups per CU in the persistent grid: $wg"
    CQ_NTT_PIPE=1 CQ_NTT_PIPE_WG_PER_CU=$wg python3 tools/ntt_perf.py 18 8 20
    CQ_NTT_PIPE=1 CQ_NTT_PIPE_WG_PER_CU=$wg python3 tools/ntt_perf.py 20 8 10
    CQ_NTT_PIPE=1 CQ_NTT_PIPE_WG_PER_CU=$wg python3 tools/ntt_perf.py 19 8 10
  done
  echo "-- half-size tiles (CQ_NTT_PIPE_TILE=512)"
  CQ_NTT_PIPE=1 CQ_NTT_PIPE_TILE=512 python3 -m pytest tests/test_ntt_gpu.py -m gpu -x -q 2>&1 | tail -1
  CQ_NTT_PIPE=1 CQ_NTT_PIPE_TILE=512 python3 tools/ntt_perf.py 18 8 20
  CQ_NTT_PIPE=1 CQ_NTT_PIPE_TILE=512 python3 tools/ntt_perf.py 20 8 10
done
echo "== plain kernel (CQ_NTT_PIPE=0)"
CQ_NTT_PIPE=0 python3 tools/ntt_perf.py 18 8 20
CQ_NTT_PIPE=0 python3 tools/ntt_perf.py 20 8 10
touch sha2_on_cq_halo2_amd/csrc/ntt.hip
python3 sha2_on_cq_halo2_amd/build.py > /dev/null
