#!/bin/bash
# Same-box A/B of hipcc -D flag sets (run on the GPU box): rebuilds per set, alternating, and prints proof-time medians at
# k = 14 / 16 / 18 / 20, the batched rate, and the in-proof averages of a few kernels.
#   bash tools/ab_flags_proofs.sh "-DCQ_CRIT_PRIO=0" "-DCQ_CRIT_PRIO=3"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for round in 1 2; do for flags in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
for k in (14, 16, 18, 20):
    ctx = Context(0); wl = ShaCqWorkload(ctx, k)
    ts = []
    for i in range(62 if k < 20 else 22):
        t = time.time(); wl.prove(seed=1); ts.append((time.time() - t) * 1e3)
    ts = sorted(ts[2:]); print("k=%d median %.3f min %.3f ms" % (k, ts[len(ts)//2], ts[0]), flush=True)
    wl.close() if hasattr(wl, "close") else None
PY
  python3 tools/batch_lanes.py 18 20 | tail -2
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/18 -o run -- python3 tools/prove_large.py 18 > gpurun_out/ab/log.txt 2>&1
  python3 tools/kstats.py gpurun_out/ab/18 | grep -E "batch_invert|combine_level_kernel<2|rowcol_kernel<16|weighted_quad|part_scatter|bucket_place"
  rm -rf gpurun_out/ab/18
done; done
