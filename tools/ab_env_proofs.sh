#!/bin/bash
# Same-box A/B of environment settings (no rebuild): alternates the settings, AB_ROUNDS rounds (2), proof-time medians per k.
#   AB_KS="18 20" bash tools/ab_env_proofs.sh "CQ_RANDOM_EARLY=0" "CQ_RANDOM_EARLY=1"
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${AB_ROUNDS:-2}); do for setting in "$@"; do
  echo "== $setting"
  env $setting AB_KS="${AB_KS:-14 16 18}" python3 - <<'PY'
import sys, time, os, hashlib
sys.path.insert(0, os.getcwd())
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
for k in [int(x) for x in os.environ["AB_KS"].split()]:
    ctx = Context(0); wl = ShaCqWorkload(ctx, k)
    ts = []
    for i in range(62 if k <= 18 else 22 if k == 20 else 10):
        t = time.time(); p = wl.prove(seed=1); ts.append((time.time() - t) * 1e3)
    ts = sorted(ts[2:]); print("k=%d median %.3f min %.3f ms  sha256 %s" % (k, ts[len(ts)//2], ts[0], hashlib.sha256(bytes(p)).hexdigest()[:12]), flush=True)
    wl.close(); ctx.close()
PY
done; done
