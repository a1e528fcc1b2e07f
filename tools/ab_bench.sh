#!/bin/bash
# A/B on one GPU box: rebuild with each set of hipcc -D flags and print the bench line's key numbers.
set -e
for flags in "$@"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  for rep in 1 2; do
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-plonk-variant --no-in-flight --no-generic-rng 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$flags', 'ms_per_step %.3f' % d['ms_per_step'], 'acc_ms_per_proof %.3f' % (r['avg_launch_ms']*r['launches']/d['steps']), 'valu_frac %.3f' % r['valu_frac'])"
  done
done
