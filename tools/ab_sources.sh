#!/bin/bash
# Same-box A/B of two variants of some csrc files (on the GPU box): ab_tmp/old/* and ab_tmp/new/* are copied over
# sha2_on_cq_halo2_amd/csrc in turn, the library is rebuilt, and bench.py's headline numbers are printed (two rounds).
#   mkdir -p ab_tmp/old ab_tmp/new; git show HEAD:sha2_on_cq_halo2_amd/csrc/msm.hip > ab_tmp/old/msm.hip; cp ... ab_tmp/new/
#   gpurun -- 'bash tools/ab_sources.sh > gpurun_out/ab.txt'
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
for v in old new; do
  cp ab_tmp/$v/* sha2_on_cq_halo2_amd/csrc/
  CQ_BUILD_JOBS=12 python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $v"
  for k in ${AB_KS:-16 18}; do python3 tools/prove_large.py $k | grep prove | tail -2; done
  python3 bench.py --no-extra-legs --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=sorted(d['step_ms_all_this_rank'])
print('bench mean %.3f  median %.3f  min %.3f  p25 %.3f  p75 %.3f' % (d['ms_per_step'], t[len(t)//2], t[0], t[len(t)//4], t[3*len(t)//4]), ' acc avg_launch_ms %.4f' % r['avg_launch_ms'], 'valu_frac %.3f' % r['valu_frac'])"
done
done
