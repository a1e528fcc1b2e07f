cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
for q in 0 1; do
  echo "== CQ_MSM_QUAD=$q"
  if [ $q = 0 ]; then export CQ_MSM_QUAD=0; else unset CQ_MSM_QUAD; fi
  python3 bench.py --no-extra-legs --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=sorted(d['step_ms_all_this_rank'])
print('bench mean %.3f  median %.3f  min %.3f' % (d['ms_per_step'], t[len(t)//2], t[0]))"
  for k in 14 16; do python3 tools/prove_large.py $k | grep prove | tail -1; done
done
done
