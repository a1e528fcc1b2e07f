for per in 16 8 4; do
  echo "== CQ_BI_PER_LANE=$per"
  for k in 14 16 18 20; do CQ_BI_PER_LANE=$per python3 tools/prove_large.py $k | grep prove | tail -2; done
done
echo "== auto"; for k in 14 16 18 20; do python3 tools/prove_large.py $k | grep -E "prove|sha" | tail -2; done
