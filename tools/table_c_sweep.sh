#!/bin/bash
# Proof time against the window width of the precomputed MSM tables (CQ_TABLE_C), on the GPU box:
#   tools/table_c_sweep.sh "20 22" "15 17 18 19 20"
ks=${1:-"20 22"}
cs=${2:-"15 16 17 18 19 20"}
for k in $ks; do
  for c in $cs; do
    echo "== k=$k CQ_TABLE_C=$c"
    CQ_TABLE_C=$c python3 tools/prove_large.py $k | grep -E "prove|sha256" | tail -3
  done
done
