# The commands behind profiles/r03_* (one gpurun call; everything lands under gpurun_out/r3prof and is then copied into
# profiles/ by hand).  PMC passes are separate runs, with no trace domains beside them.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3prof
mkdir -p $O
# 1. timed-region stats (headline workload only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/timed -o run -- python3 bench.py --steps 20 --warmup 5 --no-plonk-variant --no-in-flight --no-cpu-baseline --no-k20 --no-generic-rng --no-extra-legs > $O/r03_bench_timed_region_line.json 2> $O/timed.err
python3 tools/kstats.py $O/timed > $O/r03_bench_timed_region_summary.txt
cp $O/timed/run_kernel_stats.csv $O/r03_bench_timed_region_kernel_stats.csv
echo "timed region done"
# 2. timelines of one proof: k = 18 and k = 20
# (three traces: the one-proof summary of each, cut from its chronological dump; the cleanest is the one committed)
for i in 1 2 3; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tl -o run -- python3 tools/prove_large.py 18 > $O/tl.log 2>&1
  python3 tools/timeline_dump.py $O/tl/run_kernel_trace.csv 14.0 > $O/r03_timeline_k18_chronological_$i.txt
  python3 tools/timeline_one_proof.py $O/r03_timeline_k18_chronological_$i.txt > $O/r03_timeline_k18_proof_$i.txt
  [ $i -lt 3 ] && rm -rf $O/tl
done
rocprofv3 --kernel-trace --output-format csv -d $O/tl20 -o run -- python3 tools/prove_large.py 20 > $O/tl20.log 2>&1
python3 tools/timeline.py $O/tl20/run_kernel_trace.csv 26 > $O/r03_timeline_k20_proof.txt
echo "timelines done"
# 3. PMC passes (separate)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 tools/prove_large.py 18 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 tools/prove_large.py 18 > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r03_pmc_traffic_k18_proof.json > $O/pmc_traffic.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 tools/prove_large.py 18 > $O/pmc_sq.log 2>&1
python3 tools/pmc_sq.py $O/pmc_sq msm_accumulate_kernel $O/r03_pmc_sq_accumulate_k18.json "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python3 tools/prove_large.py 18" > $O/pmc_sq.txt
echo "pmc done"
# 4. the host's milestones of one proof, no profiler attached (what the gaps between rounds are made of)
CQ_TRACE_HOST=1 python3 tools/prove_large.py 18 > $O/host_trace.out 2> $O/r03_host_trace_k18.txt
# 5. the plain bench line on the same box
python3 bench.py --steps 20 --warmup 5 > $O/r03_bench_create_proof_k18_line_unprofiled.json 2> $O/bench.err
rm -rf $O/timed $O/tl $O/tl20 $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls $O
