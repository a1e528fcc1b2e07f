set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f
mkdir -p $O
# 1. timed-region stats (headline workload only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/timed -o run -- python3 bench.py --steps 20 --warmup 5 --no-plonk-variant --no-in-flight --no-cpu-baseline --no-k20 --no-generic-rng > $O/timed_line.json 2> $O/timed.err
python3 tools/kstats.py $O/timed > $O/timed_summary.txt
cp $O/timed/run_kernel_stats.csv $O/timed_kernel_stats.csv
# 2. timeline of one proof
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o run -- python3 tools/prove_large.py 18 > $O/tl.log 2>&1
python3 tools/timeline.py $O/tl/run_kernel_trace.csv 8.5 > $O/timeline.txt
# 3. PMC passes (separate)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 tools/prove_large.py 18 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 tools/prove_large.py 18 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 tools/prove_large.py 18 > $O/pmc_sq.log 2>&1
ls $O
