# PMC passes over the stand-alone NTT (tools/ntt_perf.py 18 8): what the pass kernel's VALU issue utilisation is and what the
# waves wait for.  Separate --pmc runs, no trace domains beside them.  Output: gpurun_out/nttpmc/*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/nttpmc; mkdir -p $O
python3 tools/ntt_perf.py 18 8 20 > $O/plain.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES_EQ_64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/sq$i -o run -- python3 tools/ntt_perf.py 18 8 3 > $O/sq$i.log 2>&1 || echo "set $i failed"
done
python3 - <<'PY' > $O/ntt_counters.txt
import csv, glob, collections
rows=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/nttpmc/sq*/**/*counter_collection.csv", recursive=True)):
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "ntt_pass_kernel" not in r["Kernel_Name"]: continue
        k=int(r["Dispatch_Id"]); per.setdefault(k,{})
        per[k][r["Counter_Name"]]=per[k].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    # launches 4..6 of each run (one transform: three passes)
    ks=sorted(per)[3:6]
    for j,k in enumerate(ks):
        rows.setdefault(j,{}).update(per[k])
for j,c in rows.items():
    print("pass",j)
    for n,v in sorted(c.items()): print("   %-28s %14.0f" % (n,v))
PY
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/sq4
cat $O/plain.txt $O/ntt_counters.txt
