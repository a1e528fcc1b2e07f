# Diagnostic: rebuild with the NTT workgroup log and print the placement / timing summary (gpurun_out/ntt_wg_trace.txt).
cd $GRAFT_REPO_ROOT
CQ_EXTRA_HIPCC_FLAGS="-DCQ_NTT_TRACE" CQ_BUILD_JOBS=12 python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
mkdir -p gpurun_out
python3 tools/ntt_wg_trace.py 18 8 > gpurun_out/ntt_wg_trace.txt 2>&1
python3 tools/ntt_wg_trace.py 20 2 >> gpurun_out/ntt_wg_trace.txt 2>&1
cat gpurun_out/ntt_wg_trace.txt
