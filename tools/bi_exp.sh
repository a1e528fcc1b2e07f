# batch_invert_kernel: where its time goes (timing experiments with wrong results: -DCQ_BI_EXP=1 no inversion, =2 no unwinding;
# -DCQ_BI_CT=1: the constant-time safegcd on the lone lane instead of the variable-time one)
cd $GRAFT_REPO_ROOT
for flags in ${BI_FLAGS:-"-DCQ_BI_EXP=0" "-DCQ_BI_EXP=1" "-DCQ_BI_EXP=2" "-DCQ_BI_CT=1"}; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/batch_invert_perf.py; python3 tools/batch_invert_perf.py 65536; python3 tools/batch_invert_perf.py 4096
done
CQ_BUILD_JOBS=12 python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
timeout -k 10 600 python3 -m pytest tests/test_poly_gpu.py tests/test_rounds_gpu.py -x -q -m gpu 2>&1 | tail -2
