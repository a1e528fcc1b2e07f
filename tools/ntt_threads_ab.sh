set -e
for flags in "-DCQ_NTT_THREADS=256" "-DCQ_NTT_THREADS=512" "-DCQ_NTT_THREADS=128"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/ntt_perf.py 18 8 20; python3 tools/ntt_perf.py 18 4 20; python3 tools/ntt_perf.py 20 4 10
  python3 tools/prove_large.py 18 | tail -2 | head -1
  python3 tools/prove_large.py 20 | tail -2 | head -1
done
