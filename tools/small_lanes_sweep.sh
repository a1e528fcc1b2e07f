set -e
for flags in "-DCQ_MSM_SMALL_LANES=65536" "-DCQ_MSM_SMALL_LANES=131072" "-DCQ_MSM_SMALL_LANES=196608" "-DCQ_MSM_SMALL_LANES=393216"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  for k in 14 16 18; do python3 tools/prove_large.py $k | grep prove | tail -2; done
done
