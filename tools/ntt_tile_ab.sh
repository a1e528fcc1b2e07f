#!/bin/bash
# Tile size / workgroup size of the NTT pass kernel, same box: 1024 x 256 (default), 512 x 128, 512 x 256, 2048 x 512, 2048 x 256
set -e
for flags in "-DCQ_DEFAULT" "-DCQ_NTT_TILE_ELEMS=512 -DCQ_NTT_THREADS=128" "-DCQ_NTT_TILE_ELEMS=512 -DCQ_NTT_THREADS=256" "-DCQ_NTT_TILE_ELEMS=2048 -DCQ_NTT_THREADS=512" "-DCQ_NTT_TILE_ELEMS=2048 -DCQ_NTT_THREADS=256"; do
  CQ_BUILD_JOBS=12 CQ_EXTRA_HIPCC_FLAGS="$flags" python3 sha2_on_cq_halo2_amd/build.py --force > /dev/null
  echo "== $flags"
  python3 tools/ntt_perf.py 18 8 40; python3 tools/ntt_perf.py 18 8 40; python3 tools/ntt_perf.py 18 4 40; python3 tools/ntt_perf.py 20 4 20; python3 tools/ntt_perf.py 22 2 8
  python3 tools/prove_large.py 18 | grep prove | tail -2
  python3 tools/prove_large.py 20 | grep prove | tail -1
done
