#!/bin/bash
# tile kernel ablations (RK_TILE_DEBUG: 1 no counting, 2 no evaluation, 4 empty workgroups, 8 no adds, 16 no merge/extraction, 32 / 64 only the first 512 / 256 tiles, 256 a carry of weight 8 per eight masks instead of one of weight 16 per sixteen)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile or near_window or crowded or random_alldist" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/(events.*hits/hits/;s/(row_step.*)//'; }
for d in 0; do
  echo "debug $d 10k: $(RK_TILE_DEBUG=$d RK_DIST_TILES=1 drv dist 10000 30)"
  echo "debug $d clade1000: $(RK_TILE_DEBUG=$d RK_DIST_TILES=1 drv dist 10000 30 1 0 0 1000)"
  echo "debug $d 50k: $(RK_TILE_DEBUG=$d RK_DIST_TILES=1 drv dist 50000 10)"
done
