#!/bin/bash
# the tile kernel: parity tests, then timings against the row kernels on clades of 10 / 100 / 1,000 and tiny sketches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile_self_join or near_window or crowded or random_alldist" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for c in 10 100 1000; do
  echo "clade $c tiles: $(RK_DIST_TILES=1 drv dist 10000 20 1 0 0 $c)"
done
echo "tiny 1 jaccard tiles: $(RK_DIST_TILES=1 drv dist 10000 20 1 0 0 10 1 0)"
echo "tiny 50 contain tiles: $(RK_DIST_TILES=1 drv dist 10000 20 1 0 0 10 50 1)"
echo "50k tiles: $(RK_DIST_TILES=1 drv dist 50000 10)"
echo "50k near: $(drv dist 50000 10)"
echo "10k shuffled tiles: $(RK_DIST_TILES=1 drv dist 10000 20 1 0 1)"
echo "1/8 shard tiles rb64: $(RK_DIST_TILES=1 drv dist 10000 20 8 64)"
