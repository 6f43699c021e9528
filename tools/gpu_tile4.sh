#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile or near_window or crowded or random_alldist" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/(events.*hits/hits/;s/(row_step.*)//'; }
for c in 10 100 1000; do echo "clade $c: $(RK_DIST_TILES=1 drv dist 10000 30 1 0 0 $c)"; done
echo "50k: $(RK_DIST_TILES=1 drv dist 50000 10)"
echo "tiny: $(RK_DIST_TILES=1 drv dist 10000 30 1 0 0 10 1 0)"
