#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile_self_join or without_slice or crowded" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
export RK_DIST_TILES=1
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES:-1} | cut -c1-60; }
for ms in 1 4 8 16; do
echo "min share $ms: 10k $(RK_TILE_MIN_SHARE=$ms drv dist 10000 20) | c100 $(RK_TILE_MIN_SHARE=$ms drv dist 10000 20 1 0 0 100) | c1000 $(RK_TILE_MIN_SHARE=$ms drv dist 10000 20 1 0 0 1000) | 50k $(RK_TILE_MIN_SHARE=$ms drv dist 50000 10)"
done
