#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RK_DIST_TILES=1
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES:-1}; }
echo "10k: $(RK_DIST_DEBUG=1 LINES=2 drv dist 10000 20)"
echo "50k: $(RK_DIST_DEBUG=1 LINES=2 drv dist 50000 10)"
echo "c1000: $(RK_DIST_DEBUG=1 LINES=2 drv dist 10000 10 1 0 0 1000)"
bash tools/kernel_trace.sh prof_tb dist 10000 3 > gpurun_out/tile_build_kernels.txt 2>&1; tail -45 gpurun_out/tile_build_kernels.txt
