#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile_self_join or without_slice or crowded" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
export RK_DIST_TILES=1
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES:-1} | cut -c1-110; }
echo "10k: $(RK_DIST_DEBUG=1 LINES=2 drv dist 10000 20)"
echo "c100: $(RK_DIST_DEBUG=1 LINES=2 drv dist 10000 20 1 0 0 100)"
echo "c1000: $(RK_DIST_DEBUG=1 LINES=2 drv dist 10000 20 1 0 0 1000)"
echo "50k: $(RK_DIST_DEBUG=1 LINES=2 drv dist 50000 10)"
bash tools/kernel_trace.sh prof_tb dist 10000 3 > gpurun_out/tile_build_kernels.txt 2>&1; grep -v "rk_near\|bucket_emit\|part_\|row_\|minhash\|heads\|rank_keys\|cluster" gpurun_out/tile_build_kernels.txt | tail -22
