#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile_self_join" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
export RK_DIST_TILES=1
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES:-1}; }
echo "10k: $(drv dist 10000 20)"
echo "10k 256: $(RK_TILE_THREADS=256 drv dist 10000 20)"
for c in 100 1000; do echo "clade $c: $(drv dist 10000 20 1 0 0 $c)"; done
echo "50k: $(drv dist 50000 10)"
echo "1/8 shard rb64: $(drv dist 10000 20 8 64)"
echo "1/8 shard rb64 1024: $(RK_TILE_THREADS=1024 drv dist 10000 20 8 64)"
echo "1/2 shard rb64: $(drv dist 10000 20 2 64)"
echo "1/8 shard 50k rb64: $(drv dist 50000 20 8 64)"
echo "1/8 shard 50k rb64 512: $(RK_TILE_THREADS=512 drv dist 50000 20 8 64)"
echo "near 1/8 shard 50k: $(RK_DIST_TILES=0 drv dist 50000 20 8 64)"
