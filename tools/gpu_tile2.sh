#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tile_self_join or crowded" > gpurun_out/tile_tests.log 2>&1 || { tail -40 gpurun_out/tile_tests.log; exit 1; }
tail -2 gpurun_out/tile_tests.log
export RK_DIST_TILES=1
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES:-1}; }
echo "10k: $(drv dist 10000 20)"
for d in 2 10; do echo "debug $d: $(RK_TILE_DEBUG=$d drv dist 10000 20)"; done
for t in 512 1024; do echo "threads $t: $(RK_TILE_THREADS=$t drv dist 10000 20)"; done
for c in 100 1000; do echo "clade $c: $(drv dist 10000 20 1 0 0 $c)"; echo "clade $c 512: $(RK_TILE_THREADS=512 drv dist 10000 20 1 0 0 $c)"; done
echo "50k: $(drv dist 50000 10)"
echo "50k 512: $(RK_TILE_THREADS=512 drv dist 50000 10)"
echo "tiny 50 contain: $(drv dist 10000 20 1 0 0 10 50 1)"
echo "1/8 shard rb64: $(drv dist 10000 20 8 64)"
