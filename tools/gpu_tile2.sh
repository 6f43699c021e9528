#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-40; }
for rb in 16 32 64; do
echo "near rb $rb: 10k 1/8 $(drv dist 10000 40 8 $rb) | 10k 1/2 $(drv dist 10000 40 2 $rb) | 50k 1/8 $(drv dist 50000 20 8 $rb) | 50k 1/2 $(drv dist 50000 20 2 $rb)"
done
export RK_DIST_TILES=1
for rb in 16 32 64; do
echo "tile rb $rb: c100 1/8 $(drv dist 10000 40 8 $rb 0 100) | c100 1/2 $(drv dist 10000 40 2 $rb 0 100) | c1000 1/8 $(drv dist 10000 20 8 $rb 0 1000)"
done
