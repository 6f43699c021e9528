#!/bin/bash
# tile kernel: threads per tile x collection, and row shards (tile vs near)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/(events.*hits/hits/'; }
for t in 256 512 1024; do
  for c in 10 100 1000; do echo "T $t clade $c: $(RK_TILE_THREADS=$t RK_DIST_TILES=1 drv dist 10000 20 1 0 0 $c)"; done
  echo "T $t 50k: $(RK_TILE_THREADS=$t RK_DIST_TILES=1 drv dist 50000 10)"
done
for rs in 2 4 8; do
  echo "10k shard 1/$rs tiles: $(RK_DIST_TILES=1 drv dist 10000 20 $rs 32)"
  echo "10k shard 1/$rs near : $(RK_DIST_TILES=0 drv dist 10000 20 $rs 32)"
  echo "50k shard 1/$rs tiles: $(RK_DIST_TILES=1 drv dist 50000 10 $rs 32)"
  echo "50k shard 1/$rs near : $(RK_DIST_TILES=0 drv dist 50000 10 $rs 32)"
done
