#!/bin/bash
# the tile kernel's two variants (RK_TILE_SROW=1: row masks as 64-bit scalars, one v_cndmask per record; 0: both masks through LDS), alternating, three times each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/ ms.step.*//;s/dist //'; }
for cfg in "4000 60" "10000 60" "10000 60 1 0 0 100" "20000 40" "10000 60 1 0 0 1000" "50000 20" "10000 60 8 32" "10000 60 2 32"; do
  a=""; b=""
  for rep in 1 2 3; do
    a="$a $(RK_TILE_SROW=1 RK_DIST_TILES=1 drv dist $cfg)"
    b="$b $(RK_TILE_SROW=0 RK_DIST_TILES=1 drv dist $cfg)"
  done
  echo "[$cfg] srow:$a | lds:$b"
done
