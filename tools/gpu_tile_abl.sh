#!/bin/bash
# tile kernel: one ablation switch against none on a few collections (RK_ABL: bits of RK_TILE_DEBUG; results may be wrong, times are not)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/ ms.step.*//;s/dist //'; }
for cfg in "10000 40 1 0 0 1000" "10000 40 1 0 0 100" "50000 20" "10000 60"; do
  a=""; b=""
  for rep in 1 2; do
    a="$a $(RK_DIST_TILES=1 drv dist $cfg)"
    b="$b $(RK_TILE_DEBUG=${RK_ABL:-4096} RK_DIST_TILES=1 drv dist $cfg)"
  done
  echo "[$cfg] plain:$a | debug ${RK_ABL:-4096}:$b"
done
