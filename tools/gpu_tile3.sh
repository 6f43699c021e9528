#!/bin/bash
# tile kernel ablations (RK_TILE_DEBUG: 1 no counting, 2 no evaluation, 4 empty workgroups, 8 no adds, 16 no merge/extraction, 32 / 64 only the first 512 / 256 tiles); results are wrong, times are not
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/(events.*hits/hits/;s/(row_step.*)//'; }
for d in ${RK_ABL:-0 32 64}; do
  echo "debug $d: $(RK_TILE_DEBUG=$d RK_DIST_DEBUG=0 RK_DIST_TILES=1 drv dist 10000 30)"
done
RK_DIST_DEBUG=1 RK_DIST_TILES=1 timeout -k 10 100 python3 tools/prof_driver.py dist 10000 2 2>&1 | grep "\[rk\]" | head -5
