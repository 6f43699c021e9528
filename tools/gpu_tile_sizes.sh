#!/bin/bash
# whole-matrix self join at 500 .. 20,000 genomes: the tile kernel (a resident index from its second join on) against the near-window kernel -- the crossover behind RK_DIST_TILES_MIN_GENOMES
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/(events.*hits/hits/;s/(row_step.*)//'; }
for n in 500 1000 2000 4000 20000; do
  echo "n $n tiles: $(drv dist $n 30)"
  echo "n $n near : $(RK_DIST_TILES_AFTER=1000000 drv dist $n 30)"
done
