#!/bin/bash
# self join on collections that do not look like the clade-of-10 generator: wide species, tiny sketches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for c in 10 100 1000; do
  echo "clade $c near: $(drv dist 10000 20 1 0 0 $c)"
  echo "clade $c full: $(RK_DIST_NEAR=0 drv dist 10000 20 1 0 0 $c)"
done
echo "tiny 1 jaccard: $(drv dist 10000 20 1 0 0 10 1 0)"
echo "tiny 1 contain: $(drv dist 10000 20 1 0 0 10 1 1)"
echo "tiny 0 contain: $(drv dist 10000 20 1 0 0 10 0 1)"
echo "tiny 50 jaccard: $(drv dist 10000 20 1 0 0 10 50 0)"
