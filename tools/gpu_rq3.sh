#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for f in 0 4 8 16; do
  echo "min_fit $f configs4: $(RK_DISTQ_MIN_FIT=$f drv dist_rq_dev 100000 1000 5)"
  echo "min_fit $f 10k x 10k related: $(RK_DISTQ_MIN_FIT=$f drv dist_rq_dev 10000 10000 5 1220 1220 28)"
  echo "min_fit $f 1k x 1k: $(RK_DISTQ_MIN_FIT=$f drv dist_rq_dev 1000 1000 10 1220 1220 28)"
done
