#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for d in 0 1; do
  RK_MEMBER_DEBUG=$d bash tools/kernel_trace.sh prof_rq_d$d dist_rq_dev 100000 1000 3 2>&1 | grep -E "k_member|rk_distq" | sed "s/^/debug $d: /"
done
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for d in "" 0; do
  echo "sliced '$d' configs4: $(RK_DISTQ_SLICED=$d drv dist_rq_dev 100000 1000 5)"
done
