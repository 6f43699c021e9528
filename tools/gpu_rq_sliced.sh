#!/bin/bash
# the sliced membership pass (RK_DISTQ_SLICED=1): its parity test, configs[4]'s shape against the fused kernel, the two kernels' times
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "sliced" > gpurun_out/rq_tests_sliced.log 2>&1 || { tail -40 gpurun_out/rq_tests_sliced.log; exit 1; }
tail -2 gpurun_out/rq_tests_sliced.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
echo "sliced 1 configs4: $(RK_DISTQ_SLICED=1 drv dist_rq_dev 100000 1000 5)"
echo "sliced 0 configs4: $(RK_DISTQ_SLICED=0 drv dist_rq_dev 100000 1000 5)"
RK_DISTQ_SLICED=1 bash tools/kernel_trace.sh prof_rq dist_rq_dev 100000 1000 5 2>&1 | grep -E "k_member|rk_distq"
