#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
RK_DISTQ_SLICED=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_reference_binding.py -x -q -k "quer or rq or dist or distq or ref" > gpurun_out/rq_tests_sliced.log 2>&1 || { tail -40 gpurun_out/rq_tests_sliced.log; exit 1; }
tail -2 gpurun_out/rq_tests_sliced.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "config4 or queries" > gpurun_out/rq_tests2.log 2>&1 || { tail -40 gpurun_out/rq_tests2.log; exit 1; }
tail -2 gpurun_out/rq_tests2.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for d in "" 0; do
  echo "sliced '$d' configs4: $(RK_DISTQ_SLICED=$d drv dist_rq_dev 100000 1000 5)"
done
bash tools/kernel_trace.sh prof_rq dist_rq_dev 100000 1000 5 2>&1 | grep -E "k_member|rk_distq"
