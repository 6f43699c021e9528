#!/bin/bash
# the query kernel: parity tests (also with the sliced membership pass forced on small shapes), then configs[4]'s shape and related small sketches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_wide_hashes.py tests/test_reference_layout.py -x -q -k "quer or rq or dist or distq or ref" > gpurun_out/rq_tests.log 2>&1 || { tail -40 gpurun_out/rq_tests.log; exit 1; }
tail -2 gpurun_out/rq_tests.log
RK_DISTQ_SLICED=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_reference_binding.py tests/test_cli.py -x -q -k "quer or rq or dist or distq or ref" > gpurun_out/rq_tests_sliced.log 2>&1 || { tail -40 gpurun_out/rq_tests_sliced.log; exit 1; }
tail -2 gpurun_out/rq_tests_sliced.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "config4 or queries" > gpurun_out/rq_tests2.log 2>&1 || { tail -40 gpurun_out/rq_tests2.log; exit 1; }
tail -2 gpurun_out/rq_tests2.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for d in "" 0; do
  echo "sliced '$d' configs4: $(RK_DISTQ_SLICED=$d drv dist_rq_dev 100000 1000 5)"
done
echo "sliced 1 10k x 10k related: $(RK_DISTQ_SLICED=1 drv dist_rq_dev 10000 10000 5 1220 1220 28)"
echo "sliced 0 10k x 10k related: $(RK_DISTQ_SLICED=0 drv dist_rq_dev 10000 10000 5 1220 1220 28)"
