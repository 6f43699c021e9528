#!/bin/bash
# the query kernel: parity tests, then configs[4]'s shape and related small sketches, pipelined look-up (big rows) vs not (RK_DISTQ_PIPE=0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_wide_hashes.py tests/test_reference_layout.py -x -q -k "quer or rq or dist or distq or ref" > gpurun_out/rq_tests.log 2>&1 || { tail -40 gpurun_out/rq_tests.log; exit 1; }
tail -2 gpurun_out/rq_tests.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "config4 or queries" > gpurun_out/rq_tests2.log 2>&1 || { tail -40 gpurun_out/rq_tests2.log; exit 1; }
tail -2 gpurun_out/rq_tests2.log
drv() { timeout -k 10 300 python3 tools/prof_driver.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for d in 1 0; do
  echo "pipe $d configs4: $(RK_DISTQ_PIPE=$d drv dist_rq_dev 100000 1000 5)"
  echo "pipe $d 10k x 10k related: $(RK_DISTQ_PIPE=$d drv dist_rq_dev 10000 10000 5 1220 1220 28)"
  echo "pipe $d 1k x 1k: $(RK_DISTQ_PIPE=$d drv dist_rq_dev 1000 1000 5 1220 1220 28)"
done
