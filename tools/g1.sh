cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/prof_driver.py index 10000 4 2>&1 | grep -v amdgpu.ids | tail -7
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_multirank_gloo.py tests/test_gpu_stress.py -m gpu -x -q > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
for o in 0 1; do timeout -k 10 200 python3 tools/prof_driver.py dist 10000 50 1 0 $o 2>&1 | grep -v amdgpu.ids | tail -1; done
