cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_reference_layout.py -m gpu -x -q -k "not sketch" > gpurun_out/t5.log 2>&1 || { tail -50 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
for o in 0 1; do timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 1 0 $o 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout -k 10 200 python3 tools/prof_driver.py dist 50000 20 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 200 python3 tools/prof_driver.py dist 50000 50 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1
