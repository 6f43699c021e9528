cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -m gpu -x -q -k "sketch" > gpurun_out/t13.log 2>&1 || { tail -40 gpurun_out/t13.log; exit 1; }
tail -1 gpurun_out/t13.log
for rep in 1 2 3; do echo "scan2: $(timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; echo "img1:  $(RK_SKETCH_IMG=1 timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; done
RK_SCAN2_TRACE=$GRAFT_REPO_ROOT/gpurun_out/scan_trace.bin timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1
python3 tools/trace_scan.py gpurun_out/scan_trace.bin | head -12
timeout -k 10 300 python3 tools/prof_driver.py sketch 1000 5000000 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 300 python3 tools/prof_driver.py sketch 1 1000000000 2>&1 | grep -v amdgpu.ids | tail -1
