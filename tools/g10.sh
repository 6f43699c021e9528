cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -m gpu -x -q -k "sketch" > gpurun_out/t10.log 2>&1 || { tail -40 gpurun_out/t10.log; exit 1; }
tail -2 gpurun_out/t10.log
for img in 2 1; do for cb in 0 38 77; do echo "img $img cb $cb"; if [ $cb = 0 ]; then unset RK_SKETCH_CB; else export RK_SKETCH_CB=$cb; fi; RK_SKETCH_IMG=$img timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -2; done; done
unset RK_SKETCH_CB
timeout -k 10 300 python3 tools/prof_driver.py sketch 1000 5000000 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 300 python3 tools/prof_driver.py sketch 37 3000000 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 300 python3 tools/prof_driver.py sketch 1 1000000000 2>&1 | grep -v amdgpu.ids | tail -1
