cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -m gpu -x -q -k "sketch" > gpurun_out/t8.log 2>&1 || { tail -40 gpurun_out/t8.log; exit 1; }
tail -3 gpurun_out/t8.log
for img in 2 1; do RK_SKETCH_IMG=$img timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -3; done
