cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -14 gpurun_out/gpu_tests.log
