set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rq0 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py dist_rq > $GRAFT_REPO_ROOT/gpurun_out/prof_rq0.log 2>&1 )
cat gpurun_out/prof_rq0.log | tail -5
head -12 gpurun_out/prof_rq0/*/run_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_rq0 -name '*kernel_stats.csv' | head
