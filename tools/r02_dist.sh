set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not sketch" > gpurun_out/gpu_dist_tests.log 2>&1 || { tail -60 gpurun_out/gpu_dist_tests.log; exit 1; }
tail -3 gpurun_out/gpu_dist_tests.log
for n in 10000 10000 50000 28284; do
timeout -k 10 300 python3 tools/prof_driver.py dist $n 200 > gpurun_out/d$n.log 2>&1 || { tail -20 gpurun_out/d$n.log; exit 1; }
echo n $n; tail -1 gpurun_out/d$n.log
done
timeout -k 10 300 python3 tools/prof_driver.py dist 10000 200 8 16 > gpurun_out/d8.log 2>&1; tail -1 gpurun_out/d8.log
