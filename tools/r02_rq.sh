set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not sketch" > gpurun_out/gpu_dist_tests.log 2>&1 || { tail -60 gpurun_out/gpu_dist_tests.log; exit 1; }
tail -3 gpurun_out/gpu_dist_tests.log
for i in 1 2; do
timeout -k 10 300 python3 tools/prof_driver.py dist_rq_dev 100000 1000 20 > gpurun_out/rq.log 2>&1 || { tail -20 gpurun_out/rq.log; exit 1; }
tail -2 gpurun_out/rq.log
done
