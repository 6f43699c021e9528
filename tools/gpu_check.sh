set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 200 python3 tools/prof_driver.py dist_rq > gpurun_out/rq.log 2>&1 || { tail -20 gpurun_out/rq.log; exit 1; }
cat gpurun_out/rq.log
timeout -k 10 200 python3 tools/prof_driver.py sketch > gpurun_out/sk.log 2>&1 || { tail -20 gpurun_out/sk.log; exit 1; }
cat gpurun_out/sk.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
