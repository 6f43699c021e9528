cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q -k "near or self_join or order or alldist" > gpurun_out/t5.log 2>&1 || { tail -50 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 200 python3 tools/prof_driver.py dist 50000 20 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1
bash tools/kernel_trace.sh ktn dist 10000 20 1 0 0 > /dev/null 2>&1; f=$(find gpurun_out/ktn -name "*kernel_stats.csv" | head -1); grep "rk_" $f | cut -d, -f1-5 | cut -c1-150
