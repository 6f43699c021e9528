cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_reference_layout.py -m gpu -x -q -k "not sketch" > gpurun_out/t5.log 2>&1 || { tail -50 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
for uw in 0 1 2; do echo "UW $uw"; 
RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1
RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 50000 20 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1
RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 10000 100 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1
RK_DIST_NEAR_UW=$uw timeout -k 10 200 python3 tools/prof_driver.py dist 50000 50 8 16 0 2>&1 | grep -v amdgpu.ids | tail -1
done > gpurun_out/uw.log 2>&1
cat gpurun_out/uw.log
for dbg in 0 1; do
RK_NEAR_DEBUG=$dbg bash tools/kernel_trace.sh ktn dist 10000 20 1 0 0 > /dev/null 2>&1; f=$(find gpurun_out/ktn -name "*kernel_stats.csv" | head -1); echo "debug $dbg: $(grep rk_near $f | cut -d, -f3-5)"
done
