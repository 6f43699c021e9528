cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for p in 2 0; do
RK_SCAN2_PRIO=$p RK_SCAN2_TRACE=$GRAFT_REPO_ROOT/gpurun_out/scan_trace.bin timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1
python3 tools/trace_scan.py gpurun_out/scan_trace.bin | tail -10
done
