cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do for p in 2 1 0; do echo "128 prio $p: $(RK_SCAN2_PRIO=$p timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
for p in 2 1 0; do echo "1000 prio $p: $(RK_SCAN2_PRIO=$p timeout -k 10 300 python3 tools/prof_driver.py sketch 1000 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; done
RK_SCAN2_PRIO=2 RK_SCAN2_TRACE=$GRAFT_REPO_ROOT/gpurun_out/scan_trace.bin timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1
python3 tools/trace_scan.py gpurun_out/scan_trace.bin | head -12
