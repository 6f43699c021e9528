cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cb in 0 8 10 13 16 20 26; do if [ $cb = 0 ]; then unset RK_SKETCH_CB; else export RK_SKETCH_CB=$cb; fi; echo "cb $cb: $(timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"; done
export RK_SKETCH_CB=13
RK_SCAN2_TRACE=$GRAFT_REPO_ROOT/gpurun_out/scan_trace.bin timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1
python3 tools/trace_scan.py gpurun_out/scan_trace.bin
unset RK_SKETCH_CB
echo "img1: $(RK_SKETCH_IMG=1 timeout -k 10 300 python3 tools/prof_driver.py sketch 128 5000000 2>&1 | grep -v amdgpu.ids | tail -1)"
