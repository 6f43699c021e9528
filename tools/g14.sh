cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_multirank_gloo.py -m gpu -x -q -k "near or alldist or self or shard or genome_order or config" > gpurun_out/t14.log 2>&1 || { tail -40 gpurun_out/t14.log; exit 1; }
tail -2 gpurun_out/t14.log
for sk in 1 0; do for s in 1 8; do echo "skip $sk S $s: $(RK_DIST_FB_SKIP=$sk timeout -k 10 120 python3 tools/prof_driver.py dist 10000 200 $s 16 2>&1 | grep -v amdgpu.ids | tail -1)"; done; echo "skip $sk 50k: $(RK_DIST_FB_SKIP=$sk timeout -k 10 200 python3 tools/prof_driver.py dist 50000 100 1 16 2>&1 | grep -v amdgpu.ids | tail -1)"; echo "skip $sk 50k S 8: $(RK_DIST_FB_SKIP=$sk timeout -k 10 200 python3 tools/prof_driver.py dist 50000 100 8 16 2>&1 | grep -v amdgpu.ids | tail -1)"; done
