# index build timing + kernel timeline, with the developer ablations of k_bucket_emit (results are wrong under RK_INDEX_DEBUG, times are not)
cd $GRAFT_REPO_ROOT
python3 tools/prof_driver.py index 10000 8 2>&1 | grep "index build" | tail -3
for d in ${RK_ABL:-1 2 4}; do RK_INDEX_DEBUG=$d bash tools/kernel_trace.sh prof_index_d$d index 10000 4 2>&1 | grep k_bucket_emit | sed "s/^/debug $d: /"; done
bash tools/kernel_trace.sh prof_index index 10000 6 > gpurun_out/index_kernels.txt 2>&1 && python3 tools/index_timeline.py gpurun_out/prof_index
