cd $GRAFT_REPO_ROOT
for d in 0 64 65 1; do RK_INDEX_DEBUG=$d bash tools/kernel_trace.sh prof_index_d$d index 10000 4 2>&1 | grep k_bucket_emit | sed "s/^/debug $d: /"; done
for t in 256 1024; do RK_INDEX_EMIT_T=$t bash tools/kernel_trace.sh prof_index_t$t index 10000 4 2>&1 | grep k_bucket_emit | sed "s/^/T $t: /"; done
