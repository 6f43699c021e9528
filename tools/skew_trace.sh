#!/bin/bash
# developer probe: the kernels of rk_index_build over a collection whose hashes crowd one end of the hash space (rocprofv3 kernel stats of
# tools/skew_probe.py; second argument as there: 0 / 1 = the canonical k-mer's 7 : 5 : 3 : 1 on one / two levels, else a power x 10)
#   gpurun -- "bash tools/skew_trace.sh [n_genomes] [skew]"
cd $GRAFT_REPO_ROOT
T=200 TOP=14 bash tools/gpu_trace_any.sh tr_skew skew_probe.py ${1:-10000} ${2:-1} | tail -16
