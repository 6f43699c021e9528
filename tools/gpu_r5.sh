#!/bin/bash
# Round-5 developer loop on the GPU box (gpurun -- 'bash tools/gpu_r5.sh <what>').
#   tiles   the tests of the tile records that come with the index build + build / first-join timings at 10k / 50k
#   trace   per-kernel timeline of the index build (rocprofv3 --kernel-trace --stats) at $2 genomes
#   all     the whole GPU suite
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tests() { timeout -k 10 ${T:-900} python3 -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/r5_tests.log 2>&1; rc=$?; tail -${LINES_SHOWN:-25} gpurun_out/r5_tests.log; [ $rc -eq 0 ] || exit 1; }
case "$1" in
tiles)
  tests "tile or kernel_follows or index_without or near_window or shard or passes or beyond"
  for n in 10000 50000; do
    RK_DIST_DEBUG=1 timeout -k 10 300 python3 tools/prof_driver.py index $n 6 > gpurun_out/r5_index_$n.log 2>&1 || { tail -30 gpurun_out/r5_index_$n.log; exit 1; }
    grep -v amdgpu.ids gpurun_out/r5_index_$n.log | tail -14
    timeout -k 10 300 python3 tools/prof_driver.py dist $n 200 > gpurun_out/r5_dist_$n.log 2>&1 || { tail -30 gpurun_out/r5_dist_$n.log; exit 1; }
    tail -1 gpurun_out/r5_dist_$n.log
  done ;;
trace)
  n=${2:-10000}
  bash tools/kernel_trace.sh r5_trace_$n index $n 8 || exit 1
  python3 tools/index_timeline.py gpurun_out/r5_trace_$n 4 ;;
all)
  T=1150 LINES_SHOWN=40 tests "" ;;
*) echo "usage: $0 tiles|trace|all"; exit 2 ;;
esac
