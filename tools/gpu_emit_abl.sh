#!/bin/bash
# ablation of k_bucket_emit_tiles (RK_INDEX_DEBUG) and of the tile sort: per-kernel averages of 8 builds each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=${1:-10000}
for dbg in 0 8 16 32 4 6 2 1; do
  out=$GRAFT_REPO_ROOT/gpurun_out/abl_$dbg
  rm -rf $out; mkdir -p $out
  ( cd /tmp && RK_INDEX_DEBUG=$dbg timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py index_only $n 8 > $out/log.txt 2>&1 ) || { tail -5 $out/log.txt; }
  python3 - <<PY
import csv,glob
for f in glob.glob("$out/**/run_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bucket_emit" in r["Name"]: print("debug %3s  k_bucket_emit_tiles avg %8.1f us (calls %s)" % ("$dbg", float(r["AverageNs"])/1e3, r["Calls"]))
PY
done
for t in 256 1024; do
  RK_INDEX_EMIT_T=$t timeout -k 10 120 python3 tools/prof_driver.py index_only $n 6 2>&1 | tail -1 | sed "s/^/EMIT_T=$t /"
done
