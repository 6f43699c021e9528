#!/bin/bash
# usage: tools/gpu_trace_any.sh <tag> <script.py> [args...]  -> rocprofv3 kernel stats of any tools/ script (top kernels by total time)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out; mkdir -p $out
( cd /tmp && timeout -k 10 ${T:-600} rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/$@ > $out/log.txt 2>&1 ) || { tail -20 $out/log.txt; exit 1; }
grep -v "^[EWI]2026\|amdgpu.ids" $out/log.txt | tail -8
python3 - <<PY
import csv,glob,re
for f in glob.glob("$out/**/run_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:${TOP:-22}]:
        n=r["Name"]; m=re.search(r"(rk_\w+<[^>]*>|rk_\w+|k_\w+(<[^>]*>)?|radix_sort\w+|scan_impl|__amd_\w+)",n)
        print("%-50s calls %4s avg %10.1f us total %10.1f us" % ((m.group(1) if m else n[:50])[:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
