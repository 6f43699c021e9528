cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_index
( cd /tmp && RK_INDEX_EMIT_T=1024 timeout -k 5 60 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_index -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py index 10000 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_index.log 2>&1 ) || { echo "failed"; tail -3 gpurun_out/prof_index.log; exit 1; }
grep "index build\|host-incl" gpurun_out/prof_index.log
f=$(find gpurun_out/prof_index -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    n=r["Name"].replace("(anonymous namespace)::","")
    print("%-60s calls %4s avg %10.1f min %9s" % (n[:60], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index or order or alldist or crowded" 2>&1 | tail -2
