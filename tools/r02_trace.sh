set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/traceS; mkdir -p gpurun_out/traceS
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/traceS -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py sketch 128 5000000 4 ) > gpurun_out/traceS/log.txt 2>&1
tail -2 gpurun_out/traceS/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/traceS/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [r for r in rows if 'rk_sketch_kernel' in r['Kernel_Name']][-1]
t0 = int(last['Start_Timestamp'])
for r in rows:
    s = int(r['Start_Timestamp'])
    if s >= t0 - 200000 and s <= t0 + 600000:
        print('%-50s start %8.1f us dur %7.1f us' % (r['Kernel_Name'][:50], (s - t0) / 1e3, (int(r['End_Timestamp']) - s) / 1e3))
m = glob.glob('gpurun_out/traceS/**/*memory_copy_trace.csv', recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        s = int(r['Start_Timestamp'])
        if s >= t0 - 200000 and s <= t0 + 600000:
            print('copy %-30s start %8.1f us dur %7.1f us bytes %s' % (r.get('Direction', ''), (s - t0) / 1e3, (int(r['End_Timestamp']) - s) / 1e3, r.get('Bytes', r.get('Size', ''))))
PY
