set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace50; mkdir -p gpurun_out/trace50
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace50 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py dist 50000 5 ) > gpurun_out/trace50/log.txt 2>&1
tail -2 gpurun_out/trace50/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace50/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rk_dist_kernel' in r['Kernel_Name']]
for r in rows[-8:]:
    print(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else '', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us', 'lds', r.get('LDS_Block_Size'))
PY
