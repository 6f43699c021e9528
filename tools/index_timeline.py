"""Timeline of one rk_index_build out of a rocprofv3 kernel trace (tools/kernel_trace.sh <tag> index N reps): python3 tools/index_timeline.py gpurun_out/<tag> [build#]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
first = [i for i, r in enumerate(rows) if 'k_chunk_first' in r['Kernel_Name']]
a, b = first[which] - 3, first[which + 1] - 3
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1000:8.1f} {e/1000:8.1f} {(e-s)/1000:7.1f} q{r['Queue_Id']} {r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]}")
