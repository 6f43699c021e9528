#!/bin/bash
# developer probe: the kernel timeline (rocprofv3) of the last shard build and of one join build of tools/shard_probe.py -- default 500,000 genomes, eight shards
#   gpurun -- "bash tools/shard_timeline.sh [n_genomes] [shards]"
cd $GRAFT_REPO_ROOT
T=200 TOP=1 bash tools/gpu_trace_any.sh tr_abl_p shard_probe.py ${1:-500000} ${2:-8} | grep "shard 7\|join build"
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('gpurun_out/tr_abl_p/run_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_chunk_first' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
seen_join=0
for r in rows[i0:]:
    n=r['Kernel_Name']
    if 'trampoline' in n or 'at::' in n: continue
    if 'k_virtual_regions' in n:
        seen_join+=1
        t0=int(r['Start_Timestamp'])
    if seen_join>1: break
    m=re.search(r"(rk_\w+|k_\w+|radix_sort\w+|\w*scan\w*|\w+_kernel\w*)",n)
    print("%9.1f %9.1f us  q%s  %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),(m.group(1) if m else n[:60])[:60]))
PY
