#!/bin/bash
# developer ablation of k_trec_rowsort at 500,000 genomes, eight shards (RK_ROWSORT_DEBUG bits, rk_index_tiles.inc)
cd $GRAFT_REPO_ROOT
for x in 0; do
  RK_ROWSORT_DEBUG=$x T=200 TOP=1 bash tools/gpu_trace_any.sh tr_abl_r shard_probe.py 500000 8 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/tr_abl_r/run_kernel_stats.csv')):
    if 'k_trec_rowsort' in r['Name']: print($x, 'k_trec_rowsort', float(r['AverageNs'])/1e3)
PY
done
