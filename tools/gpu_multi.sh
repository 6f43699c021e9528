#!/bin/bash
# rehearsal of the N > 1 plumbing of bench.py on ONE card: 2 ranks over gloo, both on cuda:0 (RCCL needs one device per rank)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in sketches blob; do
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 \
   --backend gloo --same-device --replicate $mode --no-cpu-baseline --no-sketch > gpurun_out/bench_2ranks_$mode.json 2> gpurun_out/bench_2ranks_$mode.err || { tail -30 gpurun_out/bench_2ranks_$mode.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("gpurun_out/bench_2ranks_$mode.json").read().strip().split("\n")[-1])
print("$mode", d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["hits"], d["multi_gpu"])
print("   config3", d["config3"]["ms_per_step"], d["config3"]["hits"], d["config3"]["multi_gpu"])
print("   rq", d["dist_rq"]["ms_per_step"], d["dist_rq"]["hits"], d["dist_rq"].get("replicate_ms"))
PY
done
