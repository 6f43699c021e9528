set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout -k 10 600 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 1; }
echo "bench N=1 took $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.load(open('gpurun_out/bench.json'))
def show(k,v,ind=0):
    if isinstance(v,dict):
        print(' '*ind+k+':')
        for a,b in v.items(): show(a,b,ind+2)
    else:
        s=str(v); print(' '*ind+k+': '+(s if len(s)<140 else s[:140]+'...'))
for k,v in d.items(): show(k,v)
"
t0=$(date +%s)
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 3 > gpurun_out/bench2.json 2> gpurun_out/bench2.err || { tail -30 gpurun_out/bench2.err; exit 1; }
echo "bench N=2 rehearsal (gloo, one GPU) took $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.loads(open('gpurun_out/bench2.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','scaling')}, d['config']['hits'], d['config3']['value'], d['config3']['hits'], d['dist_rq']['value'], d['dist_rq']['hits'])
"
