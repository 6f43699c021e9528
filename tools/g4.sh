cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_reference_layout.py tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -x -q --durations=5 > gpurun_out/t4.log 2>&1 || { tail -50 gpurun_out/t4.log; exit 1; }
tail -9 gpurun_out/t4.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_try.json 2> gpurun_out/bench_try.err || { tail -30 gpurun_out/bench_try.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_try.json").read().strip().split("\n")[-1])
print("value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]))
print("roofline", {k: d["roofline"][k] for k in ("achieved","frac","traffic","contract_frac","issue_frac","kernel","kernel_ms","kernel_ms_min_median_max")})
print("build_plus_dist", d["build_plus_dist"]["ms"], d["build_plus_dist"]["index_build"]["frac"], d["setup"]["index_build_ms"])
print("orders", d["alldist_order"])
print("rehearsal", {n: {s: (round(v["slowest_ms"],4), round(v["predicted_efficiency"],3)) for s, v in r.items()} for n, r in d["scaling_rehearsal"].items()})
print("config3", d["config3"]["ms_per_step"], d["config3"]["index_build_ms"], d["config3"]["build_plus_dist_ms"], d["config3"]["roofline"]["frac"])
print("rq", d["dist_rq"]["ms_per_step"], d["dist_rq"]["roofline"]["frac"])
print("sketch", d["sketch"]["ms_per_pass"], d["sketch"]["roofline"]["frac"])
PY
