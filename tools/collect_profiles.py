#!/usr/bin/env python3
"""Turns the outputs of tools/measure_round.sh (merged back under gpurun_out/) into the committed
summaries under profiles/.      python3 tools/collect_profiles.py r01"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def short_name(name):
    m = re.search(r"\b(rk_\w+(<[^>]*>)?|k_\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"rocprim::\w+::detail::trampoline_kernel<[^,]*?detail::(\w+)", name) or \
        re.search(r"rocprim::\w+::detail::(\w+)", name)
    if m:
        return "rocprim::" + m.group(1)
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
    if m:
        return "torch::" + m.group(1)
    return re.sub(r"\(.*", "", name).replace("void ", "")


def main(tag):
    bench = os.path.join(OUT, "bench_r1.json")
    d = json.load(open(bench))
    json.dump(d, open(os.path.join(PROF, tag + "_bench.json"), "w"))
    f = glob.glob(os.path.join(OUT, "prof_final", "**", "*kernel_stats.csv"), recursive=True)[0]
    with open(os.path.join(PROF, tag + "_bench_kernel_stats.csv"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "
                "(MI355X, tools/measure_round.sh)\nName,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in csv.DictReader(open(f)):
            o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                  r["Percentage"], r["MinNs"], r["MaxNs"]))
    dirs = sorted(glob.glob(os.path.join(OUT, "pmcD_*")) + glob.glob(os.path.join(OUT, "pmcSk_*")) +
                  glob.glob(os.path.join(OUT, "pmcS_*")))
    summ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + dirs, capture_output=True,
                          text=True, check=True).stdout
    with open(os.path.join(PROF, tag + "_pmc_summary.csv"), "w") as o:
        o.write("# rocprofv3 --pmc <group> --kernel-include-regex <kernel> --kernel-trace -- python3 tools/prof_driver.py "
                "{dist 10000 4 | sketch 128 5000000}; one group per pass (tools/pmc_pass.sh, groups in "
                "tools/pmc_groups_*.txt); mean over launches (tools/pmc_summary.py)\nKernel,Counter,Launches,MeanValue\n")
        o.write(summ)
    vals = {}
    for line in summ.splitlines():
        k, c, n, v = line.split(",")
        vals[(k, c)] = float(v)
    fetch, write = vals[("rk_dist_kernel", "FETCH_SIZE")], vals[("rk_dist_kernel", "WRITE_SIZE")]
    miss = vals.get(("rk_dist_kernel", "TCC_MISS_sum"), 0.0)
    sk = vals.get(("rk_sketch_kernel", "FETCH_SIZE"), 0.0)
    json.dump({
        "kernel": "rk_dist_kernel", "workload": "alldist 10,000 sketches (tools/prof_driver.py dist 10000 4), MI355X",
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "hbm_bytes_per_launch": int(round((fetch + write) * 1024)),
        "correction": "none applied to FETCH_SIZE: this kernel's reads are 8 B/lane gathers and slice loads that leave L2 as "
                      "64-B requests (TCC_MISS_sum %.2f M x 64 B = %.1f MB vs FETCH_SIZE x 1024 = %.1f MB). The gfx950 x2 "
                      "correction of MI355X_MICROARCH.md applies to wide 16 B/lane streams only; checked on rk_sketch_kernel in "
                      "the same session: FETCH_SIZE %.0f KB x 2 x 1024 = %.1f MB for 640.0 MB of sequence bytes read once."
                      % (miss / 1e6, miss * 64 / 1e6, fetch * 1024 / 1e6, sk, sk * 2 * 1024 / 1e6),
        "passes": "FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots), see "
                  "profiles/%s_pmc_summary.csv" % tag}, open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
    print("value %.4g %s, %.4f ms/step, roofline frac %.3f, traffic %d B" % (
        d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], int(round((fetch + write) * 1024))))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
