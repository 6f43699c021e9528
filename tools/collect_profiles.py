#!/usr/bin/env python3
"""Turns the outputs of tools/measure_round.sh (merged back under gpurun_out/) into the committed
summaries under profiles/.      python3 tools/collect_profiles.py r02"""
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
    m = re.search(r"\b(rk_\w+(<[^>]*>)?|k_\w+(<[^>]*>)?)", name)
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


def variant_of(dirs, base):
    """full kernel name (with template arguments) of `base` as the counter passes saw it"""
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"\b(%s<[^>]*>)" % base, r["Kernel_Name"])
                if m:
                    return m.group(1)
    return base


def summary(dirs):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + dirs, capture_output=True,
                          text=True, check=True).stdout


def values(summ):
    vals = {}
    for line in summ.splitlines():
        k, c, n, v = line.split(",")
        vals[(k, c)] = float(v)
    return vals


def main(tag, traffic_only=False):
    """traffic_only: on the GPU box, between the counter passes and bench.py, so that the bench line of a round
    carries the counter traffic measured in the same call"""
    d = None
    if not traffic_only:
        d = json.loads(open(os.path.join(OUT, "bench_round.json")).read().strip().split("\n")[-1])
        json.dump(d, open(os.path.join(PROF, tag + "_bench.json"), "w"))
        f = glob.glob(os.path.join(OUT, "prof_final", "**", "*kernel_stats.csv"), recursive=True)[0]
        with open(os.path.join(PROF, tag + "_bench_kernel_stats.csv"), "w") as o:
            o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "
                    "(MI355X, tools/measure_round.sh)\nName,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in csv.DictReader(open(f)):
                o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"]))
        idx_txt = os.path.join(OUT, "index_kernels.txt")
        if os.path.exists(idx_txt):
            with open(os.path.join(PROF, tag + "_index_build_kernels.txt"), "w") as o:
                o.write("# rocprofv3 --kernel-trace --stats -- python3 tools/prof_driver.py index 10000 6 (MI355X, tools/measure_round.sh): "
                        "the kernels of rk_index_build at 10,000 genomes, and its wall time per call\n")
                o.write(open(idx_txt).read())
    # FETCH_SIZE calibration: a streaming read of 1 GiB at 4 / 8 / 16 bytes per lane (k_calib_read)
    calib = {}
    for w in (4, 8, 16):
        v = values(summary(sorted(glob.glob(os.path.join(OUT, "pmcC%d_*" % w)))))
        f = v.get(("k_calib_read", "FETCH_SIZE"))
        if f:
            calib[w] = (1024 << 20) / (f * 1024)
    if calib:
        with open(os.path.join(PROF, tag + "_fetch_calibration.txt"), "w") as o:
            o.write("# FETCH_SIZE against a known byte count: k_calib_read streams 1 GiB at W bytes per lane (tools/prof_driver.py calib); "
                    "factor = true bytes / (FETCH_SIZE x 1 KiB)\n")
            for w, fct in sorted(calib.items()):
                o.write("%2d B/lane: factor %.3f\n" % (w, fct))
    groups = {"dist": sorted(glob.glob(os.path.join(OUT, "pmcD_*"))),
              "dist_50k": sorted(glob.glob(os.path.join(OUT, "pmcD50_*"))),
              "tile_clade10": sorted(glob.glob(os.path.join(OUT, "pmcT10_*"))),
              "tile_50k": sorted(glob.glob(os.path.join(OUT, "pmcT50_*"))),
              "tile_clade100": sorted(glob.glob(os.path.join(OUT, "pmcT100_*"))),
              "tile_clade1000": sorted(glob.glob(os.path.join(OUT, "pmcT1000_*"))),
              "sketch_1000": sorted(glob.glob(os.path.join(OUT, "pmcSk1000_*"))),
              "index": sorted(glob.glob(os.path.join(OUT, "pmcI_*"))),
              "rq": sorted(glob.glob(os.path.join(OUT, "pmcQ_*"))),
              "sketch": sorted(glob.glob(os.path.join(OUT, "pmcSk_*")) + glob.glob(os.path.join(OUT, "pmcS_*"))),
              "sketch_img1": sorted(glob.glob(os.path.join(OUT, "pmcS1_*")))}
    with open(os.path.join(PROF, tag + "_pmc_summary.csv"), "w") as o:
        o.write("# rocprofv3 --pmc <group> --kernel-include-regex <kernel> --kernel-trace -- python3 tools/prof_driver.py "
                "{dist 10000 4 | dist_rq_dev 100000 1000 3 | sketch 128 5000000 | index 10000 3}; one group per pass (tools/pmc_pass.sh, "
                "groups in tools/pmc_groups_*.txt); mean over launches (tools/pmc_summary.py).  Section sketch = the two-stage scan "
                "kernel (default), sketch_img1 = rk_sketch_kernel with the 64 KiB LDS image (RK_SKETCH_IMG=1, round 2's default)\n"
                "Section,Kernel,Counter,Launches,MeanValue\n")
        for sec, dirs in groups.items():
            for line in summary(dirs).splitlines():
                o.write(sec + "," + line + "\n")
    # counter-measured HBM bytes per launch, keyed by the exact kernel variant (bench.py refuses a mismatch)
    def traffic(dirs, base, fname, workload, width, coalesced_bytes=0):
        """coalesced_bytes: a known coalesced stream inside an otherwise scattered read pattern (the query hashes of the rq kernel):
        FETCH_SIZE saw half of it, so the estimate is FETCH + coalesced/2, between the bounds FETCH x 1 and FETCH x 2"""
        if not dirs:
            return None
        v = values(summary(dirs))
        fetch, write = v.get((base, "FETCH_SIZE")), v.get((base, "WRITE_SIZE"))
        if fetch is None or write is None:
            print("no FETCH_SIZE/WRITE_SIZE for", base)
            return None
        miss = v.get((base, "TCC_MISS_sum"), 0.0)
        # gfx950: FETCH_SIZE reports half the bytes of a wide coalesced 16 B/lane stream (MI355X_MICROARCH.md, HBM); other
        # widths are calibrated in this very run (k_calib_read above); width 0 = scattered 4-8 B gathers, which leave L2 as
        # 64-B requests and need no correction (cross-check: TCC_MISS x 64 B)
        factor = round(calib.get(width, 2.0 if width == 16 else 1.0), 2) if width else 1.0
        hbm = int(round((fetch * factor + write) * 1024))
        extra = {}
        if coalesced_bytes:
            hbm = int(round((fetch + write) * 1024 + coalesced_bytes / 2))
            extra = {"hbm_bytes_lower": int(round((fetch + write) * 1024)), "hbm_bytes_upper": int(round((2 * fetch + write) * 1024)),
                     "coalesced_stream_bytes": coalesced_bytes,
                     "estimate": "FETCH_SIZE reports half of a coalesced stream: FETCH + WRITE + coalesced stream / 2"}
        json.dump({**extra, "kernel": variant_of(dirs, base), "workload": workload, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                   "SQ_ACTIVE_INST_VALU": v.get((base, "SQ_ACTIVE_INST_VALU")), "SQ_INSTS_VALU": v.get((base, "SQ_INSTS_VALU")),
                   "GRBM_GUI_ACTIVE": v.get((base, "GRBM_GUI_ACTIVE")),
                   "fetch_correction": factor, "TCC_MISS_sum": miss, "TCC_MISS_x64B": miss * 64,
                   "hbm_bytes_per_launch": hbm,
                   "passes": "FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, see profiles/%s_pmc_summary.csv" % tag},
                  open(os.path.join(PROF, fname), "w"), indent=1)
        return hbm
    # the self join streams its 8-byte slice records (compact slices carry their posting list): a coalesced 8 B/lane
    # stream, corrected by the factor calibrated above (check: ~5.4 M uncovered slices x 8 B = 43 MB)
    t_d = traffic(groups["dist"], "rk_near_kernel", "pmc_traffic.json", "alldist 10,000 sketches (tools/prof_driver.py dist 10000 4), MI355X", 8)
    t_q = traffic(groups["rq"], "rk_distq_kernel", "pmc_traffic_rq.json", "dist 100,000 refs x 1,000 queries (tools/prof_driver.py dist_rq_dev), MI355X", 0,
                  coalesced_bytes=4 * 45776 * 1000)   # the query hashes, read once, 4 B per lane
    traffic(groups["dist_50k"], "rk_near_kernel", "pmc_traffic_50k.json", "alldist 50,000 sketches (tools/prof_driver.py dist 50000 3), MI355X", 8)
    traffic(groups["tile_clade10"], "rk_tile_kernel", "pmc_traffic_tile_clade10.json", "alldist 10,000 sketches, clades of 10, tile records from the index build: every join (tools/prof_driver.py dist 10000 4), MI355X", 8)
    traffic(groups["tile_50k"], "rk_tile_kernel", "pmc_traffic_tile_50k.json", "alldist 50,000 sketches, tile records from the index build: every join (tools/prof_driver.py dist 50000 3), MI355X", 8)
    traffic(groups["tile_clade100"], "rk_tile_kernel", "pmc_traffic_tile_clade100.json", "alldist 10,000 sketches in species of 100 (tools/prof_driver.py dist 10000 4 1 0 0 100), MI355X", 8)
    traffic(groups["tile_clade1000"], "rk_tile_kernel", "pmc_traffic_tile_clade1000.json", "alldist 10,000 sketches in species of 1,000 (tools/prof_driver.py dist 10000 4 1 0 0 1000), MI355X", 8)
    traffic(groups["sketch_1000"], "rk_scan2_kernel", "pmc_traffic_sketch1000.json", "sketch 1,000 x 5 Mb (tools/prof_driver.py sketch 1000 5000000 2), MI355X", 16)
    t_s = traffic(groups["sketch"], "rk_scan2_kernel", "pmc_traffic_sketch.json", "sketch 128 x 5 Mb (tools/prof_driver.py sketch 128 5000000), MI355X", 16)
    print("traffic dist %s rq %s sketch %s B/launch" % (t_d, t_q, t_s))
    if d:
        print("value %.4g %s, %.4f ms/step, contract frac %.3f, hbm frac %s" % (
            d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("hbm_frac")))


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    main(args[0] if args else "r04", "--traffic-only" in sys.argv)
