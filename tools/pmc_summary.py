#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean value per (kernel, counter) over launches.
    python3 tools/pmc_summary.py gpurun_out/pmcA_*"""
import csv, glob, re, sys, collections
acc = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            m = re.search(r"\b(rk_\w+|k_\w+)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"].split("(")[0].split("::")[-1]
            per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            acc[(k, c)].append(v)
for (k, c), v in sorted(acc.items()):
    print("%s,%s,%d,%.3f" % (k, c, len(v), sum(v) / len(v)))
