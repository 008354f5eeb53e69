#!/usr/bin/env python
"""Trim rocprofv3 CSV output to this library's kernels: <prof_dir>/trace/*_kernel_stats.csv and the
pmc_fetch / pmc_write counter collections -> small summaries in <out_dir> (what profiles/ keeps)."""
import collections
import csv
import glob
import sys

prof, out = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
stats = glob.glob(f"{prof}/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "ff::" in r[0]]
    for r in keep[1:]:
        r[0] = r[0].split("(")[0].replace("void ", "")
    csv.writer(open(f"{out}/{tag}_kernel_stats.csv", "w")).writerows(keep)
for k in ("fetch", "write"):
    files = glob.glob(f"{prof}/pmc_{k}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "ff::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(f"{out}/{tag}_pmc_{k}_summary.csv", "w") as o:
        o.write("kernel,counter,launches,mean_value\n")
        for (kn, c), v in sorted(agg.items()):
            o.write('"%s",%s,%d,%.1f\n' % (kn, c, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(f"{out}/{tag}_*.csv")):
    print(f)
    print(open(f).read())
