#!/usr/bin/env python
"""Durations of this library's kernels in a rocprofv3 kernel trace, grouped by the kernel that ran
just before (warm repeat of the same small kernels vs first launch after a streaming pass)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ff::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("void ", "").split("<")[0].split("(")[0].replace("ff::", "")


d = collections.defaultdict(list)
prev = "-"
for r in rows:
    k = short(r["Kernel_Name"])
    d[(k, prev)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    prev = k
for (k, p), v in sorted(d.items()):
    if len(v) >= 3:
        print(f"{k:18s} after {p:18s} n={len(v):3d} mean {sum(v) / len(v):6.1f} min {min(v):6.1f}")
