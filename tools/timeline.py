#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV into the per-call timeline of the merge step:
kernel durations and the gaps between consecutive kernels of one FrameFusion.forward call."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "ff::" in r["Kernel_Name"]]     # (memset / fill kernels of the runtime are not part of a call)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "")
    return n.split("<")[0].split("(")[0].replace("ff::", "").replace("k_plan_fast", "k_plan").replace("k_pair_similarity_tile", "k_pair_similarity")
# a call = a maximal run of kernels ending with k_merge_compact; the product's merge call is either
# K0 (two launches) + similarity + hist + flags + scan + merge, or - with a layout hint - starts at
# the similarity kernel
calls = []
cur = []
for r in rows:
    k = short(r["Kernel_Name"])
    cur.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if k == "k_merge_compact":
        calls.append(cur)
        cur = []
tail = ["k_pair_similarity", "k_plan", "k_merge_compact"]
old_tail = ["k_pair_similarity", "k_hist_level", "k_flags", "k_scan", "k_merge_compact"]      # round-1 builds
shapes = [["k_order_stats", "k_build_order"] + tail, tail, old_tail]
calls = [c for c in calls if [k for k, _, _ in c] in shapes]
if calls:
    common = max(shapes, key=lambda sh: sum([k for k, _, _ in c] == sh for c in calls))
    calls = [c for c in calls if [k for k, _, _ in c] == common][5:]
calls = [c for c in calls if (c[0][2] - c[0][1]) / 1e3 >= 0.75 * max((x[0][2] - x[0][1]) / 1e3 for x in calls)]   # the bench workload only
dur = defaultdict(list)
gap = defaultdict(list)
span = []
for c in calls:
    for i, (k, s, e) in enumerate(c):
        dur[(i, k)].append((e - s) / 1e3)
        if i:
            gap[(i, k)].append((s - c[i - 1][2]) / 1e3)
    span.append((c[-1][2] - c[0][1]) / 1e3)
between = [(calls[i + 1][0][1] - calls[i][-1][2]) / 1e3 for i in range(len(calls) - 1)]
between = [b for b in between if b < 1000]
if between:
    between.sort()
    print(f"idle between consecutive calls (last kernel end -> next first kernel start): median {between[len(between)//2]:.1f} us  "
          f"min {between[0]:.1f}  max {between[-1]:.1f}")
ss = sorted(span)
print(f"{len(calls)} calls; first-kernel-start to last-kernel-end: mean {sum(span)/len(span):.1f} us  min {min(span):.1f}  "
      f"median {ss[len(ss)//2]:.1f}  p90 {ss[int(len(ss)*0.9)]:.1f}")
for (i, k), v in sorted(dur.items()):
    g = gap.get((i, k), [0])
    print(f"  {i} {k:18s} dur {sum(v)/len(v):7.1f} us   gap before {sum(g)/len(g):6.1f} us")
