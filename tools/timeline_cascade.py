#!/usr/bin/env python
"""Per-kernel timeline of a prefill cascade from a rocprofv3 --kernel-trace CSV of tools/trace_config.py: the kernels
behind the last idle gap (> 10 ms) are `n` identical cascades issued back to back; prints, per position in the cascade,
the kernel, its mean duration and the mean gap in front of it, then the sums (kernel time, gaps, span per cascade).

    python tools/timeline_cascade.py <kernel_trace.csv> <cascades in the last block>
"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
last_gap = 0
for i in range(1, len(ev)):
    if ev[i][1] - ev[i - 1][2] > 10_000_000:
        last_gap = i
block = ev[last_gap:]
def short(name):
    name = name.replace("void ", "")
    head = name.split("(")[0]
    return head.replace("ff::", "") if "ff::" in head else "[torch] " + head[:60]
if len(block) % n:
    print(f"# {len(block)} kernels in the last block do not divide into {n} cascades: dropping the first {len(block) % n}")
    block = block[len(block) % n:]
K = len(block) // n
casc = [block[i * K:(i + 1) * K] for i in range(n)]
names = [short(k[0]) for k in casc[0]]
same = [c for c in casc if [short(k[0]) for k in c] == names]
print(f"{len(same)} of {n} back-to-back cascades with the same {K} kernels" + ("; the first one (launched from an idle GPU) is left out" if len(same) > 2 else ""))
if len(same) > 2:
    same = same[1:]
tot_k = tot_g = 0.0
for j, name in enumerate(names):
    dur = sum(c[j][2] - c[j][1] for c in same) / len(same) / 1e3
    gap = sum((c[j][1] - c[j - 1][2]) for c in same) / len(same) / 1e3 if j else 0.0
    tot_k += dur
    tot_g += gap
    print(f"  {j:2d} {name:62s} dur {dur:7.1f} us   gap before {gap:6.1f} us")
span = sum(c[-1][2] - c[0][1] for c in same) / len(same) / 1e3
between = [same[i + 1][0][1] - same[i][-1][2] for i in range(len(same) - 1) if casc.index(same[i + 1]) == casc.index(same[i]) + 1]
print(f"per cascade: kernels {tot_k:.1f} us + gaps inside {tot_g:.1f} us = span {span:.1f} us; idle between consecutive cascades "
      f"{(sum(between) / len(between) / 1e3) if between else 0.0:.1f} us; kernel_us {tot_k:.1f}")
if any("k_merge_resident" in nm for nm in names):
    print("# note: k_merge_resident waits INSIDE the kernel for the host's mail (outputs allocated after the launch; on the threshold "
          "branch after the result block): its duration here contains the host's reaction time, which the tracer lengthens. Device "
          "time without the wait: tools/flow_stamps.py --wg (first workgroup start -> last workgroup end).")
