#!/usr/bin/env python
"""Per-kernel HBM fractions of the committed cascade timelines (profiles/<tag>_timeline_<cfg>.txt): every streaming kernel's
COMPULSORY bytes (bench.call_bytes' rules, split per kernel) / its duration / 8 TB/s.  No GPU needed.
    python tools/kernel_fractions.py r05 > profiles/r05_kernel_fractions.txt"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.trace_config import CONFIGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK, ELT, DH = 8.0e12, 2, 128
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
print(f"# tools/kernel_fractions.py {tag}: compulsory bytes per kernel / duration (rocprofv3 timelines of the same tag) / 8 TB/s.")
print("# K1 similarity: Nv rows read.  K4 of a folding merge: (L_in + L_out) rows + cos/sin in + out + 8-byte ints (SURVEY 8d).  K4 of an")
print("# identity merge: nothing (early exit).  The prune's gather: 2 x L_out x (row + cos/sin rows) - dropped rows are never read.  K5: H_kv x S x dh.")
print("# plan kernels move < 1 MB: latency, no fraction.\n")
worst = 0.0
for cfg in ("c2", "7b", "c3", "c5", "c5topk", "c2thr"):
    path = os.path.join(ROOT, "profiles", f"{tag}_timeline_{cfg}.txt" if cfg != "c2" else f"{tag}_timeline.txt")
    if not os.path.exists(path):
        continue
    text = open(path).read()
    c = CONFIGS[cfg]
    row = c["d"] * ELT
    pe_row = 2 * DH * ELT * (3 if c["mrope"] else 1)
    vis = c["F"] * c["P"]
    if cfg == "c2":
        calls = [("merge", vis, 11060)]
        kern = [(m.group(1), float(m.group(2))) for m in re.finditer(r"^\s+\d+ (k_\w+)\s+dur\s+([0-9.]+) us", text, re.M)]
    else:
        calls = [(k, int(a), int(b)) for k, a, b in re.findall(r"(merge|prune):(\d+)->(\d+)", text.split("\n")[1])]
        kern = [(m.group(1), float(m.group(2))) for m in re.finditer(r"^\s+\d+ (k_\w+)<[^>]*>\s+dur\s+([0-9.]+) us", text, re.M)]
    print(f"## {cfg}: {c['F']} x {c['P']} x {c['d']} bf16" + (", M-RoPE" if c["mrope"] else "") + f"   ({os.path.basename(path)})")
    q = 0
    nv = vis
    for kind, a, b in calls:
        if kind == "merge" and kern[q][0].startswith("k_merge_resident"):
            # the one-launch kernel: every row read ONCE, l_out rows written (an identity call: the visual rows read)
            per = [("k_merge_resident", nv * row if a == b else a * row + b * row + (a + b) * pe_row + 8 * (a + b))]
            nv -= a - b
        elif kind == "merge":
            fold_bytes = 0 if a == b else (a + b) * row + (a + b) * pe_row + 8 * (a + b)
            per = [("k_pair_similarity", nv * row), ("k_plan", 0), ("k_merge_compact", fold_bytes)]
            if a != b and q + 3 < len(kern) and kern[q + 2][0].startswith("k_merge_compact") and kern[q + 3][0].startswith("k_merge_compact"):
                # exactly sized outputs guessed for the other branch: the guarded merge kernel wrote nothing, the real one follows
                per = [("k_pair_similarity", nv * row), ("k_plan", 0), ("k_merge_compact", 0), ("k_merge_compact", fold_bytes)]
            nv -= a - b
        else:
            per = [("k_lq", 0), ("k_lq", c["kv_heads"] * a * DH * ELT), ("k_plan", 0), ("k_prune_gather", 2 * b * (row + pe_row))]
        for want, nbytes in per:
            name, us = kern[q]
            assert name.startswith(want) or (want == "k_prune_gather" and name.startswith("k_merge_compact")), (cfg, q, name, want)
            q += 1
            if want == "k_lq" and nbytes == 0:            # scores + finish: one line for the pair
                name2, us2 = kern[q]
                us, name = us + us2, name + " + " + name2
                nbytes = per[1][1]
                q += 1
                per.pop(1)
            if nbytes:
                frac = nbytes / (us * 1e-6) / PEAK
                worst = max(worst, frac)
                print(f"  {kind:5s} {a:6d} -> {b:6d}  {name:34s} {us:7.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:5.2f} TB/s  {frac:5.2f}")
            else:
                print(f"  {kind:5s} {a:6d} -> {b:6d}  {name:34s} {us:7.1f} us  (latency)")
    print()
print("# k_merge_resident: its traced duration contains the wait for the host's mail (see the note in the timelines); device time without it:")
print("# <tag>_resident_stamps.txt.")
print(f"# largest fraction: {worst:.2f} (nothing above the 8 TB/s peak, nothing above the chip's ~6.3 TB/s copy rate = 0.79)")
