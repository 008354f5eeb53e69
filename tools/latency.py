#!/usr/bin/env python
"""Latency of ONE isolated FrameFusion.forward merge call (idle GPU before, synchronise after - what a prefill
pays per call, and what bench.py's `extra.configs` report), split at the two crossings of the C ABI:

    python tools/latency.py [F P d [pre post]]

  pre      python before ff_ctx_merge_begin (checks, budget, packing the call block)
  begin    ff_ctx_merge_begin (K1 launch; + K0 when unhinted)
  alloc    python between the crossings (output tensors, aux descriptors)
  finish   ff_ctx_merge_finish = enqueue plan + K4, then the poll (`wait` of it is the poll)
  post     python after the result (state machine, views)
  drain    torch.cuda.synchronize() after the call returned (K4 still running)
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa                                   # noqa: E402
from framefusion_amd import _lib                                # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables   # noqa: E402

DEV = "cuda:0"
a = [int(x) for x in sys.argv[1:]]
F, P, d = (a + [64, 210, 3584])[:3] if len(a) < 3 else a[:3]
pre, post = (a[3], a[4]) if len(a) >= 5 else (14, 20)
h, pt = video_tokens(F, P, d, p_change=0.2, seed=1234, pre=pre, post=post, dtype=torch.bfloat16, device=DEV)
L = h.shape[1]
cos, sin = rotary_tables(L, 128, torch.bfloat16, device=DEV)
ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)
lib = _lib.load()
marks = {}
raw_begin, raw_finish = lib.ff_ctx_merge_begin, lib.ff_ctx_merge_finish
now = time.perf_counter_ns


def begin(*args):
    marks["b0"] = now()
    rc = raw_begin(*args)
    marks["b1"] = now()
    return rc


def finish(*args):
    marks["f0"] = now()
    rc = raw_finish(*args)
    marks["f1"] = now()
    return rc


lib.ff_ctx_merge_begin, lib.ff_ctx_merge_finish = begin, finish
rows = []
for it in range(60):
    ff.prepare(pt, P, pre, pre + F * P - 1, F * P, L)
    torch.cuda.synchronize()
    time.sleep(0.0005)
    t0 = now()
    out, pe, _ = ff(h, [cos, sin], None)
    t1 = now()
    torch.cuda.synchronize()
    t2 = now()
    w = ff.last_call["wait_ns"]
    if it >= 10:
        rows.append(dict(pre=marks["b0"] - t0, begin=marks["b1"] - marks["b0"], alloc=marks["f0"] - marks["b1"],
                         finish=marks["f1"] - marks["f0"], wait=w, post=t1 - marks["f1"], drain=t2 - t1, total=t2 - t0))
print(f"[1, {pre}+{F}x{P}+{post}, {d}] bf16: {L} -> {out.shape[1]} tokens; isolated call, median of {len(rows)} (us)")
for k in rows[0]:
    v = sorted(r[k] / 1e3 for r in rows)
    print(f"  {k:7s} median {statistics.median(v):7.1f}   min {v[0]:7.1f}   p90 {v[int(0.9 * (len(v) - 1))]:7.1f}")
