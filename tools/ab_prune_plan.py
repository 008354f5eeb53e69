#!/usr/bin/env python
"""A/B of the prune's plan enqueued by the attention hook's importance kernel (k_lq_finish_plan) against the separate plan
launch: whole prefill cascades back to back, the two modes alternated block by block (development tool).
    python tools/ab_prune_plan.py [c3|c5|c2thr] [--blocks 4 --reps 40]"""
import argparse, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables
from tools.trace_config import CONFIGS

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="c3")
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
c = CONFIGS[a.config]
dev = torch.device("cuda", 0)
F, P, d = c["F"], c["P"], c["d"]
h0, pt = video_tokens(F, P, d, p_change=c["p_change"], sigma=0.3, sigma_hi=c["sigma_hi"], seed=c["seed"], pre=c["pre"], post=c["post"],
                      dtype=torch.bfloat16, device=str(dev))
L = h0.shape[1]
pe0 = rotary_tables(L, 128, torch.bfloat16, device=str(dev), mrope=c["mrope"])
gen = torch.Generator(device=dev).manual_seed(7)
q = torch.randn(1, c["heads"], c["num"], 128, generator=gen, device=dev).to(torch.bfloat16)
k_full = torch.randn(1, c["kv_heads"], L, 128, generator=gen, device=dev).to(torch.bfloat16)
ff = ffa.FrameFusion(0.3, c["thr"], 0.1)
k_of = {}
lib = _lib.load()


def prefill():
    ff.prepare(pt, P, c["pre"], c["pre"] + F * P - 1, F * P, L)
    h, pe = h0, [t for t in pe0]
    n = 0
    while not (ff.finish_merging and ff.finish_pruning) and n < 30:
        n_in = h.shape[1]
        w = None
        if ff.finish_merging and not ff.finish_pruning:
            if n_in not in k_of:
                k_of[n_in] = k_full[:, :, :n_in].contiguous()
            w = ffa.last_query_importance(q, k_of[n_in], num=c["num"], is_causal=True, framefusion=ff)
        h, pe, _ = ff(h, pe, None, w)
        n += 1
    return h


for _ in range(30):
    prefill()
res, outs = {0: [], 1: []}, {}
for b in range(a.blocks):
    for mode in (1, 0):
        lib.ff_set_fused_prune_plan(mode)
        for _ in range(5):
            out = prefill()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = prefill()
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / a.reps * 1e6)
        outs[mode] = out
lib.ff_set_fused_prune_plan(1)
assert outs[0].shape == outs[1].shape and torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
for mode in (0, 1):
    v = res[mode]
    print(f"{a.config} {'plan in the hook launch' if mode else 'separate plan launch  '}: {L}->{outs[mode].shape[1]}: median {statistics.median(v):.1f} us per cascade  "
          f"min {min(v):.1f}  max {max(v):.1f}  ({len(v)} blocks of {a.reps})")
