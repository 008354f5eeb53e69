#!/usr/bin/env python
"""A/B of the three ways a prune call gets its importance (development tool; one process, forms interleaved, several rounds):
  hook      last_query_importance(q, k, framefusion=ff) in the attention hook, then forward(..., importance)     (rounds 3-4)
  early     forward(..., LastQuery handle): importance launched at the top of _prune, plan + gather at its end  (shipped)
  one       forward(..., LastQuery handle) with ff.prune_in_one_crossing: ff_ctx_prune_from_qk at the end of _prune
Whole prefill cascades of a trace_config configuration back to back.   python tools/ab_prune_forms.py [c3|c5|c2thr] [rounds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from tools.trace_config import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = CONFIGS[name]
dev = torch.device("cuda", 0)
F, P, d = c["F"], c["P"], c["d"]
h0, pt = video_tokens(F, P, d, p_change=c["p_change"], sigma=0.3, sigma_hi=c["sigma_hi"], seed=c["seed"], pre=c["pre"], post=c["post"],
                      dtype=torch.bfloat16, device=str(dev))
L = h0.shape[1]
pe0 = rotary_tables(L, 128, torch.bfloat16, device=str(dev), mrope=c["mrope"])
gen = torch.Generator(device=dev).manual_seed(7)
q = torch.randn(1, c["heads"], c["num"], 128, generator=gen, device=dev).to(torch.bfloat16)
k_full = torch.randn(1, c["kv_heads"], L, 128, generator=gen, device=dev).to(torch.bfloat16)
k_of = {}


def make(form):
    ff = ffa.FrameFusion(0.3, c["thr"], 0.1, compact_outputs=False)
    if form == "one":
        ff.prune_in_one_crossing = True

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
                w = ffa.last_query_importance(q, k_of[n_in], num=c["num"], is_causal=True, framefusion=ff, defer=(form != "hook"))
            h, pe, _ = ff(h, pe, None, w)
            n += 1
        return h
    return prefill


forms = {f: make(f) for f in ("hook", "early", "one")}
outs = {f: fn() for f, fn in forms.items()}
assert all(torch.equal(outs["hook"], o) for o in outs.values()), "the three forms disagree"
for fn in forms.values():
    for _ in range(50):
        fn()
torch.cuda.synchronize()
n = 400
res = {f: [] for f in forms}
for r in range(rounds):
    for f, fn in forms.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[f].append((time.perf_counter() - t0) / n * 1e6)
print(f"# {name}: L = {L}, whole cascades back to back, us per cascade, {rounds} interleaved rounds of {n}")
for f, v in res.items():
    print(f"{f:6s} min {min(v):7.1f}  median {sorted(v)[len(v) // 2]:7.1f}  all {' '.join(f'{x:.1f}' for x in v)}")
