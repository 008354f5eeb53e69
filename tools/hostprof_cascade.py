#!/usr/bin/env python
"""cProfile of the Python path of a whole prefill cascade (merge, merge, importance, prune) at a small shape, where the
host, not the GPU, sets the pace (development tool).   python tools/hostprof_cascade.py [c3|7b|c2thr]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from tools.trace_config import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
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
ff = ffa.FrameFusion(0.3, c["thr"], 0.1, compact_outputs=False)
k_of = {}


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
            w = ffa.last_query_importance(q, k_of[n_in], num=c["num"], is_causal=True, framefusion=ff, defer=True)
        h, pe, _ = ff(h, pe, None, w)
        n += 1
    return n


for _ in range(100):
    calls = prefill()
torch.cuda.synchronize()
n = 1000
t0 = time.perf_counter()
for _ in range(n):
    prefill()
torch.cuda.synchronize()
print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per cascade of {calls} calls (back to back, L = {L})")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    prefill()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
