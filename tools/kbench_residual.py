#!/usr/bin/env python
"""Decoder-layer slice around call B: `hidden = residual + attn_out` followed by
FrameFusion.forward (what the reference does, framefusion/models/qwen2/modeling_qwen2.py:64-67) against
forward_residual (the add formed inside the two streaming passes).  64 x 576 x 4096 bf16; merge call in the
top-k regime and the prune call; hipEvent time over the whole slice, inputs alternated between two buffers."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables

dev = "cuda:0"
F, P, d = 64, 576, 4096
h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1234, device=dev)
L = h.shape[1]
g = torch.Generator(device=dev).manual_seed(1)
res = [(0.5 * torch.randn(1, L, d, generator=g, device=dev)).bfloat16() for _ in range(2)]
att = [(h - r).contiguous() for r in res]
cos, sin = rotary_tables(L, 128, device=dev)
ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)


def timeit(fn, n=40):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def merge_unfused(i):
    ff.prepare(pt, P, 0, L, L, L)
    return ff(res[i & 1] + att[i & 1], [cos, sin], None)


def merge_fused(i):
    ff.prepare(pt, P, 0, L, L, L)
    return ff.forward_residual(res[i & 1], att[i & 1], [cos, sin], None)


w = torch.rand(1, 1, 1, L, generator=g, device=dev).bfloat16()


def prune_unfused(i):
    ff.prepare(pt, P, 0, L, L, L, finish_merging=True, sparsity_list=[0.0])
    return ff(res[i & 1] + att[i & 1], [cos, sin], None, w)


def prune_fused(i):
    ff.prepare(pt, P, 0, L, L, L, finish_merging=True, sparsity_list=[0.0])
    return ff.forward_residual(res[i & 1], att[i & 1], [cos, sin], None, w)


a, b = timeit(merge_unfused), timeit(merge_fused)
o = merge_fused(0)[0]
print(f"merge call  ({L} -> {o.shape[1]}): eager add + forward {a:7.1f} us   forward_residual {b:7.1f} us   (-{a - b:.1f} us)")
a, b = timeit(prune_unfused), timeit(prune_fused)
o = prune_fused(0)[0]
print(f"prune call  ({L} -> {o.shape[1]}): eager add + forward {a:7.1f} us   forward_residual {b:7.1f} us   (-{a - b:.1f} us)")
