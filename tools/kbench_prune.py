#!/usr/bin/env python
"""Timing of the prune side (importance + prune call) on one GPU (development tool)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=11094)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=8)
    ap.add_argument("--dh", type=int, default=128)
    ap.add_argument("--num", type=int, default=1)
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(1, a.H, 16, a.dh, generator=g, device=dev).bfloat16()
    k = torch.randn(1, a.Hkv, a.S, a.dh, generator=g, device=dev).bfloat16()
    h = torch.randn(1, a.S, a.dim, generator=g, device=dev).bfloat16()
    cos, sin = rotary_tables(a.S, a.dh, device=dev)

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_w = timeit(lambda: ffa.scaled_dot_product_attention(q, k, None, num=a.num, is_causal=True, enable_gqa=True))
    t_i = timeit(lambda: ffa.last_query_importance(q, k, num=a.num, is_causal=True))
    w = ffa.scaled_dot_product_attention(q, k, None, num=a.num, is_causal=True, enable_gqa=True)
    imp = ffa.last_query_importance(q, k, num=a.num, is_causal=True)
    ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)
    pt = torch.full((1, a.S), 0, dtype=torch.long, device=dev)
    start, n_img = 14, a.S - 34

    def prune(weights):
        ff.prepare(pt, 576, start, start + n_img, n_img, a.S, finish_merging=True, sparsity_list=[0.5])
        return ff(h, [cos, sin], None, weights)

    def fused():
        ff.prepare(pt, 576, start, start + n_img, n_img, a.S, finish_merging=True, sparsity_list=[0.5])
        return ff(h, [cos, sin], None, ffa.last_query_importance(q, k, num=a.num, is_causal=True, framefusion=ff))

    t_p = timeit(lambda: prune(w))
    t_pi = timeit(lambda: prune(imp))
    t_f = timeit(fused)
    out = prune(w)[0]
    print(f"S={a.S} H={a.H}/{a.Hkv} dh={a.dh} num={a.num} d={a.dim}: weights {t_w:.1f} us, importance {t_i:.1f} us, "
          f"prune call (weights) {t_p:.1f} us, prune call (importance) {t_pi:.1f} us, importance(+tables) + prune {t_f:.1f} us, "
          f"{a.S} -> {out.shape[1]}")
    kb = a.Hkv * a.S * a.dh * 2
    print(f"importance kernels: K traffic {kb/1e6:.0f} MB -> {kb/t_i/1e3:.0f} GB/s")
    bytes_prune = (a.S + out.shape[1]) * a.dim * 2
    print(f"prune gather algorithmic {bytes_prune/1e6:.0f} MB -> {bytes_prune/t_pi/1e3:.0f} GB/s")


if __name__ == "__main__":
    main()
