#!/usr/bin/env python
"""Whole-prefill cascade on one GPU: every FrameFusion.forward call of one sample (call A, then call B
per layer until merging and pruning are finished), timed per call with the host clock after a
device synchronise (development tool)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from tests import harness


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--patches", type=int, default=576)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--p-change", type=float, default=0.5)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda:0"
    F, P, d = a.frames, a.patches, a.dim
    h0, pt = video_tokens(F, P, d, p_change=a.p_change, sigma=0.3, sigma_hi=1.6, seed=1234, pre=14, post=20, device=dev)
    L = h0.shape[1]
    ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)
    agg = {}
    for rep in range(a.reps + 2):
        ff.prepare(pt, P, 14, 14 + F * P, F * P, L)
        h, pe = h0, rotary_tables(L, 128, device=dev)
        calls = []
        layer = -1
        while not (ff.finish_merging and ff.finish_pruning) and layer < 27:
            w = None
            if ff.finish_merging and not ff.finish_pruning:
                w = harness.attention_stub(28, 1, h.shape[1], h.dtype, dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_in = h.shape[1]
            h, pe, _ = ff(h, pe, None, w)
            torch.cuda.synchronize()
            calls.append((ff.last_call["kind"], n_in, h.shape[1], (time.perf_counter() - t0) * 1e6))
            layer += 1
        if rep >= 2:
            for i, c in enumerate(calls):
                agg.setdefault(i, []).append(c)
    total = 0.0
    for i, cs in sorted(agg.items()):
        us = sum(c[3] for c in cs) / len(cs)
        total += us
        print(f"call {i}: {cs[0][0]:5s} {cs[0][1]:6d} -> {cs[0][2]:6d}   {us:8.1f} us")
    print(f"cascade total {total:.1f} us, {L} -> {cs[0][2]}: {(L - cs[0][2]) / total:.1f} M tokens reduced/s over the prefill")


if __name__ == "__main__":
    main()
