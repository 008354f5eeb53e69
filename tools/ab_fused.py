#!/usr/bin/env python
"""A/B of the fused plan + merge launch (csrc/ff_fused.hip) against the two-launch form, same process, same inputs:
the bench step (prepare + one merge call, back to back) on the event clock, alternating the two modes block by block
so that clock / thermal drift hits both alike (development tool).

    python tools/ab_fused.py [--frames 64 --patches 576 --dim 4096 --p-change 0.2] [--blocks 6 --steps 40]
"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--patches", type=int, default=576)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--p-change", type=float, default=0.2)
    ap.add_argument("--pre", type=int, default=0)
    ap.add_argument("--post", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    F, P, d = a.frames, a.patches, a.dim
    h, pt = video_tokens(F, P, d, p_change=a.p_change, sigma=0.3, seed=1234, pre=a.pre, post=a.post, dtype=torch.bfloat16, device=str(dev))
    L = h.shape[1]
    h2 = h.clone()
    cos, sin = rotary_tables(L, 128, torch.bfloat16, device=str(dev))
    ff = ffa.FrameFusion(0.3, 0.6, 0.1)
    lib = _lib.load()
    flip = [0]

    def step():
        flip[0] ^= 1
        ff.prepare(pt, P, a.pre, a.pre + F * P - 1, F * P, L)
        return ff(h2 if flip[0] else h, [cos, sin], None)[0]

    def timed(n):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for _ in range(5):
            step()
        marks[0].record()
        for i in range(n):
            out = step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        return [marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(n)], out.shape[1]

    for _ in range(60):          # leave the post-idle transient behind
        step()
    res = {0: [], 1: []}
    outs = {}
    for b in range(a.blocks):
        for mode in (1, 0):
            lib.ff_set_fused_launch(mode)
            us, lo = timed(a.steps)
            res[mode] += us
            outs[mode] = lo
    lib.ff_set_fused_launch(1)
    assert outs[0] == outs[1]
    for mode in (0, 1):
        v = sorted(res[mode])
        print(f"{'fused' if mode else 'two launches'}: {F}x{P}x{d} {L}->{outs[mode]}: median {statistics.median(v):.1f} us  mean {sum(v)/len(v):.1f}  "
              f"min {v[0]:.1f}  p90 {v[int(0.9 * (len(v) - 1))]:.1f}  ({len(v)} steps)")


if __name__ == "__main__":
    main()
