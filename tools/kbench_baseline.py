#!/usr/bin/env python
"""Fixed-sparsity merging baseline at full size: per-layer GPU time of the [0.1] * 28 schedule on
64 x 576 x 4096 bf16 (normed activations merged, residual + cos/sin + patch_type compacted).
    python tools/kbench_baseline.py [--layers 28] [--sparsity 0.1]
(the CPU oracle of the same layer is timed by tests/bench_eager_gpu.py --baseline-cpu)
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd.baseline import FixedSparsityMerging, compute_density_overhead   # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables                          # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--patches", type=int, default=576)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--sparsity", type=float, default=0.1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-residual", action="store_true")
    a = ap.parse_args()
    hidden, ptype = video_tokens(a.frames, a.patches, a.dim, p_change=0.3, sigma=0.3, seed=1234, dtype=torch.bfloat16)
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, 128, torch.bfloat16)
    hd, pt = hidden.to(DEV), ptype.to(DEV)
    res0 = None if a.no_residual else hd.clone()
    sched = [a.sparsity] * a.layers
    m = FixedSparsityMerging(sched)
    per_layer = None
    for rep in range(a.reps + 1):
        m.prepare(pt, a.patches)
        h, r, pos = hd, res0, [cos.to(DEV), sin.to(DEV)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.layers + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        lens = []
        for layer in range(a.layers):
            h, mask, r = m.merge(layer, h, pos, r)
            ev[layer + 1].record()
            lens.append(h.shape[1])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if rep:
            t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(a.layers)]
            per_layer = t if per_layer is None else [min(x, y) for x, y in zip(per_layer, t)]
    print(f"L={L} schedule {a.sparsity} x {a.layers}: density {compute_density_overhead(sched)}")
    print("lengths:", lens[:4], "...", lens[-1])
    print("per-layer us:", " ".join(f"{x:.0f}" for x in per_layer))
    esz = 2
    rows = [L] + lens
    byt = [(rows[i] + rows[i + 1]) * a.dim * esz * (1 if a.no_residual else 2) for i in range(a.layers)]
    print("algorithmic GB/s per layer:", " ".join(f"{b / (t * 1e-6) / 1e9:.0f}" for b, t in zip(byt, per_layer)))
    print(f"whole schedule: {sum(per_layer) / 1e3:.2f} ms GPU, {wall:.2f} ms wall (last rep)")


if __name__ == "__main__":
    main()
