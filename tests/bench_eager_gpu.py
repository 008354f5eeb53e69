#!/usr/bin/env python
"""PyTorch-eager on the MI355X: the oracle's torch restatement of FrameFusion.forward run on GPU
tensors (what running the reference itself through PyTorch-ROCm costs), beside the HIP path, on the
bench workload (64 x 576 x 4096 bf16, cost 0.3, top-k branch).  Measurement aid only - the product
never imports the oracle (which is why this script lives under tests/).
    python tests/bench_eager_gpu.py [--reps 10] [--baseline-cpu]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa                                   # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables   # noqa: E402
from oracle import ff_oracle as orc                             # noqa: E402

DEV = "cuda:0"


def run(make, hidden, pt, P, pos, reps):
    times, out = [], None
    for i in range(reps + 2):
        ff = make()
        ff.prepare(pt, P, 0, hidden.shape[1] - 1, hidden.shape[1], hidden.shape[1])
        h = hidden.clone()
        p = [t.clone() for t in pos]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ff.forward(h, p, None)
        torch.cuda.synchronize()
        if i >= 2:
            times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    return times[len(times) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--p-change", type=float, default=0.2)
    ap.add_argument("--baseline-cpu", action="store_true",
                    help="also time one layer of the fixed-sparsity baseline (sparsity 0.1) on the CPU oracle")
    a = ap.parse_args()
    F, P, d = 64, 576, 4096
    hidden, pt = video_tokens(F, P, d, p_change=a.p_change, sigma=0.3, seed=1234, dtype=torch.bfloat16)
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, 128, torch.bfloat16)
    hd, ptd, pos = hidden.to(DEV), pt.to(DEV), [cos.to(DEV), sin.to(DEV)]
    t_eager, o_e = run(lambda: orc.OracleFrameFusion(0.3, 0.6, 0.1), hd, ptd, P, pos, a.reps)
    t_hip, o_h = run(lambda: ffa.FrameFusion(0.3, 0.6, 0.1), hd, ptd, P, pos, a.reps)
    Le, Lh = o_e[0].shape[1], o_h[0].shape[1]
    print(f"L={L} -> eager {Le}, hip {Lh}")
    print(f"PyTorch eager on MI355X: {t_eager:8.3f} ms/call  ({(L - Le) / t_eager / 1e3:8.2f} M tokens reduced/s)")
    print(f"HIP path               : {t_hip:8.3f} ms/call  ({(L - Lh) / t_hip / 1e3:8.2f} M tokens reduced/s)   x{t_eager / t_hip:.1f}")
    if a.baseline_cpu:
        t0 = time.perf_counter()
        orc.fixed_sparsity_merge(hidden, pt, P, 0.1, [cos, sin], hidden)
        print(f"fixed-sparsity baseline, layer 0, CPU oracle: {(time.perf_counter() - t0) * 1e3:.0f} ms on "
              f"{torch.get_num_threads()} threads")


if __name__ == "__main__":
    main()
