#!/usr/bin/env python
"""Host-side cost of a FrameFusion.forward merge call: tiny tensors (the GPU work is a few us), many
calls, cProfile of the Python path."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa                                   # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables   # noqa: E402

DEV = "cuda:0"
F, P, d = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 64, 256)))
h, pt = video_tokens(F, P, d, p_change=0.3, seed=1, pre=4, post=4, dtype=torch.bfloat16, device=DEV)
L = h.shape[1]
cos, sin = rotary_tables(L, 64, torch.bfloat16, device=DEV)
ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)


def call():
    ff.prepare(pt, P, 4, 4 + F * P - 1, F * P, L)
    return ff(h, [cos, sin], None)


for _ in range(200):
    call()
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    call()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e6:.1f} us per prepare+forward at L={L}")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    call()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
