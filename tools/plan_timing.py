#!/usr/bin/env python
"""Phase stamps of the plan kernel (last workgroup, 100 MHz steady counter -> us since kernel entry) inside
the real step: after the first barrier (round-1 loads landed), after the k-th key, after t*, after the
pass over the slots, at the end (development tool)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables

dev = "cuda:0"
F, P, d = 64, 576, 4096
p_change = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, seed=1234, device=dev)
h2 = h.clone()
L = h.shape[1]
cos, sin = rotary_tables(L, 128, device=dev)
ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)
acc = []
for it in range(60):
    ff.prepare(pt, P, 0, L, L, L)
    out, _, _ = ff(h2 if it & 1 else h, [cos, sin], None)
    torch.cuda.synchronize()
    sc = ff.last_call["scratch"]
    dstats = sc.stats.cpu().numpy()        # (the diagnostic words stay on the device: the pinned words behind 15 are the host's)
    acc.append(np.concatenate((np.array(dstats[_lib.STAT_T_PLAN:_lib.STAT_T_PLAN + 7], dtype=np.float64),
                               np.array(dstats[_lib.STAT_T_ORDER:_lib.STAT_T_ORDER + 3], dtype=np.float64))) / 100.0)
a = np.stack(acc[10:])
names = ["loads issued + LDS filled", "first barrier", "level 0", "level 1 + t*", "classified", "exchanged", "end", "[last wave started", "first wave ready", "last wave ready]"]
pk = int(dstats[_lib.STAT_T_ORDER + 3])
print("per-wave ready (x0.4 us):", [(pk >> (4 * x)) & 15 for x in range(16)])
print(f"L={L} -> {out.shape[1]}; plan kernel phase stamps (us since entry, last workgroup): " +
      ", ".join(f"{n} {v:.2f}" for n, v in zip(names, a.mean(0))))
