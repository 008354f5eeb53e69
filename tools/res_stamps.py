#!/usr/bin/env python
"""Phase stamps of the one-launch merge kernel (workgroup 0, csrc/ff_resident.hip): stats[FF_STAT_T_PLAN..] after single calls.
    python tools/res_stamps.py [--F 64 --P 210 --d 3584 --pre 14 --post 20] [--views]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables

ap = argparse.ArgumentParser()
ap.add_argument("--F", type=int, default=64); ap.add_argument("--P", type=int, default=210); ap.add_argument("--d", type=int, default=3584)
ap.add_argument("--pre", type=int, default=14); ap.add_argument("--post", type=int, default=20); ap.add_argument("--p_change", type=float, default=0.2)
ap.add_argument("--views", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
h, pt = video_tokens(a.F, a.P, a.d, p_change=a.p_change, seed=1234, pre=a.pre, post=a.post)
L = h.shape[1]
h, pt = h.to(dev), pt.to(dev)
pe = [t.to(dev) for t in rotary_tables(L, 128, torch.bfloat16)]
ff = ffa.FrameFusion(compact_outputs=not a.views)
subs = ["keymasks", "tieslot", "poswords", "published", "ldsrows", "vgprrows", "contin", "outputs"]
names = ["rows+sims", "barrier", "decision", "plan", "fold", "roles", "end"]
for it in range(6):
    ff.prepare(pt, a.P, a.pre, a.pre + a.F * a.P, a.F * a.P, L)
    torch.cuda.synchronize()
    out, _, _ = ff(h, [t.clone() for t in pe], None)
    torch.cuda.synchronize()
    sc = ff.last_call["scratch"]
    st = sc.stats.cpu().tolist()
    print(f"call {it}: one_launch={ff.last_call['one_launch']} applied={ff.last_call['applied']} wait_us={ff.last_call['wait_ns'] / 1e3:.1f} {L}->{out.shape[1]}  " +
          "  ".join(f"{n} {st[_lib.STAT_T_PLAN + x] / 100:.1f}" for x, n in enumerate(names)) + "\n        " +
          "  ".join(f"{n} {st[_lib.STAT_T_ORDER + x] / 100:.1f}" for x, n in enumerate(subs)))
print("mail slot taken by the last call:", ff.last_call["mail_slot"])
