"""Per-step durations of the bench step from a COLD process (no warm-up at all): is a short run (the driver's
`--steps 20 --warmup 5`) still on the ramp?  Events between consecutive steps on the launch stream.
    python tools/coldstart.py [steps = 80] [idle seconds before the second series = 2.0]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
import framefusion_amd as ffa                           # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables   # noqa: E402


def series(step, n):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(n):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    return [marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(n)], wall


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    dev = torch.device("cuda", 0)
    F, P, d = bench.FRAMES, bench.PATCHES, bench.DIM
    hidden, ptype = video_tokens(F, P, d, p_change=bench.P_CHANGE, sigma=bench.SIGMA, seed=1234, dtype=torch.bfloat16,
                                 device=str(dev))
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, bench.HEAD_DIM, torch.bfloat16, device=str(dev))
    ff = ffa.FrameFusion(bench.COST, bench.THRESHOLD, bench.RATIO_LB, **bench.VIEWS)
    alt = hidden.clone()
    flip = [0]

    def step():
        flip[0] ^= 1
        ff.prepare(ptype, P, 0, L, L, L)
        return ff(alt if flip[0] else hidden, [cos, sin], None)[0]

    torch.cuda.synchronize()
    for label, pause in (("cold process", 0.0), (f"after {idle} s idle", idle), ("after 0.2 s idle", 0.2), ("no idle", 0.0)):
        time.sleep(pause)
        us, wall = series(step, n)
        head = " ".join(f"{u:6.1f}" for u in us[:25])
        tail = sorted(us[25:])
        print(f"{label:>20}: wall {wall:6.1f} us/step | first 25: {head} | steps 26..{n}: median {tail[len(tail) // 2]:.1f} "
              f"min {tail[0]:.1f} max {tail[-1]:.1f}")
        for k in (5, 10, 20):
            print(f"{'':>20}  mean of steps {k + 1}..{k + 20}: {sum(us[k:k + 20]) / 20:.1f} us")


if __name__ == "__main__":
    main()
