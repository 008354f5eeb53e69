#!/usr/bin/env python
"""Layout builders on the GPU vs the reference's way (Python list + upload) at full size.
    python tools/kbench_layout.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd import layout as L       # noqa: E402

DEV = "cuda:0"
VID = 151656


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def list_way(ids, P):
    """What the packers do (qwenvl/modeling_qwen2_vl.py:123-127): two torch.where, a list, an upload."""
    hits = torch.where(ids[0] == VID)[0]
    start, end = hits[0], hits[-1]
    n_frames = hits.numel() // P
    row = [-1] * start + list(range(P)) * n_frames + [-1] * (ids.shape[1] - end - 1)
    return torch.tensor([row], device=ids.device)


def main():
    for F, P, grid in ((64, 576, (64, 48, 48)), (64, 195, (64, 26, 30)), (128, 576, (128, 48, 48))):
        ids = torch.tensor([[7] * 14 + [VID] * (F * P) + [9] * 20], device=DEV)
        a = L.qwen2_vl_layout(ids, VID, grid, 2).patch_type
        b = list_way(ids, P)
        assert torch.equal(a, b)
        t_hip = timed(lambda: L.qwen2_vl_layout(ids, VID, grid, 2))
        t_fill = timed(lambda: L.fill_patch_type(ids.shape[1], [(14, F * P, 0, P)], DEV))
        t_list = timed(lambda: list_way(ids, P), n=10, warm=2)
        print(f"F={F:4d} P={P:4d} L={ids.shape[1]:6d}: hip layout {t_hip:8.1f} us (fill alone {t_fill:6.1f} us)   "
              f"list+upload {t_list:9.1f} us   x{t_list / t_hip:.0f}")


if __name__ == "__main__":
    main()
