#!/usr/bin/env python
"""K0 (by-patch order) on layouts the closed form does not cover: text between the frames (InternVL-style,
internvl/modeling_internvl_chat.py:59-74) and the ragged patch_type a merge call leaves behind - the counting
sort of ff_order.hip.  hipEvent time per ff_build_order call (development tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd import _lib

dev = "cuda:0"
lib = _lib.load()


def bench(name, pt, P):
    L = pt.numel()
    ptd = pt.to(dev)
    order = torch.empty(L, dtype=torch.int32, device=dev)
    inv = torch.empty(L, dtype=torch.int32, device=dev)
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=dev)
    wsb = int(lib.ff_workspace_bytes(L, P))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    call = lambda: _lib.check(lib.ff_build_order(ptd.data_ptr(), L, P, order.data_ptr(), inv.data_ptr(), stats.data_ptr(),
                                                 ws.data_ptr(), wsb, _lib.stream_ptr()), "ff_build_order")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: L={L} P={P} nv={int(stats[_lib.STAT_NV])}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per ff_build_order")


F, P = 64, 576
frame = torch.arange(P)
bench("frame-major (closed form)", torch.cat([torch.full((14,), -1), frame.repeat(F), torch.full((20,), -1)]), P)
sep = torch.full((3,), -1)
bench("text between the frames (counting sort)", torch.cat([torch.full((14,), -1)] + [torch.cat((frame, sep)) for _ in range(F)]), P)
g = torch.Generator().manual_seed(1)
ragged = torch.cat([torch.full((14,), -1), frame.repeat(F)[torch.rand(F * P, generator=g) < 0.5], torch.full((20,), -1)])
bench("ragged after a merge (counting sort)", ragged, P)
