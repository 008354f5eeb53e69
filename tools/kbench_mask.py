#!/usr/bin/env python
"""Attention-mask gather of a merge call (main.py:137-138: mask[:, :, keep][:, :, :, keep]) at the C3 and C2 lengths with a
dense bf16 [1, 1, L, L] mask: the two-level gather (k_invert_dst + k_gather_mask, ff_merge.hip) against torch's own
double index on the same device.  hipEvent time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd import _lib                                # noqa: E402

dev = "cuda:0"
lib = _lib.load()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for L, keep_frac in ((12507, 0.5), (36898, 0.3)):
    g = torch.Generator(device=dev).manual_seed(L)
    mask = torch.randn(1, 1, L, L, generator=g, device=dev, dtype=torch.bfloat16)
    keep = torch.rand(L, generator=g, device=dev) < keep_frac
    idx = torch.nonzero(keep).reshape(-1)
    n = idx.numel()
    dst = torch.full((L,), -1, dtype=torch.int32, device=dev)
    dst[idx] = torch.arange(n, dtype=torch.int32, device=dev)
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=dev)
    stats[_lib.STAT_LOUT] = n
    scratch = torch.empty(L, dtype=torch.int32, device=dev)
    out = torch.empty(1, 1, n, n, dtype=mask.dtype, device=dev)

    def hip():
        _lib.check(lib.ff_gather_mask(mask.data_ptr(), out.data_ptr(), 2, L, n, dst.data_ptr(), stats.data_ptr(),
                                      scratch.data_ptr(), _lib.stream_ptr()), "ff_gather_mask")

    def eager():
        return mask[:, :, idx, :][:, :, :, idx]
    t_hip = timeit(hip)
    ref = eager()
    assert torch.equal(out, ref)
    t_eager = timeit(eager)
    written = n * n * 2
    print(f"L={L} -> {n}: two-level gather {t_hip:8.1f} us ({written / 1e6:.0f} MB written, {written / t_hip / 1e3:.0f} GB/s of output)   "
          f"torch double index {t_eager:8.1f} us")
