#!/usr/bin/env python
"""Golden vectors for the `attn_mask` argument of utils.scaled_dot_product_attention (framefusion/utils.py:40-44):
boolean and additive masks, broadcast [1, S] and full [num, S].  Runs ONLY in the build container (the reference
function is compiled from the file's AST at generation time, as in make_golden.py; nothing of it is written to
disk), asserts reference == oracle bit for bit and writes tests/golden/importance_mask.npz.

    python oracle/make_golden_mask.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_golden import DT, OUT, bits, load_reference_sdpa, one_thread, same     # noqa: E402
from oracle import ff_oracle as orc                                   # noqa: E402
from tests import harness                                             # noqa: E402


def main():
    sdpa = load_reference_sdpa()
    out = {}
    #        name        H  Hk  S   dh num  kind        dtype
    specs = [("m_bool",   4, 4, 45, 16, 1, "bool_row", "bf16"), ("m_bool4", 8, 2, 70, 32, 4, "bool_full", "bf16"),
             ("m_add",    4, 4, 33, 16, 4, "add_full", "fp16"), ("m_add32", 2, 2, 29, 8, 1, "add_row", "fp32"),
             ("m_addwide", 6, 3, 300, 64, 4, "add_full_f32", "bf16")]
    for name, H, Hk, S, dh, num, kind, dt in specs:
        dtype = DT[dt]
        g = torch.Generator().manual_seed(900 + S)
        q = harness.snap(torch.randn(1, H, S, dh, generator=g), dtype)
        k = harness.snap(torch.randn(1, Hk, S, dh, generator=g), dtype)
        rows = 1 if kind.endswith("row") else num
        if kind.startswith("bool"):
            mask = torch.rand(rows, S, generator=g) < 0.7
            mask[:, 0] = True                                  # no fully masked row (softmax of all -inf is NaN in both)
        else:
            mask = harness.snap(torch.randn(rows, S, generator=g), torch.float32 if kind.endswith("f32") else dtype)
            mask[:, 3] = float("-inf")
        gqa = H != Hk
        with one_thread():
            w_r = sdpa(q, k, torch.zeros_like(k), num=num, attn_mask=mask, enable_gqa=gqa)
            w_o = orc.last_query_attention(q, k, num=num, enable_gqa=gqa, attn_mask=mask)
        assert same(w_r, w_o), name
        out[f"{name}/q"], out[f"{name}/k"] = bits(q[0]), bits(k[0])
        out[f"{name}/mask"] = mask.numpy().copy() if mask.dtype in (torch.bool, torch.float32) else bits(mask)
        out[f"{name}/mask_dtype"] = np.array({torch.bool: "bool", torch.float32: "fp32"}.get(mask.dtype, dt))
        out[f"{name}/meta"] = np.array([H, Hk, S, dh, num])
        out[f"{name}/dtype"] = np.array(dt)
        out[f"{name}/weights"] = bits(w_r[0])
        print(f"  {name}: {kind} mask {tuple(mask.shape)} -> weights {tuple(w_r.shape)}")
    np.savez_compressed(os.path.join(OUT, "importance_mask.npz"), **out)
    print("importance_mask ok")


if __name__ == "__main__":
    main()
