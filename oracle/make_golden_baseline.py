#!/usr/bin/env python
"""Golden vectors for the fixed-sparsity merging baseline.

Runs ONLY in the build container.  The baseline lives inside an attention forward of
framefusion/models/qwen2/modeling_qwen2_baseline.py, which cannot be imported here (transformers
4.45 internals).  Its token-merging block (:905-1053) is self-contained: this script cuts it out
BY LINE RANGE at generation time, executes it on seeded inputs with stand-ins for `self` / `model`,
asserts bit-equality with oracle/ff_oracle.fixed_sparsity_merge and writes inputs' seeds + outputs
to tests/golden/baseline.npz.  compute_density_overhead (:26-43) is taken from the file's AST.
No reference text is stored.

    python oracle/make_golden_baseline.py
"""
from __future__ import annotations

import ast
import math
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import framefusion.main as ref                       # noqa: E402
from oracle import ff_oracle as orc                  # noqa: E402
from framefusion_amd.synth import video_tokens       # noqa: E402

SRC = os.path.join(REF, "framefusion", "models", "qwen2", "modeling_qwen2_baseline.py")
OUT = os.path.join(os.environ.get("FF_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden"), "baseline.npz")
DT = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}


def bits(t):
    t = t.detach().contiguous()
    return t.view(torch.int16).numpy().copy() if t.dtype in (torch.bfloat16, torch.float16) else t.numpy().copy()


def merging_block():
    lines = open(SRC).read().splitlines()[905:1052]            # :906-1052, between the two markers
    text = textwrap.dedent("\n".join(lines))
    assert "prune_num = math.floor(sparsity * frame_token_num)" in text and ".mean(" in text, "block moved"
    return compile(text, "<reference modeling_qwen2_baseline.py:906-1052>", "exec")


def density_fn():
    tree = ast.parse(open(SRC).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "compute_density_overhead"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference compute_density_overhead>", "exec"), ns)
    return ns["compute_density_overhead"]


def run_reference(code, hidden, patch_type, patch_num, sparsity):
    model = types.SimpleNamespace(sparsity=[sparsity], patch_type=patch_type.clone(), patch_num=patch_num)
    me = types.SimpleNamespace(layer_idx=0, hidden_size=hidden.shape[-1])
    ns = dict(torch=torch, math=math, TEXT_TOKEN=ref.TEXT_TOKEN, IGNORE_TOKEN=ref.IGNORE_TOKEN,
              find_contigious_latter_index=ref.find_contigious_latter_index, self=me, model=model,
              hidden_states=hidden.clone(), bsz=1, q_len=hidden.shape[1])
    exec(code, ns)
    return ns["hidden_states"], ns["token_mask"], model.patch_type, ns.get("similarity_by_patch"), ns.get("prune_num")


CASES = [
    # name, F, P, d, dtype, p_change, sparsity, pre, post, grid
    ("bf16_grid_s10", 12, 16, 64, "bf16", 0.3, 0.10, 3, 5, True),
    ("bf16_grid_s45", 16, 12, 64, "bf16", 0.2, 0.45, 2, 2, True),
    ("bf16_gauss_s30", 10, 24, 128, "bf16", 0.4, 0.30, 4, 6, False),
    ("fp32_gauss_s25", 8, 16, 64, "fp32", 0.4, 0.25, 0, 3, False),
    ("fp16_grid_s20", 9, 8, 64, "fp16", 0.3, 0.20, 1, 0, True),
    ("bf16_tiny_s0", 4, 4, 32, "bf16", 0.3, 0.01, 1, 1, True),       # prune_num == 0: untouched
]


def unique_cut(sim, k):
    v = torch.sort(sim.float(), descending=True).values
    return k == 0 or k >= v.numel() or bool(v[k - 1] != v[k])


def main():
    code = merging_block()
    dens = density_fn()
    store = {}
    for name, F, P, d, dt, pc, sp, pre, post, grid in CASES:
        for seed in range(100, 400):
            hidden, ptype = video_tokens(F, P, d, p_change=pc, sigma=0.3, seed=seed, pre=pre, post=post, dtype=DT[dt],
                                         grid=0.125 if grid else None)
            sim, _ = orc.pair_similarity(hidden, ptype, P)
            k = math.floor(sp * int((ptype != -1).sum()))
            if unique_cut(sim[0], k):
                break
        else:
            raise RuntimeError(f"{name}: no seed with a unique top-k cut")
        h_ref, mask_ref, pt_ref, sim_ref, k_ref = run_reference(code, hidden, ptype, P, sp)
        o = orc.fixed_sparsity_merge(hidden, ptype, P, sp)
        assert k_ref == o["prune_num"] == k, name
        if k == 0:
            assert mask_ref is None and o["token_mask"] is None and torch.equal(h_ref, hidden)
        else:
            assert torch.equal(mask_ref, o["token_mask"]), name
            assert np.array_equal(bits(sim_ref), bits(o["sim"])), name
            assert np.array_equal(bits(h_ref), bits(o["hidden"])), name
            assert torch.equal(pt_ref, o["patch_type"]), name
        store[f"{name}/cfg"] = np.asarray([F, P, d, pre, post, seed, int(grid), k], dtype=np.int64)
        store[f"{name}/fcfg"] = np.asarray([pc, sp], dtype=np.float64)
        store[f"{name}/hidden"] = bits(h_ref)
        store[f"{name}/patch_type"] = pt_ref.numpy().copy()
        store[f"{name}/mask"] = (mask_ref if mask_ref is not None else torch.ones(1, hidden.shape[1], dtype=torch.bool)).numpy().copy()
        if k:
            store[f"{name}/sim"] = bits(sim_ref)
        runs = orc.run_lengths((~o["token_mask"][0, o["order"][0]]).long()[None])[0].max().item() if k else 0
        print(f"{name:18s} seed={seed} L={hidden.shape[1]} k={k} L_out={h_ref.shape[1]} longest run={runs}")
    lists = [[0.1] * 28, [0.0] * 28, [0.5, 0.25, 0.0, 0.1], [0.3]]
    for i, sl in enumerate(lists):
        a = dens(sl)
        b = orc.density_overhead(sl)
        assert a == b, (a, b)
        store[f"density/{i}/in"] = np.asarray(sl, dtype=np.float64)
        store[f"density/{i}/out"] = np.asarray(a, dtype=np.float64)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
