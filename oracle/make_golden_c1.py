#!/usr/bin/env python
"""Golden vectors for BASELINE.json configs[0]: synthetic hidden_states [1, 8x576 (+14+20 text),
1024] fp32 through the REAL reference's FrameFusion merge call on torch-cpu (SURVEY.md §8d "C1").

Runs ONLY in the build container (imports /root/reference/framefusion/main.py).  Asserts the oracle
reproduces the reference bit for bit, then stores what a travelling test can check without the
reference: kept indices, similarities, state flags, a SHA-256 of the output activations and 64
sampled output rows -> tests/golden/c1.npz.

    python oracle/make_golden_c1.py
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import framefusion.main as ref                       # noqa: E402
from oracle import ff_oracle as orc                  # noqa: E402
from framefusion_amd.synth import video_tokens       # noqa: E402

OUT = os.path.join(os.environ.get("FF_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden"), "c1.npz")
F, P, D, PRE, POST, SEED = 8, 576, 1024, 14, 20, 1234


def main():
    store = {}
    for name, p_change, sigma_hi in (("topk", 0.2, None), ("thr", 0.5, 1.6), ("low", 0.95, None)):
        h, pt = video_tokens(F, P, D, p_change=p_change, sigma=0.3, sigma_hi=sigma_hi, seed=SEED, pre=PRE, post=POST,
                             dtype=torch.float32)
        L = h.shape[1]
        r = ref.FrameFusion(0.3, 0.6, 0.1)
        r.prepare(pt.clone(), P, PRE, PRE + F * P - 1, F * P, L)
        pos_r = torch.arange(L)[None]
        hr, pr, _ = r.forward(h.clone(), pos_r, None)
        o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        o.prepare(pt.clone(), P, PRE, PRE + F * P - 1, F * P, L)
        ho, po, _ = o.forward(h.clone(), torch.arange(L)[None], None)
        assert torch.equal(pr, po) and torch.equal(hr, ho), name          # fp32: bitwise
        assert (r.finish_merging, r.finish_pruning, r.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list)
        sim_r, order_r = ref.FrameFusion.compute_similarity_and_token_index_by_patch(h.clone(), pt, P)
        assert torch.equal(sim_r, o.last_sim) and torch.equal(order_r, o.last_order)
        v = torch.sort(sim_r[0], descending=True).values
        k = int(o.last_merge_idx.numel())
        unique_cut = not (r.finish_pruning and 0 < k < v.numel() and bool(v[k - 1] == v[k]))
        rows = torch.linspace(0, hr.shape[1] - 1, 64).long()
        store[f"{name}/cfg"] = np.asarray([F, P, D, PRE, POST, SEED, int(unique_cut)], dtype=np.int64)
        store[f"{name}/fcfg"] = np.asarray([p_change, -1.0 if sigma_hi is None else sigma_hi], dtype=np.float64)
        store[f"{name}/kept"] = pr[0].numpy().astype(np.int32)
        store[f"{name}/sim"] = sim_r[0].numpy().copy()
        store[f"{name}/flags"] = np.asarray([int(r.finish_merging), int(r.finish_pruning)], dtype=np.int64)
        store[f"{name}/sparsity"] = np.asarray(r.sparsity_list, dtype=np.float64)
        store[f"{name}/rows_idx"] = rows.numpy()
        store[f"{name}/rows"] = hr[0, rows].numpy().copy()
        store[f"{name}/sha256"] = np.frombuffer(hashlib.sha256(hr.numpy().tobytes()).digest(), dtype=np.uint8).copy()
        print(f"{name}: L={L} -> {hr.shape[1]}  flags={r.finish_merging, r.finish_pruning} sparsity={r.sparsity_list} "
              f"unique_cut={unique_cut}")
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
