"""Deterministic stand-ins for the model code that surrounds FrameFusion.forward in the
reference's decoder loop (``framefusion/models/qwen2/modeling_qwen2.py:45-46,67,275-305``), so a
whole prefill cascade (call A at layer 0, call B after every layer's attention, the one prune
call fed by attention weights) can be replayed identically through the oracle, the reference
(fixture generation only) and the HIP path.

Everything here is elementwise fp32 arithmetic on a dyadic grid, so the inputs each call sees
are bit-identical on any machine and the similarity sums stay exact (order-independent).
"""
from __future__ import annotations

import torch

GRID = 0.125
CLIP = 4.0


def snap(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 -> multiples of GRID clipped to +-CLIP -> dtype."""
    return (x.float() / GRID).round().mul(GRID).clamp(-CLIP, CLIP).to(dtype)


def layer_stub(hidden: torch.Tensor, layer: int) -> torch.Tensor:
    """What 'attention + residual' does to the activations between two FrameFusion calls: a
    fixed, cheap, deterministic mixing that keeps neighbouring frames similar."""
    h = hidden.float()
    mixed = 0.75 * h + 0.25 * torch.roll(h, shifts=1 + layer, dims=-1)
    return snap(mixed, hidden.dtype)


def attention_stub(heads: int, num: int, length: int, dtype, device="cpu") -> torch.Tensor:
    """[1, H, num, S] attention probabilities of the last `num` queries with a unique ranking
    after the head/query mean (values are small integers over a power of two: the fp32 mean is
    exact, so the ranking is machine-independent)."""
    s = torch.arange(length, device=device, dtype=torch.int64)
    h = torch.arange(heads, device=device, dtype=torch.int64)[:, None, None]
    n = torch.arange(num, device=device, dtype=torch.int64)[None, :, None]
    raw = ((s[None, None, :] * 37 + h * 11 + n * 5) % 127 + 1).float() / 1024.0
    return raw.to(dtype)[None]


def run_cascade(ff, hidden, patch_type, patch_num, position_embeddings, attention_mask,
                layers: int, heads: int = 4, num: int = 1, start=None, n_visual=None):
    """Replays the reference call protocol.  `ff` is any object with prepare()/forward() and the
    finish_merging / finish_pruning flags (oracle, reference or the HIP module).
    Returns a list of per-call records."""
    L = hidden.shape[1]
    pt = patch_type.reshape(-1)
    vis = torch.nonzero(pt >= 0).reshape(-1)
    if start is None:
        start = int(vis[0]) if vis.numel() else 0
    if n_visual is None:
        n_visual = int(vis.numel())
    ff.prepare(patch_type, patch_num, start, start + n_visual, n_visual, L)
    log = []

    def record(tag, h, pe):
        log.append(dict(tag=tag, length=h.shape[1], finish_merging=bool(ff.finish_merging),
                        finish_pruning=bool(ff.finish_pruning),
                        sparsity=list(ff.sparsity_list), hidden=h,
                        pos=list(pe) if isinstance(pe, list) else pe))

    h, pe, mask = ff.forward(hidden, position_embeddings, attention_mask)          # call A
    record("A0", h, pe)
    for layer in range(layers):
        attn_w = None
        if h.shape[1] > 1 and ff.finish_merging and not ff.finish_pruning:          # modeling_qwen2.py:168
            attn_w = attention_stub(heads, num, h.shape[1], h.dtype, h.device)
        h = layer_stub(h, layer)
        h, pe, mask = ff.forward(h, pe, mask, attn_w)                                # call B
        record(f"B{layer}", h, pe)
    return log, mask
