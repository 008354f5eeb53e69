"""Synthetic video-token workloads: deterministic stand-ins for the ViT tokens a
video-LLM hands to the first decoder layer.  Shared by bench.py, the tests and the golden-vector
generator so every side sees identical inputs for a given seed.

Layout follows the reference's adapters (``framefusion/models/llava_video/modeling_llava_video.py:321-336``):
``pre`` text rows, then F frames of P patch tokens each (frame-major), then ``post`` text rows;
``patch_type`` is -1 on text and the patch position 0..P-1 on visual tokens.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

TEXT_TOKEN = -1


def video_tokens(frames: int, patches: int, dim: int, p_change: float = 0.2, sigma: float = 0.3,
                 seed: int = 1234, pre: int = 0, post: int = 0, dtype=torch.bfloat16,
                 device: str = "cpu", sigma_hi: Optional[float] = None,
                 grid: Optional[float] = None, clip: float = 4.0
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (hidden [1, pre + F*P + post, dim] of ``dtype``, patch_type [1, L] int64).

    Frame f is frame f-1 plus N(0, sigma) noise, except that each patch is replaced, with
    probability ``p_change``, by fresh noise of the previous frame's scale (a scene change at
    that patch).  ``sigma_hi`` draws a per-(frame, patch) sigma from U(sigma, sigma_hi) so the
    similarities spread across the threshold.  ``grid`` snaps the values to multiples of ``grid``
    clipped to +-``clip``: with grid=1/8 every fp32 partial sum of the similarity recipe is exact
    for dim <= 8192, which makes results independent of the reduction order (bit-exact parity
    tests use it).
    """
    gen = torch.Generator(device=device).manual_seed(seed)
    rows = torch.empty(frames, patches, dim, dtype=torch.float32, device=device)
    rows[0] = torch.randn(patches, dim, generator=gen, device=device)
    for f in range(1, frames):
        prev = rows[f - 1]
        if sigma_hi is None:
            s = sigma
        else:
            s = sigma + (sigma_hi - sigma) * torch.rand(patches, 1, generator=gen, device=device)
        drift = prev + s * torch.randn(patches, dim, generator=gen, device=device)
        fresh = prev.std() * torch.randn(patches, dim, generator=gen, device=device)
        cut = torch.rand(patches, 1, generator=gen, device=device) < p_change
        rows[f] = torch.where(cut, fresh, drift)
    parts = []
    if pre:
        parts.append(torch.randn(pre, dim, generator=gen, device=device))
    parts.append(rows.reshape(frames * patches, dim))
    if post:
        parts.append(torch.randn(post, dim, generator=gen, device=device))
    hidden = torch.cat(parts) if len(parts) > 1 else parts[0]
    if grid is not None:
        hidden = (hidden / grid).round_().mul_(grid).clamp_(-clip, clip)
    patch_type = torch.cat((
        torch.full((pre,), TEXT_TOKEN, dtype=torch.long),
        torch.arange(patches, dtype=torch.long).repeat(frames),
        torch.full((post,), TEXT_TOKEN, dtype=torch.long),
    ))
    return hidden.to(dtype)[None], patch_type[None].to(device)


def rotary_tables(length: int, head_dim: int = 128, dtype=torch.bfloat16, device: str = "cpu",
                  mrope: bool = False, base: float = 1e6):
    """[cos, sin] as the decoder stack passes them to FrameFusion.forward: each [1, L, dh]
    (Qwen2, ``modeling_qwen2.py:263-266``) or [3, 1, L, dh] (Qwen2-VL M-RoPE)."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=device) / head_dim))
    ang = torch.arange(length, dtype=torch.float32, device=device)[:, None] * inv[None, :]
    ang = torch.cat((ang, ang), dim=-1)
    cos, sin = ang.cos().to(dtype), ang.sin().to(dtype)
    if mrope:
        return [cos[None, None].repeat(3, 1, 1, 1).contiguous(), sin[None, None].repeat(3, 1, 1, 1).contiguous()]
    return [cos[None].contiguous(), sin[None].contiguous()]
