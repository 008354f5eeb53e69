#!/usr/bin/env python
"""Golden vectors for the token-layout builders.

Runs ONLY in the build container.  The reference's adapters cannot be imported here (they need
llava / VILA / transformers 4.45 internals), but the blocks that build ``patch_type`` are a few
self-contained statements between FRAMEFUSION markers.  This script cuts those statements out of
the adapter sources BY LINE RANGE at generation time, executes them against stand-in inputs (tiny
fake `self`, tensors of the right shapes), records what they pass to ``framefusion.prepare`` and
asserts equality with oracle/layout_oracle.py.  Only inputs and recorded outputs are written to
tests/golden/layout.npz; no reference text is stored.

    python oracle/make_golden_layout.py
"""
from __future__ import annotations

import collections
import math
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/framefusion/models"
sys.path.insert(0, ROOT)

from oracle import layout_oracle as lay                # noqa: E402

OUT = os.path.join(os.environ.get("FF_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden"), "layout.npz")
TEXT_TOKEN, IMAGE_TOKEN_INDEX = -1, -200


def block(rel: str, first: int, last: int, must_contain: str):
    src = open(os.path.join(REF, rel)).read().splitlines()[first - 1:last]
    text = textwrap.dedent("\n".join(src))
    assert must_contain in text, f"{rel}:{first}-{last} moved (expected {must_contain!r})"
    return compile(text, f"<reference {rel}:{first}-{last}>", "exec")


class Recorder:
    def prepare(self, *args):
        self.args = args


def as_int(v):
    if isinstance(v, torch.Tensor):
        assert v.numel() == 1
        return int(v.reshape(-1)[0].item())
    assert float(v) == int(v)
    return int(v)


def record(rec: Recorder):
    ptype, patch_num, start, end, itl, orig = rec.args
    assert ptype.dtype == torch.int64 and ptype.shape[0] == 1
    return ptype[0].tolist(), as_int(patch_num), as_int(start), as_int(end), as_int(itl), as_int(orig)


def same(ref_out, orc_out):
    assert ref_out[0] == [int(v) for v in orc_out[0]], "patch_type differs"
    assert ref_out[1:] == tuple(int(v) for v in orc_out[1:]), (ref_out[1:], orc_out[1:])


# ---- LLaVA-Video ------------------------------------------------------------------------------
def run_llava_video(pre, post, frames, side, mode):
    code = block("llava_video/modeling_llava_video.py", 322, 338, "patch_num = patch_size * (patch_size + 1)")
    ps = math.ceil(side / 2) if mode == "bilinear" else side // 2
    itl = frames * ps * (ps + 1)
    ids = list(range(100, 100 + pre)) + [IMAGE_TOKEN_INDEX] + list(range(200, 200 + post))
    rec = Recorder()
    fake = types.SimpleNamespace(config=types.SimpleNamespace(mm_spatial_pool_mode=mode), framefusion=rec,
                                 get_vision_tower=lambda: types.SimpleNamespace(num_patches_per_side=side))
    ns = dict(math=math, torch=torch, TEXT_TOKEN=TEXT_TOKEN, IMAGE_TOKEN_INDEX=IMAGE_TOKEN_INDEX, self=fake,
              batch_size=1, num_images=1, image_features=[torch.zeros(itl, 4)], input_ids=torch.tensor([ids]),
              new_input_embeds=torch.zeros(1, pre + itl + post, 4))
    exec(code, ns)
    out = record(rec)
    same(out, lay.llava_video(ids, IMAGE_TOKEN_INDEX, itl, side, mode))
    return dict(ids=ids, itl=itl, side=side, bilinear=int(mode == "bilinear")), out


# ---- Qwen2-VL ---------------------------------------------------------------------------------
def run_qwen2_vl(pre, post, t, h, w, merge):
    code = block("qwenvl/modeling_qwen2_vl.py", 118, 137, "patch_num = (video_grid_thw[0,1] * video_grid_thw[0,2])")
    vid = 151656
    n = t * (h * w) // (merge * merge)
    ids = list(range(10, 10 + pre)) + [vid] * n + list(range(50, 50 + post))
    rec = Recorder()
    fake = types.SimpleNamespace(config=types.SimpleNamespace(video_token_id=vid), framefusion=rec,
                                 visual=types.SimpleNamespace(config=types.SimpleNamespace(spatial_merge_size=merge)))
    ns = dict(torch=torch, TEXT_TOKEN=TEXT_TOKEN, self=fake, inputs_embeds=torch.zeros(1, len(ids), 4),
              video_grid_thw=torch.tensor([[t, h, w]]), video_embeds=torch.zeros(n, 4), input_ids=torch.tensor([ids]))
    exec(code, ns)
    out = record(rec)
    same(out, lay.qwen2_vl(ids, vid, h, w, merge, n, len(ids)))
    return dict(ids=ids, grid=[t, h, w], merge=merge, n=n), out


# ---- MiniCPM-V --------------------------------------------------------------------------------
def run_minicpmv(pre, post, frames, slices, slice_tokens, sep):
    code = block("minicpmv/modeling_minicpmv.py", 91, 106, "token_per_frame = image_bound[patch_per_frame, 0]")
    bounds, pos = [], pre
    for _ in range(frames):
        for _ in range(slices):
            pos += 1                                  # <image> marker
            bounds.append((pos, pos + slice_tokens))
            pos += slice_tokens + 1                   # </image>
        pos += sep
    length = pos - sep + post
    rec = Recorder()
    fake = types.SimpleNamespace(num_frames=frames, framefusion=rec)
    ns = dict(torch=torch, TEXT_TOKEN=TEXT_TOKEN, self=fake, bs=1, i=0, vllm_embedding=torch.zeros(1, length, 4),
              data={"image_bound": [torch.tensor(bounds)]})
    exec(code, ns)
    out = record(rec)
    same(out, lay.minicpmv(bounds, frames, length))
    return dict(bounds=bounds, frames=frames, length=length), out


# ---- InternVL ---------------------------------------------------------------------------------
def run_internvl(pre, post, frames, patch_num, gaps):
    code = block("internvl/modeling_internvl_chat.py", 38, 82, "text_length_list = count_consecutive_false(selected)[1:-1]")
    sel = [False] * pre
    for f in range(frames):
        sel += [True] * patch_num
        if f + 1 < frames:
            sel += [False] * gaps[f % len(gaps)]
    sel += [False] * post
    N = len(sel)
    rec = Recorder()
    fake = types.SimpleNamespace(framefusion=rec)
    ns = dict(torch=torch, TEXT_TOKEN=TEXT_TOKEN, self=fake, selected=torch.tensor(sel), N=N,
              vit_embeds=torch.zeros(frames, patch_num, 4), input_ids=torch.zeros(1, N, dtype=torch.long),
              input_embeds=torch.zeros(1, N, 4))
    exec(code, ns)
    out = record(rec)
    same(out, lay.internvl(sel, frames, patch_num))
    return dict(selected=[int(s) for s in sel], frames=frames, patch_num=patch_num), out


# ---- NVILA ------------------------------------------------------------------------------------
def run_nvila(pre, post, media_frames, pool, tokens_per_frame, kind):
    head = block("nvila/llava_arch.py", 41, 58, "patch_num = media_embeds['video'][0].shape[0] / num_frames")
    body = block("nvila/llava_arch.py", 86, 88, "patch_type[k, inputs_mk[0].shape[0]")
    tail = block("nvila/llava_arch.py", 92, 103, "image_token_length = image_token_end_index - image_token_start_index + 1")
    if kind == "video":
        n_feat = (media_frames // pool) * tokens_per_frame
        media = {"video": [torch.zeros(media_frames, 3)]}
        media_embeds = {"video": collections.deque([torch.zeros(n_feat, 4)])}
        cfg = types.SimpleNamespace(video_encoder={"pool_sizes": [[pool]]} if pool != 1 else "basic")
        chunks = [pre, n_feat, post]
        n_placeholders = 1
    else:
        n_feat = media_frames * tokens_per_frame
        media = {"image": [torch.zeros(3)] * media_frames}
        media_embeds = {"image": collections.deque(torch.zeros(tokens_per_frame, 4) for _ in range(media_frames))}
        cfg = types.SimpleNamespace(video_encoder="basic")
        chunks = [pre] + [tokens_per_frame] * media_frames + [post]
        n_placeholders = media_frames
    rec = Recorder()
    fake = types.SimpleNamespace(config=cfg, framefusion=rec)
    ns = dict(torch=torch, TEXT_TOKEN=TEXT_TOKEN, self=fake, batch_size=1, media=media, media_embeds=media_embeds,
              text_embeds=[torch.zeros(pre + post + n_placeholders, 4)], k=0,
              inputs_mk=[torch.zeros(c, 4) for c in chunks])
    exec(head, ns)
    exec(body, ns)
    exec(tail, ns)
    out = record(rec)
    same(out, lay.nvila(chunks, n_feat, media_frames, pool, kind))
    return dict(chunks=chunks, n_feat=n_feat, media_frames=media_frames, pool=pool, video=int(kind == "video")), out


CASES = [
    ("llava_video/bilinear27", run_llava_video, (14, 20, 4, 27, "bilinear")),
    ("llava_video/average24", run_llava_video, (5, 9, 3, 24, "average")),
    ("llava_video/no_text_before", run_llava_video, (0, 3, 2, 6, "bilinear")),
    ("qwen2_vl/360x420", run_qwen2_vl, (15, 12, 4, 26, 30, 2)),
    ("qwen2_vl/odd_grid", run_qwen2_vl, (3, 1, 2, 6, 10, 2)),
    ("qwen2_vl/no_text_after", run_qwen2_vl, (4, 0, 3, 4, 4, 2)),
    ("minicpmv/one_slice", run_minicpmv, (6, 11, 4, 1, 16, 0)),
    ("minicpmv/three_slices", run_minicpmv, (3, 5, 3, 3, 8, 2)),
    ("internvl/uniform_gaps", run_internvl, (7, 9, 4, 16, [5])),
    ("internvl/growing_gaps", run_internvl, (2, 4, 5, 8, [3, 4, 6])),
    ("nvila/video", run_nvila, (9, 13, 8, 1, 16, "video")),
    ("nvila/video_pooled", run_nvila, (4, 6, 8, 2, 12, "video")),
    ("nvila/images", run_nvila, (5, 3, 3, 1, 7, "image")),
]


def main():
    store = {}
    for name, fn, args in CASES:
        inputs, (ptype, patch_num, start, end, itl, orig) = fn(*args)
        for key, val in inputs.items():
            store[f"{name}/in/{key}"] = np.asarray(val, dtype=np.int64)
        store[f"{name}/patch_type"] = np.asarray(ptype, dtype=np.int64)
        store[f"{name}/scalars"] = np.asarray([patch_num, start, end, itl, orig], dtype=np.int64)
        print(f"{name:28s} L={len(ptype):5d} P={patch_num:4d} start={start:3d} end={end:5d} n={itl:5d} orig={orig}")
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
