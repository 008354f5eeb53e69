"""Token layouts: the ``patch_type`` row and the five scalars of ``FrameFusion.prepare``.

In the reference every multimodal packer ends with a FRAMEFUSION block that locates the visual span
with ``torch.where``, builds ``patch_type`` as a Python list of L ints, uploads it and calls
``self.framefusion.prepare(...)``:

    llava_video  framefusion/models/llava_video/modeling_llava_video.py:322-338
    qwen2_vl     framefusion/models/qwenvl/modeling_qwen2_vl.py:118-137
    minicpmv     framefusion/models/minicpmv/modeling_minicpmv.py:91-106
    internvl     framefusion/models/internvl/modeling_internvl_chat.py:38-82
    nvila        framefusion/models/nvila/llava_arch.py:41-58, 86-88, 92-103

The builders below keep each block's arithmetic (same scalars, same row, including a row shorter
than the sequence when the feature count is not a whole number of frames) but write the row on the
device: the span search is ``ff_token_span`` / ``ff_patch_type_from_mask`` and the row is
``ff_fill_patch_type`` from a few segment descriptors (include/framefusion_hip.h).  A packer calls
``layout.prepare(model.framefusion)`` where the reference calls ``self.framefusion.prepare(...)``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch

from . import _lib
from ._lib import FFSegment, FrameFusionHipError

TEXT_TOKEN = -1

Segment = Tuple[int, int, int, int]          # begin, count, first type, period


@dataclass
class TokenLayout:
    patch_type: torch.Tensor                 # [1, row_length] int64 on the device
    patch_num: int
    image_token_start_index: int
    image_token_end_index: int
    image_token_length: int
    original_length: int
    # False when the builder knows the visual tokens are NOT `frames x patch_num` contiguous tokens
    # (text between InternVL's frames, MiniCPM-V's slice markers): FrameFusion then skips the
    # frame-major shortcut of its first merge call instead of finding out on the device
    frame_major: bool = True

    def prepare(self, framefusion, **flags):
        """What the reference's packers do last (e.g. modeling_llava_video.py:338)."""
        framefusion.prepare(self.patch_type, self.patch_num, self.image_token_start_index,
                            self.image_token_end_index, self.image_token_length, self.original_length, **flags)
        if not self.frame_major and hasattr(framefusion, "_layout_hint"):
            framefusion._layout_hint = None
        return self


def _device_of(t) -> torch.device:
    _lib.require_gpu(t, "layout input")
    return t.device


def fill_patch_type(length: int, segments: Sequence[Segment], device) -> torch.Tensor:
    """[1, length] int64: TEXT_TOKEN everywhere except (first + offset) % period inside each segment."""
    device = torch.device(device)
    if device.type != "cuda":
        raise FrameFusionHipError("patch_type rows are built on the GPU; there is no CPU/eager fallback")
    lib = _lib.load()
    row = torch.empty((1, length), dtype=torch.int64, device=device)
    segs = (FFSegment * max(len(segments), 1))()
    for k, (begin, count, first, period) in enumerate(segments):
        segs[k] = FFSegment(int(begin), int(count), int(first), int(period))
    with torch.cuda.device(device):
        _lib.check(lib.ff_fill_patch_type(row.data_ptr(), length, segs, len(segments), _lib.stream_ptr()),
                   "ff_fill_patch_type")
    return row


def token_span(ids: torch.Tensor, token: int) -> Tuple[int, int, int]:
    """(first index, last index, count) of `token` in a 1-D / [1, n] int64 id tensor; (-1, -1, 0)
    if absent.  One launch + one 24-byte readback (the reference's torch.where, e.g. qwenvl :123-124)."""
    device = _device_of(ids)
    row = ids.reshape(-1)
    if row.dtype != torch.int64:
        row = row.to(torch.int64)
    row = row.contiguous()
    span = torch.empty(8, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().ff_token_span(row.data_ptr(), row.numel(), int(token), span.data_ptr(),
                                             _lib.stream_ptr()), "ff_token_span")
    first, last, count = span[:3].tolist()
    return first, last, count


def _row_layout(device, start: int, patch_num: int, n_frames: int, tail: int, scalars) -> TokenLayout:
    """[TEXT]*start + range(P)*n_frames + [TEXT]*tail  (the list expression shared by the llava_video
    and qwen2_vl blocks)."""
    visual = patch_num * n_frames
    row = fill_patch_type(start + visual + max(tail, 0), [(start, visual, 0, patch_num)] if visual else [], device)
    return TokenLayout(row, *scalars)


def llava_video_layout(input_ids: torch.Tensor, image_token_index: int, image_token_length: int,
                       num_patches_per_side: int, mm_spatial_pool_mode: str) -> TokenLayout:
    """modeling_llava_video.py:322-338.  `input_ids` ([1, n] or [n]) holds ONE placeholder that the
    packer expands to `image_token_length` features."""
    assert input_ids.ndim == 1 or input_ids.shape[0] == 1                     # batch_size == 1, :328
    if mm_spatial_pool_mode == "bilinear":                                    # :322-325
        patch_size = math.ceil(num_patches_per_side / 2)
    else:
        patch_size = num_patches_per_side // 2
    patch_num = patch_size * (patch_size + 1)                                 # :326
    first, last, count = token_span(input_ids, image_token_index)
    assert count == 1                                                         # num_images == 1, :329
    n_frames = image_token_length // patch_num                                # :331
    start = first                                                             # :332
    end = start + image_token_length - 1                                      # :333
    original_length = input_ids.numel() + image_token_length - 1              # :334
    return _row_layout(input_ids.device, start, patch_num, n_frames, original_length - end - 1,
                       (patch_num, start, end, image_token_length, original_length))


def qwen2_vl_layout(input_ids: torch.Tensor, video_token_id: int, video_grid_thw, spatial_merge_size: int,
                    n_video_features: int = None, seq_len: int = None) -> TokenLayout:
    """qwenvl/modeling_qwen2_vl.py:118-137.  `input_ids` is already expanded (one video_token_id
    per feature); `video_grid_thw` is the first video's (t, h, w) as ints or a tensor row."""
    if isinstance(video_grid_thw, torch.Tensor):
        video_grid_thw = video_grid_thw.reshape(-1, 3)[0].tolist()            # video_grid_thw[0, :], :119
    _, grid_h, grid_w = (int(v) for v in video_grid_thw)
    patch_num = int((grid_h * grid_w) / (spatial_merge_size * spatial_merge_size))   # :119-120
    first, last, count = token_span(input_ids, video_token_id)
    if count == 0:
        raise FrameFusionHipError("no video tokens in input_ids")
    image_token_length = count if n_video_features is None else int(n_video_features)   # :121 (== count, :98-103)
    original_length = input_ids.numel() if seq_len is None else int(seq_len)  # :122
    start, end = first, last                                                  # :123-124
    n_frames = image_token_length // patch_num                                # :125
    return _row_layout(input_ids.device, start, patch_num, n_frames, original_length - end - 1,
                       (patch_num, start, end, image_token_length, original_length))


def minicpmv_layout(image_bound, num_frames: int, seq_len: int, device) -> TokenLayout:
    """modeling_minicpmv.py:91-106.  `image_bound` rows are (first, last + 1) of each image slice
    (a small host list or tensor)."""
    bound: List[List[int]] = image_bound.tolist() if isinstance(image_bound, torch.Tensor) else [list(b) for b in image_bound]
    patch_per_frame = len(bound) // num_frames                                # :96
    token_per_frame = bound[patch_per_frame][0] - bound[0][0]                 # :97
    lo, count = bound[0][0], bound[-1][1] - bound[0][0] + 2                   # :98
    if lo < 0 or lo + count > seq_len or token_per_frame < 1:
        raise ValueError(f"image_bound {lo}..{lo + count} does not fit a sequence of {seq_len} tokens")
    row = fill_patch_type(seq_len, [(lo, count, 0, token_per_frame)], device)
    start, end = lo, lo + count - 1                                           # :101-102 (argmax of type >= 0)
    # the modulo typing makes every token of the span visual, but the span is two tokens longer than
    # whole frames (:98), so it is frame-major only by accident
    return TokenLayout(row, token_per_frame, start, end, end - start + 1, seq_len,
                       frame_major=(count % token_per_frame == 0))                # :100, :103-104


def internvl_layout(selected: torch.Tensor, n_frames: int, patch_num: int) -> TokenLayout:
    """modeling_internvl_chat.py:38-82.  `selected` ([N] bool/uint8 on the device) marks the image
    context tokens: `n_frames` runs of `patch_num`, separated by the "Frame-k:" text.  The reference
    reads the gap lengths off the runs of False between the first and the last one (:67), i.e. it
    needs text on both sides of the video when there is more than one frame."""
    device = _device_of(selected)
    mask = selected.reshape(-1)
    mask = (mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)).contiguous()
    N = mask.numel()
    row = torch.empty((1, N), dtype=torch.int64, device=device)
    span = torch.empty(8, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().ff_patch_type_from_mask(mask.data_ptr(), N, int(patch_num), row.data_ptr(),
                                                       span.data_ptr(), _lib.stream_ptr()),
                   "ff_patch_type_from_mask")
    first, last, count, runs, bad = span[:5].tolist()
    if count == 0:
        raise IndexError("no image context tokens selected")                  # torch.where(selected)[0][0], :59
    if runs != n_frames or bad:
        raise ValueError(f"selected has {runs} runs ({bad} not of length {patch_num}); expected {n_frames} frames "
                         f"of {patch_num} tokens")
    if n_frames > 1 and (first == 0 or last == N - 1):
        raise IndexError("the reference's gap list (count_consecutive_false(selected)[1:-1]) needs text before "
                         "and after the frames")                              # :67-71
    return TokenLayout(row, int(patch_num), first, last, last - first + 1, N,
                       frame_major=(last - first + 1 == n_frames * int(patch_num)))   # no text between the frames


def nvila_layout(chunk_lengths: Sequence[int], n_media_features: int, media_frames: int, pool_sizes: int,
                 device, kind: str = "video") -> TokenLayout:
    """nvila/llava_arch.py:41-58, 86-88, 92-103.  `chunk_lengths`: lengths of the fused pieces
    (text, media, ..., text); types run from the end of the first piece to the start of the last."""
    length = int(sum(chunk_lengths))
    if kind == "video":
        num_frames = media_frames / pool_sizes                                # :48-49
        patch_num = n_media_features / num_frames                             # :52 (a float in the reference)
    else:
        patch_num, num_frames = 1, n_media_features                           # :58-59
    period, reps = math.ceil(patch_num), int(num_frames)                      # torch.arange(patch_num).repeat(int(num_frames)), :86
    lo, hi = int(chunk_lengths[0]), int(sum(chunk_lengths[:-1]))              # :88
    if hi - lo != period * reps:
        raise RuntimeError(f"{period * reps} types do not fit the slice {lo}:{hi}")   # shape mismatch of the slice assignment
    row = fill_patch_type(length, [(lo, hi - lo, 0, period)] if hi > lo else [], device)
    start, end = lo, hi - 1                                                   # :92-93
    return TokenLayout(row, patch_num, start, end, end - start + 1, length)   # :94-95
