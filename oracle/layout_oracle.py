"""CPU oracle for the token-layout builders - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (framefusion_amd/) never does.

Plain-Python restatement of the blocks with which the reference's multimodal packers build the
``patch_type`` row and the five scalars they hand to ``FrameFusion.prepare`` (SURVEY.md §8 row P
and §8(f) rank 2).  Every function returns

    (patch_type: list[int], patch_num, start, end, image_token_length, original_length)

with python ints.  The list is built exactly the way the reference builds it (list arithmetic or
slice assignment), so a layout the reference would get wrong (a feature count that is not a
multiple of the frame size, no text in front of the first frame ...) comes out wrong in the same way.

Pinning: oracle/make_golden_layout.py executes the reference's own statements (the marked
FRAMEFUSION blocks, cut out of the adapter sources by line range at generation time, with stand-in
inputs) and asserts equality with these functions; the results are committed as
tests/golden/layout.npz.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

TEXT_TOKEN = -1          # framefusion/main.py:5

Layout = Tuple[List[int], int, int, int, int, int]


def llava_video(input_ids: Sequence[int], image_token_index: int, image_token_length: int,
                num_patches_per_side: int, mm_spatial_pool_mode: str) -> Layout:
    """framefusion/models/llava_video/modeling_llava_video.py:322-338.  `input_ids` holds ONE
    placeholder (IMAGE_TOKEN_INDEX) that the packer expands to `image_token_length` features."""
    if mm_spatial_pool_mode == "bilinear":                                    # :322-325
        patch_size = math.ceil(num_patches_per_side / 2)
    else:
        patch_size = num_patches_per_side // 2
    patch_num = patch_size * (patch_size + 1)                                 # :326 (one newline token per row)
    n_frames = image_token_length // patch_num                                # :331
    hits = [i for i, t in enumerate(input_ids) if t == image_token_index]     # :332
    assert len(hits) == 1                                                     # :329 (num_images == 1)
    start = hits[0]
    end = start + image_token_length - 1                                      # :333
    original_length = len(input_ids) + image_token_length - 1                 # :334
    patch_type = ([TEXT_TOKEN] * start + list(range(patch_num)) * n_frames
                  + [TEXT_TOKEN] * (original_length - end - 1))               # :335
    return patch_type, patch_num, start, end, image_token_length, original_length


def qwen2_vl(input_ids: Sequence[int], video_token_id: int, grid_h: int, grid_w: int,
             spatial_merge_size: int, n_video_features: int, seq_len: int) -> Layout:
    """framefusion/models/qwenvl/modeling_qwen2_vl.py:118-137 (prefill only, :118).  `input_ids` is
    already expanded: one video_token_id per feature."""
    patch_num = int((grid_h * grid_w) / (spatial_merge_size * spatial_merge_size))   # :119-120
    image_token_length = n_video_features                                     # :121
    original_length = seq_len                                                 # :122
    hits = [i for i, t in enumerate(input_ids) if t == video_token_id]
    start, end = hits[0], hits[-1]                                            # :123-124
    n_frames = image_token_length // patch_num                                # :125
    patch_type = ([TEXT_TOKEN] * start + list(range(patch_num)) * n_frames
                  + [TEXT_TOKEN] * (original_length - end - 1))               # :126
    return patch_type, patch_num, start, end, image_token_length, original_length


def minicpmv(image_bound: Sequence[Sequence[int]], num_frames: int, seq_len: int) -> Layout:
    """framefusion/models/minicpmv/modeling_minicpmv.py:91-106.  `image_bound[r] = (first, last+1)`
    of the r-th image slice; everything from the first slice to two tokens past the last one counts
    as visual, typed by its offset modulo the distance between frame starts."""
    patch_type = [TEXT_TOKEN] * seq_len                                       # :92
    patch_per_frame = len(image_bound) // num_frames                          # :96
    token_per_frame = image_bound[patch_per_frame][0] - image_bound[0][0]     # :97
    lo, hi = image_bound[0][0], image_bound[-1][1] + 2                        # :98
    values = [v % token_per_frame for v in range(0, image_bound[-1][1] - image_bound[0][0] + 2)]
    assert 0 <= lo and hi <= seq_len, "slice assignment of a different length raises in the reference"
    patch_type[lo:hi] = values
    patch_num = token_per_frame                                               # :100
    start = next(i for i, v in enumerate(patch_type) if v >= 0)               # :101
    end = seq_len - 1 - next(i for i, v in enumerate(reversed(patch_type)) if v >= 0)   # :102
    return patch_type, patch_num, start, end, end - start + 1, seq_len        # :103-104


def internvl(selected: Sequence[bool], n_frames: int, patch_num: int) -> Layout:
    """framefusion/models/internvl/modeling_internvl_chat.py:38-82.  `selected[i]` marks the image
    context tokens; frames are separated by text ("Frame-k: " prompts), whose lengths are read off
    the runs of False between the first and the last run (:38-56, :67)."""
    N = len(selected)
    hits = [i for i, s in enumerate(selected) if s]
    start, end = hits[0], hits[-1]                                            # :59-60
    image_token_length = end - start + 1                                      # :61
    runs, cur = [], 0                                                         # count_consecutive_false, :38-56
    false_idx = [i for i, s in enumerate(selected) if not s]
    if false_idx:
        cur = 1
        for a, b in zip(false_idx, false_idx[1:]):
            if b - a != 1:
                runs.append(cur)
                cur = 0
            cur += 1
        runs.append(cur)
    text_length_list = runs[1:-1]                                             # :67
    patch_type = [TEXT_TOKEN] * start                                         # :69
    for i in range(n_frames - 1):                                             # :70-71
        patch_type = patch_type + list(range(patch_num)) + [TEXT_TOKEN] * text_length_list[i]
    patch_type = patch_type + list(range(patch_num)) + [TEXT_TOKEN] * (N - end - 1)   # :72
    return patch_type, patch_num, start, end, image_token_length, N


def nvila(chunk_lengths: Sequence[int], n_media_features: int, media_frames: int, pool_sizes: int,
          kind: str = "video") -> Layout:
    """framefusion/models/nvila/llava_arch.py:41-58, :86-88, :92-103.  `chunk_lengths` are the
    lengths of the fused pieces `inputs_mk` (text, media, ..., text): types are written from the end
    of the first piece to the start of the last one (:88)."""
    length = sum(chunk_lengths)                                               # = text + media - placeholders, :50/:56
    patch_type = [TEXT_TOKEN] * length                                        # :51/:57
    if kind == "video":
        num_frames = media_frames / pool_sizes                                # :48-49 (python floats from here on)
        patch_num = n_media_features / num_frames                             # :52
    else:
        patch_num, num_frames = 1, n_media_features                           # :58-59
    seq = [float(v) if kind == "video" else v
           for v in list(range(math.ceil(patch_num))) * int(num_frames)]      # :86 torch.arange(float).repeat
    lo, hi = chunk_lengths[0], sum(chunk_lengths[:-1])                        # :88
    assert hi - lo == len(seq), "slice assignment of a different length raises in the reference"
    patch_type[lo:hi] = [int(v) for v in seq]
    start = next(i for i, v in enumerate(patch_type) if v >= 0)               # :92
    end = length - 1 - next(i for i, v in enumerate(reversed(patch_type)) if v >= 0)   # :93
    return patch_type, patch_num, start, end, end - start + 1, length         # :94-95
