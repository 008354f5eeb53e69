"""Token layouts: the patch_type row + prepare scalars of the
five packers.  tests/golden/layout.npz holds what the reference's own FRAMEFUSION blocks produced
(oracle/make_golden_layout.py).  CPU: the oracle restatement against those vectors.  GPU: the HIP
builders (ff_token_span / ff_fill_patch_type / ff_patch_type_from_mask through framefusion_amd.layout)
against the vectors and against the oracle on larger seeded layouts - integers, so bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import layout_oracle as lay
from tests.conftest import Golden

DEV = "cuda:0"
IMAGE_TOKEN_INDEX = -200
VIDEO_TOKEN_ID = 151656


def golden_cases(prefix):
    g = Golden("layout")
    names = sorted({"/".join(k.split("/")[:2]) for k in g.z.files if k.startswith(prefix + "/")})
    assert names
    return g, names


def expected(g, name):
    return g[f"{name}/patch_type"].tolist(), tuple(int(v) for v in g[f"{name}/scalars"])


def oracle_of(g, name):
    fam = name.split("/")[0]
    i = lambda key: g[f"{name}/in/{key}"]
    if fam == "llava_video":
        return lay.llava_video(i("ids").tolist(), IMAGE_TOKEN_INDEX, int(i("itl")), int(i("side")),
                               "bilinear" if int(i("bilinear")) else "average")
    if fam == "qwen2_vl":
        t, h, w = i("grid").tolist()
        ids = i("ids").tolist()
        return lay.qwen2_vl(ids, VIDEO_TOKEN_ID, h, w, int(i("merge")), int(i("n")), len(ids))
    if fam == "minicpmv":
        return lay.minicpmv(i("bounds").tolist(), int(i("frames")), int(i("length")))
    if fam == "internvl":
        return lay.internvl([bool(v) for v in i("selected")], int(i("frames")), int(i("patch_num")))
    if fam == "nvila":
        return lay.nvila(i("chunks").tolist(), int(i("n_feat")), int(i("media_frames")), int(i("pool")),
                         "video" if int(i("video")) else "image")
    raise AssertionError(fam)


def hip_of(g, name):
    from framefusion_amd import layout as L
    fam = name.split("/")[0]
    i = lambda key: g[f"{name}/in/{key}"]
    t = lambda key, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(i(key))).to(dt).to(DEV)
    if fam == "llava_video":
        return L.llava_video_layout(t("ids")[None], IMAGE_TOKEN_INDEX, int(i("itl")), int(i("side")),
                                    "bilinear" if int(i("bilinear")) else "average")
    if fam == "qwen2_vl":
        return L.qwen2_vl_layout(t("ids")[None], VIDEO_TOKEN_ID, t("grid")[None], int(i("merge")))
    if fam == "minicpmv":
        return L.minicpmv_layout(t("bounds"), int(i("frames")), int(i("length")), DEV)
    if fam == "internvl":
        return L.internvl_layout(t("selected").bool(), int(i("frames")), int(i("patch_num")))
    if fam == "nvila":
        return L.nvila_layout(i("chunks").tolist(), int(i("n_feat")), int(i("media_frames")), int(i("pool")), DEV,
                              "video" if int(i("video")) else "image")
    raise AssertionError(fam)


FAMILIES = ["llava_video", "qwen2_vl", "minicpmv", "internvl", "nvila"]


@pytest.mark.parametrize("family", FAMILIES)
def test_oracle_layout_matches_reference_vectors(family):
    g, names = golden_cases(family)
    for name in names:
        row, scalars = expected(g, name)
        got = oracle_of(g, name)
        assert [int(v) for v in got[0]] == row, name
        assert tuple(int(v) for v in got[1:]) == scalars, name


def as_tuple(layout):
    return (layout.patch_type[0].tolist(), int(layout.patch_num), int(layout.image_token_start_index),
            int(layout.image_token_end_index), int(layout.image_token_length), int(layout.original_length))


@pytest.mark.gpu
@pytest.mark.parametrize("family", FAMILIES)
def test_hip_layout_matches_reference_vectors(family):
    g, names = golden_cases(family)
    for name in names:
        row, scalars = expected(g, name)
        got = as_tuple(hip_of(g, name))
        assert got[0] == row, name
        assert got[1:] == scalars, name
        assert hip_of(g, name).patch_type.dtype == torch.int64


@pytest.mark.gpu
def test_hip_layout_full_size_vs_oracle():
    from framefusion_amd import layout as L
    # LLaVA-Video-7B: 64 frames of 14x15 behind 14 prompt tokens (SURVEY.md §8 C2 real-model shape)
    ids = list(range(14)) + [IMAGE_TOKEN_INDEX] + list(range(20))
    itl = 64 * 210
    want = lay.llava_video(ids, IMAGE_TOKEN_INDEX, itl, 27, "bilinear")
    got = as_tuple(L.llava_video_layout(torch.tensor([ids], device=DEV), IMAGE_TOKEN_INDEX, itl, 27, "bilinear"))
    assert got[0] == want[0] and got[1:] == tuple(want[1:])
    # a feature count that is not a whole number of frames: the reference's row comes out short
    want = lay.llava_video(ids, IMAGE_TOKEN_INDEX, itl + 1, 27, "bilinear")
    got = as_tuple(L.llava_video_layout(torch.tensor([ids], device=DEV), IMAGE_TOKEN_INDEX, itl + 1, 27, "bilinear"))
    assert len(want[0]) == want[5] - 1 and got[0] == want[0] and got[1:] == tuple(want[1:])
    # Qwen2-VL, 64 temporal grids of 26x30 / 4 = 195 tokens (C3)
    ids = [7] * 15 + [VIDEO_TOKEN_ID] * (64 * 195) + [9] * 12
    want = lay.qwen2_vl(ids, VIDEO_TOKEN_ID, 26, 30, 2, 64 * 195, len(ids))
    got = as_tuple(L.qwen2_vl_layout(torch.tensor(ids, device=DEV), VIDEO_TOKEN_ID, (64, 26, 30), 2))
    assert got[0] == want[0] and got[1:] == tuple(want[1:])
    # InternVL: 48 frames of 256 tokens, "Frame-k: " gaps of varying length, > 128 segments' worth of runs
    sel = [False] * 30
    for f in range(48):
        sel += [True] * 256 + [False] * (6 + (f % 3))
    sel += [False] * 11
    want = lay.internvl(sel, 48, 256)
    got = as_tuple(L.internvl_layout(torch.tensor(sel, device=DEV), 48, 256))
    assert got[0] == want[0] and got[1:] == tuple(want[1:])


@pytest.mark.gpu
def test_hip_fill_many_segments_and_span_edges():
    from framefusion_amd import layout as L
    rng = np.random.default_rng(5)
    length, segs, pos = 40000, [], 3
    while pos < length - 200 and len(segs) < 300:          # > 128 segments: several launches
        count = int(rng.integers(1, 90))
        period = int(rng.integers(1, 40))
        segs.append((pos, count, int(rng.integers(0, period)), period))
        pos += count + int(rng.integers(0, 50))
    want = np.full(length, -1, dtype=np.int64)
    for b, c, f, p in segs:
        want[b:b + c] = (f + np.arange(c)) % p
    got = L.fill_patch_type(length, segs, DEV)[0].cpu().numpy()
    assert np.array_equal(got, want)
    assert L.fill_patch_type(5, [], DEV)[0].tolist() == [-1] * 5
    assert L.fill_patch_type(0, [], DEV).shape == (1, 0)
    ids = torch.tensor([4, 9, 9, 1, 9, 2], device=DEV)
    assert L.token_span(ids, 9) == (1, 4, 3)
    assert L.token_span(ids, 4) == (0, 0, 1)
    assert L.token_span(ids, 77) == (-1, -1, 0)
    assert L.token_span(ids.to(torch.int32), 2) == (5, 5, 1)
    big = torch.zeros(100_003, dtype=torch.int64, device=DEV)
    big[[17, 99_999, 100_002]] = 5
    assert L.token_span(big, 5) == (17, 100_002, 3)


@pytest.mark.gpu
def test_hip_layout_rejects_what_the_reference_cannot_build():
    from framefusion_amd import layout as L
    from framefusion_amd._lib import FrameFusionHipError
    ok = [False] * 3 + [True] * 4 + [False] * 2 + [True] * 4 + [False] * 2
    assert as_tuple(L.internvl_layout(torch.tensor(ok, device=DEV), 2, 4))[0] == lay.internvl(ok, 2, 4)[0]
    with pytest.raises(ValueError):                        # a run that is not one frame long
        L.internvl_layout(torch.tensor([False, True, True, True, False, True, False], device=DEV), 2, 2)
    with pytest.raises(ValueError):                        # frame count does not match
        L.internvl_layout(torch.tensor(ok, device=DEV), 3, 4)
    with pytest.raises(IndexError):                        # no text in front: the reference's gap list is short
        L.internvl_layout(torch.tensor(ok[3:], device=DEV), 2, 4)
    with pytest.raises(IndexError):
        lay.internvl(ok[3:], 2, 4)
    with pytest.raises(IndexError):
        L.internvl_layout(torch.zeros(9, dtype=torch.bool, device=DEV), 1, 4)
    one = [True] * 4 + [False]                              # a single frame needs no gaps
    assert as_tuple(L.internvl_layout(torch.tensor(one, device=DEV), 1, 4))[0] == lay.internvl(one, 1, 4)[0]
    with pytest.raises(FrameFusionHipError):
        L.qwen2_vl_layout(torch.tensor([1, 2, 3], device=DEV), VIDEO_TOKEN_ID, (1, 4, 4), 2)
    with pytest.raises(FrameFusionHipError):               # CPU tensors: no eager fallback
        L.token_span(torch.tensor([1, 2, 3]), 2)
    with pytest.raises(FrameFusionHipError):
        L.fill_patch_type(4, [], "cpu")
    with pytest.raises(ValueError):
        L.minicpmv_layout([(2, 6), (8, 12)], 2, 12, DEV)    # bound + 2 runs past the sequence
    with pytest.raises(RuntimeError):
        L.nvila_layout([3, 10, 2], 12, 4, 1, DEV)           # 12 types into a 10-token slice


@pytest.mark.gpu
def test_layouts_drive_a_merge_call():
    """A layout built on the device feeds FrameFusion.prepare and the first merge call: frame-major
    layouts take the closed-form order inside the similarity kernel, InternVL's frames with text in
    between are announced as such (no wasted hinted attempt) - both must give the oracle's result."""
    import framefusion_amd as ffa
    from framefusion_amd import layout as L
    from framefusion_amd.synth import video_tokens
    from oracle import ff_oracle as orc
    from tests.conftest import same_bits
    F, P, d, gap = 6, 16, 64, 5
    # InternVL-like: text, then F frames of P tokens separated by `gap` text tokens, then text
    sel = [False] * 3
    for f in range(F):
        sel += [True] * P + ([False] * gap if f + 1 < F else [])
    sel += [False] * 4
    N = len(sel)
    lay_i = L.internvl_layout(torch.tensor(sel, device=DEV), F, P)
    assert not lay_i.frame_major
    vis, _ = video_tokens(F, P, d, p_change=0.4, sigma=0.3, seed=3, grid=0.125)
    hidden = torch.zeros(1, N, d, dtype=vis.dtype)
    text, _ = video_tokens(1, N, d, p_change=1.0, sigma=0.3, seed=4, grid=0.125)
    hidden[0] = text[0]
    hidden[0, torch.tensor(sel)] = vis[0]
    for lay_x, hint in ((lay_i, None),
                        (L.qwen2_vl_layout(torch.tensor([7] * 3 + [VIDEO_TOKEN_ID] * (F * P) + [9] * 4, device=DEV),
                                           VIDEO_TOKEN_ID, (F, 8, 8), 2), (3, F))):
        f = ffa.FrameFusion(0.3, 0.6, 0.1)
        lay_x.prepare(f)
        assert f._layout_hint == hint
        n = lay_x.original_length
        h = hidden[:, :n].contiguous()
        o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        o.prepare(lay_x.patch_type.cpu(), lay_x.patch_num, lay_x.image_token_start_index, lay_x.image_token_end_index,
                  lay_x.image_token_length, lay_x.original_length)
        hg, pg, _ = f(h.to(DEV), torch.arange(n, device=DEV)[None], None)
        ho, po, _ = o.forward(h, torch.arange(n)[None], None)
        assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
        assert int(f.last_call["scratch"].stats[11]) == 0          # no layout mismatch was raised on the device
