"""GPU: parity at BASELINE.json's full sizes (configs C2/C3/C5) against the CPU oracle,
plus size-independent properties of the reduction."""
import numpy as np
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from oracle import ff_oracle as orc
from tests import harness
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_both(F, P, d, p_change, dtype=torch.bfloat16, pre=14, post=20, grid=0.125, sigma_hi=None,
             params=(0.3, 0.6, 0.1), pos="qwen2", seed=1234):
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, seed=seed, pre=pre, post=post, dtype=dtype,
                         grid=grid, sigma_hi=sigma_hi)
    L = h.shape[1]
    mk = (lambda: rotary_tables(L, 128, dtype, mrope=(pos == "mrope")))
    o = orc.OracleFrameFusion(*params)
    o.prepare(pt.clone(), P, pre, pre + F * P, F * P, L)
    ho, peo, _ = o.forward(h, mk(), None)
    f = ffa.FrameFusion(*params)
    f.prepare(pt.to(DEV), P, pre, pre + F * P, F * P, L)
    pe_in = [t.to(DEV) for t in mk()]
    hd = h.to(DEV)
    hg, peg, _ = f(hd, pe_in, None)
    return h, pt, o, ho, peo, f, hg, peg


@pytest.mark.parametrize("p_change,branch", [(0.2, "topk"), (0.5, "thr"), (0.95, "low")])
def test_c2_metric_shape_exact(p_change, branch):
    """[1, 64x576 + 34, 4096] bf16 on the dyadic grid: every output bit-identical to the oracle."""
    h, pt, o, ho, peo, f, hg, peg = run_both(64, 576, 4096, p_change, sigma_hi=1.6 if branch == "thr" else None)
    L = h.shape[1]
    keep_g = torch.nonzero(f.last_plan()["keep"].bool()).reshape(-1).cpu()
    assert torch.equal(keep_g, o.last_keep)                              # kept-token indices: bit-exact
    assert (f.finish_merging, f.finish_pruning, f.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list)
    assert {"topk": f.finish_pruning, "thr": not f.finish_merging, "low": f.finish_merging and not f.finish_pruning}[branch]
    assert same_bits(hg.cpu(), ho)                                        # merged hidden_states: bit-exact
    for a, b in zip(peg, peo):
        assert same_bits(a.cpu().contiguous(), b.contiguous())
    assert torch.equal(f.patch_type.cpu(), o.patch_type)
    # properties
    assert hg.shape[1] == keep_g.numel() == L - int(f.last_call["L_in"] - f.last_call["L_out"])
    text = torch.nonzero(pt[0] == -1).reshape(-1)
    assert bool(torch.isin(text, keep_g).all())                           # text tokens always survive
    if branch == "topk":
        assert L - hg.shape[1] == int(orc.budget([], 0.3) * 64 * 576)     # exactly k tokens folded


def test_c2_gaussian_within_tolerance():
    """Unquantised gaussian activations: fp32 reduction order may flip a bf16 rounding on ~1e-4 of the
    pairs.  Identical similarities => identical everything; otherwise the
    differences must be explained by those flips."""
    h, pt, o, ho, peo, f, hg, peg = run_both(64, 576, 4096, 0.5, grid=None, sigma_hi=1.6, pre=0, post=0)
    plan = f.last_plan()
    sim_g, sim_o = plan["sim"].cpu().float(), o.last_sim[0].float()
    flips = int((sim_g != sim_o).sum())
    assert flips <= 1e-3 * sim_o.numel()
    # a flipped rounding of one row norm moves T(na*nb) and then T(dot/den) by up to ~3 bf16 ulps,
    # on both pairs the row takes part in (measured: 5 flips / 36864, max 3 ulps)
    assert torch.allclose(sim_g, sim_o, rtol=2 ** -6, atol=1e-6)
    keep_g = torch.nonzero(plan["keep"].bool()).reshape(-1).cpu()
    if flips == 0:
        assert torch.equal(keep_g, o.last_keep) and same_bits(hg.cpu(), ho)
    else:
        sym = np.setxor1d(keep_g.numpy(), o.last_keep.numpy())
        assert sym.size <= 2 * flips
    common, ig, io = np.intersect1d(keep_g.numpy(), o.last_keep.numpy(), return_indices=True)
    a, b = hg[0].cpu().float()[ig], ho[0].float()[io]
    rel = ((a - b).abs() / b.abs().clamp_min(1e-3))
    assert float((rel > 1e-3).float().mean()) <= 1e-3                    # bf16 hidden within 1e-3 rel


@pytest.mark.parametrize("P", [180, 195])
@pytest.mark.parametrize("thr", [0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9])
def test_c3_qwen2vl_shape_threshold_sweep(thr, P):
    """Qwen2-VL-7B shape (BASELINE.json configs[2]): 128 frames = 64 temporal grids x {180, 195} tokens, d=3584,
    M-RoPE containers, num=4 importance, the whole similarity_lower_bound sweep of SURVEY.md §8d."""
    F, d, pre, post = 64, 3584, 15, 12
    h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.8, seed=77, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, thr, 0.1), h.clone(), pt.clone(), P,
                                  rotary_tables(L, 128, mrope=True), None, layers=3, heads=28, num=4)
    got, _ = harness.run_cascade(ffa.FrameFusion(0.3, thr, 0.1), h.to(DEV), pt.to(DEV), P,
                                 [t.to(DEV) for t in rotary_tables(L, 128, mrope=True)], None, layers=3, heads=28, num=4)
    for a, b in zip(got, want):
        assert (a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
               (b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"]), a["tag"]
        assert same_bits(a["hidden"].cpu(), b["hidden"]), a["tag"]
        for x, y in zip(a["pos"], b["pos"]):
            assert same_bits(x.cpu().contiguous(), y.contiguous())


def test_c5_72b_shape_fused_importance_and_prune():
    """LLaVA-Video-72B shape (d=8192, H=64, H_kv=8, dh=128): merge, then importance from q/k inside the
    attention hook, then the prune call."""
    F, P, d, pre, post = 64, 576, 8192, 14, 20          # BASELINE.json configs[4]: 64 frames
    H, Hk, dh = 64, 8, 128
    h, pt = video_tokens(F, P, d, p_change=0.95, sigma=0.3, seed=5, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), P, pre, pre + F * P, F * P, L)
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    f.prepare(pt.to(DEV), P, pre, pre + F * P, F * P, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    hg, pg, _ = f(h.to(DEV), torch.arange(L, device=DEV)[None], None)
    assert same_bits(hg.cpu(), ho) and torch.equal(pg.cpu(), po)
    assert o.finish_merging and not o.finish_pruning and f.finish_merging and not f.finish_pruning
    S = ho.shape[1]
    g = torch.Generator().manual_seed(3)
    q = harness.snap(torch.randn(1, H, 8, dh, generator=g), torch.bfloat16)
    k = harness.snap(torch.randn(1, Hk, S, dh, generator=g), torch.bfloat16)
    w_o = orc.last_query_attention(q, k, num=1, is_causal=True, enable_gqa=True)
    w_g = ffa.scaled_dot_product_attention(q.to(DEV), k.to(DEV), None, num=1, is_causal=True, enable_gqa=True)
    assert torch.allclose(w_g.cpu().float(), w_o.float(), rtol=2 ** -7, atol=1e-30)          # never more than one ulp
    assert float((w_g.cpu().float() != w_o.float()).float().mean()) <= 1e-3                  # measured: ~2e-5
    # feed both sides the SAME weights so the prune itself is compared exactly
    h2 = harness.layer_stub(ho, 0)
    ho2, po2, _ = o.forward(h2, po, None, w_o)
    hg2, pg2, _ = f(h2.to(DEV), pg, None, w_o.to(DEV))
    assert o.finish_pruning and f.finish_pruning
    assert torch.equal(pg2.cpu(), po2) and same_bits(hg2.cpu(), ho2)
    # and the fused [1, 1, 1, S] importance is accepted in place of the weights (a11)
    f2 = ffa.FrameFusion(0.3, 0.6, 0.1)
    f2.prepare(f.patch_type, P, pre, pre + F * P, F * P, L, finish_merging=True, sparsity_list=list(o.sparsity_list[:1]))
    imp = ffa.last_query_importance(q.to(DEV), k.to(DEV), num=1, is_causal=True, framefusion=f2)   # + the prune's select tables
    hg3, pg3, _ = f2(h2.to(DEV), pg, None, imp)
    assert hg3.shape == hg2.shape
    # ranking from HIP-computed importances: the kept set may differ from the oracle's only at positions whose
    # importance differs by an ulp or sits in the tie class of the cut
    imp_g, imp_o = imp.reshape(-1).cpu().float(), torch.mean(w_o, dim=(1, 2))[0].float()
    differ = imp_g != imp_o
    assert float(differ.float().mean()) <= 1e-3
    pos_in = pg[0].cpu()                                   # original positions of the S tokens that entered the prune
    in_g = torch.isin(pos_in, pg3[0].cpu()); in_o = torch.isin(pos_in, po2[0])
    k_cut = int(in_o.sum()) - (S - (F * P - (L - S)))      # visual tokens kept
    vis = slice(pre, pre + F * P - (L - S))
    kth = torch.sort(imp_o[vis], descending=True).values[k_cut - 1]
    unsure = differ | (imp_o == kth) | (imp_g == torch.sort(imp_g[vis], descending=True).values[k_cut - 1])
    assert torch.equal(in_g[~unsure], in_o[~unsure])
    assert int(in_g.sum()) == int(in_o.sum())


def test_128_frames_cascade_and_large_ragged_order():
    """128 frames x 576 tokens (73 762 positions, the longest configuration of BASELINE.json): a
    two-merge cascade against the oracle, and the general (counting-sort) order kernel on the ragged
    patch_type that the first merge leaves behind."""
    from framefusion_amd import _lib
    F, P, d, pre, post = 128, 576, 1024, 14, 20
    h, pt = video_tokens(F, P, d, p_change=0.55, sigma=0.3, sigma_hi=1.6, seed=9, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    o = orc.OracleFrameFusion(0.3, 0.6, 0.01)
    o.prepare(pt.clone(), P, pre, pre + F * P, F * P, L)
    f = ffa.FrameFusion(0.3, 0.6, 0.01)
    f.prepare(pt.to(DEV), P, pre, pre + F * P, F * P, L)
    ho, po = h, torch.arange(L)[None]
    hg, pg = h.to(DEV), torch.arange(L, device=DEV)[None]
    for call in range(2):
        ho, po, _ = o.forward(ho, po, None)
        hg, pg, _ = f(hg, pg, None)
        assert torch.equal(pg.cpu(), po), call
        assert same_bits(hg.cpu(), ho), call
        assert (f.finish_merging, f.finish_pruning, f.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list)
        if call == 0:
            assert not f.finish_merging                      # the second call runs on the maintained order
            # the general order kernel on this ragged layout == the maintained order == the oracle's
            lib = _lib.load()
            n = hg.shape[1]
            order = torch.empty(n, dtype=torch.int32, device=DEV)
            stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=DEV)
            wsb = int(lib.ff_workspace_bytes(n, P))
            ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
            inv = torch.empty(n, dtype=torch.int32, device=DEV)
            _lib.check(lib.ff_build_order(f.patch_type.data_ptr(), n, P, order.data_ptr(), inv.data_ptr(),
                                          stats.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()), "ff_build_order")
            assert torch.equal(inv[order.long()].cpu(), torch.arange(n, dtype=torch.int32))    # inv is the inverse permutation
            want, _ = orc.by_patch_order(o.patch_type, P)
            nv = want.numel()
            assert int(stats[_lib.STAT_NV]) == nv
            assert torch.equal(order[:nv].cpu().long(), want)
            assert torch.equal(f.last_call["scratch"].order[:nv].cpu().long(), want)
        ho, hg = harness.layer_stub(ho, call), harness.layer_stub(hg, call)


@pytest.mark.parametrize("dt", ["fp16", "fp32"])
@pytest.mark.parametrize("p_change", [0.2, 0.6])
def test_other_dtypes_cascade_exact(dt, p_change):
    """fp16 / fp32 activations (the dtype only changes where the recipe rounds): whole cascades at a
    medium size, bit-exact on grid data, through merge (both branches) and prune."""
    from tests.conftest import DT
    dtype = DT[dt]
    F, P, d, pre, post = 32, 100, 1024, 9, 11
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.6, seed=21, pre=pre, post=post,
                         dtype=dtype, grid=0.125)
    L = h.shape[1]
    want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P,
                                  rotary_tables(L, 64, dtype), None, layers=3, heads=8, num=1)
    got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), h.to(DEV), pt.to(DEV), P,
                                 [t.to(DEV) for t in rotary_tables(L, 64, dtype)], None, layers=3, heads=8, num=1)
    assert got[-1]["finish_merging"] and got[-1]["finish_pruning"]
    for a, b in zip(got, want):
        assert (a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
               (b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"]), a["tag"]
        assert same_bits(a["hidden"].cpu(), b["hidden"]), a["tag"]


@pytest.mark.parametrize("p_change,pre", [(0.3, 9), (0.6, 0)])
def test_very_long_sequence_exact(p_change, pre):
    """~300 k tokens (512 frames x 576 patches, narrow rows): 72 slices of the select / flags / scan
    kernels, by-patch runs up to hundreds of frames long, two merge calls (the second on the maintained
    order).  Bit-exact against the oracle on the dyadic grid."""
    F, P, d = 512, 576, 64
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, seed=5, pre=pre, post=13, dtype=torch.bfloat16, grid=0.125)
    L = h.shape[1]
    o = orc.OracleFrameFusion(0.3, 0.6, 0.02)
    o.prepare(pt.clone(), P, pre, pre + F * P - 1, F * P, L)
    f = ffa.FrameFusion(0.3, 0.6, 0.02)
    f.prepare(pt.to(DEV), P, pre, pre + F * P - 1, F * P, L)
    ho, po = h, torch.arange(L)[None]
    hg, pg = h.to(DEV), torch.arange(L, device=DEV)[None]
    calls = 0
    for layer in range(3):
        if o.finish_merging:
            break
        ho, po, _ = o.forward(ho, po, None)
        hg, pg, _ = f(hg, pg, None)
        calls += 1
        assert torch.equal(pg.cpu(), po), layer
        assert same_bits(hg.cpu(), ho), layer
        assert (f.finish_merging, f.finish_pruning, f.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list)
        ho, hg = harness.layer_stub(ho, layer), harness.layer_stub(hg, layer)
    assert calls >= 1 and ho.shape[1] < L


@pytest.mark.parametrize("frames", [370, 450, 570, 610, 810, 850, 930])
def test_merge_workgroup_slot_counts(frames):
    """The merge kernel sizes its workgroups per launch (a prime number of by-patch slots, ff_merge.hip
    merge_slots): with one column group (d = 64) and 100 patches these lengths land on 19, 23, 29, 31, 41, 43
    and 47 slots per workgroup - the windows of 64 slots are refilled at different points of the runs.
    One merge call each, bit-exact against the oracle on the dyadic grid."""
    P, d = 100, 64
    h, pt = video_tokens(frames, P, d, p_change=0.25, sigma=0.3, seed=frames, pre=3, post=5, dtype=torch.bfloat16, grid=0.125)
    L = h.shape[1]
    o = orc.OracleFrameFusion(0.3, 0.6, 0.02)
    o.prepare(pt.clone(), P, 3, 3 + frames * P - 1, frames * P, L)
    f = ffa.FrameFusion(0.3, 0.6, 0.02)
    f.prepare(pt.to(DEV), P, 3, 3 + frames * P - 1, frames * P, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    hg, pg, _ = f(h.to(DEV), torch.arange(L, device=DEV)[None], None)
    assert ho.shape[1] < L
    assert torch.equal(pg.cpu(), po)
    assert same_bits(hg.cpu(), ho)


@pytest.mark.parametrize("F,P,pre,post", [(128, 576, 14, 20),          # Qwen2-VL example length (BASELINE configs[2]): 73 762
                                          (256, 576, 14, 20),          # 147 490: the largest the fast plan kernel is sized for
                                          (128, 512, 0, 0),            # 65 536 exactly: last length of the 16-slice instantiation
                                          (128, 512, 1, 0),            # 65 537: first of the 40-slice one
                                          (320, 512, 0, 0),            # 163 840: its last
                                          (320, 512, 0, 1)])           # 163 841: first of the general plan kernel
@pytest.mark.parametrize("p_change", [0.2, 0.55])
def test_plan_kernel_size_classes(F, P, pre, post, p_change):
    """The plan kernel comes in three size classes (ff_plan.hip: k_plan_fast with 16 or 40 slices of level-1 rows in
    LDS, k_plan beyond 163 840 tokens).  A whole cascade (top-k regime, and threshold regime + prune) on either
    side of every boundary and at 128 / 256 frames x 576 tokens, bit-exact against the oracle on the dyadic grid."""
    d = 128
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.6, seed=F + pre, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    assert L == pre + F * P + post
    want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P,
                                  torch.arange(L)[None], None, layers=3, heads=2, num=1, start=pre, n_visual=F * P)
    got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), h.to(DEV), pt.to(DEV), P,
                                 torch.arange(L, device=DEV)[None], None, layers=3, heads=2, num=1, start=pre, n_visual=F * P)
    assert len(got) == len(want) and got[-1]["finish_merging"]
    for a, b in zip(got, want):
        assert (a["tag"], a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
               (b["tag"], b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"]), a["tag"]
        assert torch.equal(a["pos"].cpu(), b["pos"]), a["tag"]
        assert same_bits(a["hidden"].cpu(), b["hidden"]), a["tag"]
