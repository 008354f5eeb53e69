"""GPU: the HIP path (through the C ABI, via framefusion_amd) against the golden vectors of the real
reference and against the CPU oracle on the same seeded inputs.

Bars: integer/index outputs bit-exact; activations bit-exact on dyadic-grid inputs (where the fp32
sums are order-independent) and within 1e-3 relative (the north-star tolerance) elsewhere;
similarities on gaussian inputs within 1 ulp on <= 0.1 % of the pairs (fp32 reduction-order noise,
SURVEY.md Appendix B).
"""
import numpy as np
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables
from oracle import ff_oracle as orc
from tests import harness
from tests.conftest import DT, from_bits, same_bits, Golden
from tests.test_oracle_golden import make_pos, SIM_CASES, FWD_CASES, CAS_CASES, IMP_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


# ---------------------------------------------------------------------------------------------
# K0
# ---------------------------------------------------------------------------------------------
def hip_order(pt, P):
    lib = _lib.load()
    L = pt.numel()
    ptd = dev(pt.reshape(-1).contiguous())
    order = torch.full((L,), -7, dtype=torch.int32, device=DEV)
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=DEV)
    wsb = int(lib.ff_workspace_bytes(L, P))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    inv = torch.full((L,), -7, dtype=torch.int32, device=DEV)
    _lib.check(lib.ff_build_order(ptd.data_ptr(), L, P, order.data_ptr(), inv.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                  wsb, _lib.stream_ptr()), "ff_build_order")
    torch.cuda.synchronize()
    assert torch.equal(inv[order.long()].cpu(), torch.arange(L, dtype=torch.int32))        # the inverse permutation
    return order.cpu().long(), stats.cpu()


def check_order(pt, P):
    order, stats = hip_order(pt, P)
    want, _ = orc.by_patch_order(pt, P)
    nv = want.numel()
    assert int(stats[_lib.STAT_NV]) == nv
    assert int(stats[_lib.STAT_FTN]) == int((pt != -1).sum())
    assert torch.equal(order[:nv], want)
    flat = pt.reshape(-1)
    rest = torch.nonzero(~((flat >= 0) & (flat < P))).reshape(-1)
    assert torch.equal(order[nv:], rest)


@pytest.mark.parametrize("F,P,pre,post", [(8, 16, 3, 2), (64, 576, 14, 20), (5, 7, 0, 0), (3, 1, 1, 1),
                                           (40, 100, 0, 9), (7, 3000, 2, 2), (2, 20000, 0, 0)])
def test_order_regular(F, P, pre, post):
    pt = torch.cat((torch.full((pre,), -1), torch.arange(P).repeat(F), torch.full((post,), -1)))[None]
    check_order(pt, P)


def test_order_almost_regular():
    """Layouts that nearly match the frame-major closed form must still take the general sort."""
    F, P, pre = 6, 40, 5
    base = torch.cat((torch.full((pre,), -1), torch.arange(P).repeat(F), torch.full((3,), -1)))
    a = base.clone(); a[pre + 57] = -1                       # a text token inside the visual span
    b = base.clone(); b[pre + 90] = (b[pre + 90] + 1) % P    # one type off
    c = torch.cat((base[:pre + 3 * P], torch.full((2,), -1), base[pre + 3 * P:]))   # text between frames
    d = base[:-3 - 7]                                        # last frame incomplete
    e = torch.cat((torch.full((pre,), -1), (torch.arange(F * P) + 3) % P))          # first type is not 0
    for pt in (a, b, c, d, e):
        check_order(pt[None], P)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_order_ragged_random(seed):
    g = torch.Generator().manual_seed(seed)
    L = [1000, 4097, 70001][seed]
    P = [5, 70, 1300][seed]
    pt = torch.randint(-2, P + 3, (1, L), generator=g)     # text, IGNORE, out-of-range types included
    check_order(pt, P)
    check_order(torch.full((1, 333), -1), 4)               # no visual token at all
    check_order(torch.zeros(1, 129, dtype=torch.long), 1)  # one type only: every wave step collides


# ---------------------------------------------------------------------------------------------
# K1
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [c for c in SIM_CASES if c != "hand"])
def test_similarity_golden(golden, name):
    g = golden("similarity_merge")
    dtype = DT[str(g[f"{name}/dtype"])]
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    P = int(g[f"{name}/patch_num"])
    sim, order = ffa.FrameFusion.compute_similarity_and_token_index_by_patch(dev(h), dev(pt), P)
    assert np.array_equal(order[0].cpu().numpy(), g[f"{name}/order"])
    want = from_bits(g[f"{name}/sim"], dtype)
    got = sim[0].cpu()
    if bool(g[f"{name}/exact"]):
        assert same_bits(got, want)
    else:
        assert torch.allclose(got.float(), want.float(), rtol=2 ** -7 if dtype != torch.float32 else 1e-5, atol=0)


def test_hand_example(golden):
    g = golden("similarity_merge")
    h = torch.from_numpy(g["hand/hidden"])
    h = torch.cat((h, torch.zeros(h.shape[0], 4)), dim=1)[None]      # pad d to a 16-byte row (zeros are neutral)
    pt = torch.from_numpy(g["hand/patch_type"])[None]
    sim, order = ffa.FrameFusion.compute_similarity_and_token_index_by_patch(dev(h), dev(pt), 2)
    assert order.tolist() == [[1, 3, 5, 2, 4, 6]]
    assert np.allclose(sim[0].cpu().numpy(), g["hand/sim"], rtol=1e-6)


@pytest.mark.parametrize("dt,F,P,d,grid", [("bf16", 16, 33, 1024, 0.125), ("bf16", 9, 20, 3584, 0.125),
                                            ("fp16", 8, 17, 512, 0.125), ("fp32", 8, 17, 640, 0.125),
                                            ("bf16", 6, 11, 8192, 0.125), ("bf16", 12, 40, 4096, None)])
def test_similarity_vs_oracle(dt, F, P, d, grid):
    dtype = DT[dt]
    h, pt = video_tokens(F, P, d, p_change=0.3, sigma=0.2, sigma_hi=1.5, seed=5, pre=3, post=4, dtype=dtype, grid=grid)
    sim_o, ord_o = orc.pair_similarity(h, pt, P)
    sim, order = ffa.FrameFusion.compute_similarity_and_token_index_by_patch(dev(h), dev(pt), P)
    assert torch.equal(order.cpu(), ord_o)
    if grid is not None:
        assert same_bits(sim.cpu(), sim_o)
    else:
        a, b = sim.cpu().float(), sim_o.float()
        assert float((a != b).float().mean()) <= 2e-3
        assert torch.allclose(a, b, rtol=2 ** -7, atol=0)


def test_cosine_similarity_api():
    a = harness.snap(torch.randn(37, 256, generator=torch.Generator().manual_seed(1)), torch.bfloat16)
    b = harness.snap(torch.randn(37, 256, generator=torch.Generator().manual_seed(2)) * 0.5 + a.float(), torch.bfloat16)
    want = orc.staged_cosine(a, b)
    got = ffa.cosine_similarity(dev(a), dev(b))
    assert same_bits(got.cpu(), want)


# ---------------------------------------------------------------------------------------------
# plan + merge (static entry point, reference main.py:243-319)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [c for c in SIM_CASES if c != "hand"])
def test_merge_golden(golden, name):
    g = golden("similarity_merge")
    dtype = DT[str(g[f"{name}/dtype"])]
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    sim = from_bits(g[f"{name}/sim"], dtype)[None]
    order = torch.from_numpy(g[f"{name}/order"])[None]
    for sname in ("thr", "all", "alt", "none"):
        midx = torch.from_numpy(g[f"{name}/merge_{sname}/idx"])
        hd = dev(h.clone())
        merged, keep = ffa.FrameFusion.merge_tokens_and_get_mask(hd, dev(sim), dev(order), dev(midx))
        assert merged.data_ptr() == hd.data_ptr()                      # in place, like the reference
        assert np.array_equal(keep[0].cpu().numpy(), g[f"{name}/merge_{sname}/keep"]), sname
        assert same_bits(merged[0].cpu(), from_bits(g[f"{name}/merge_{sname}/hidden"], dtype)), sname


# ---------------------------------------------------------------------------------------------
# FrameFusion.forward
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_golden(golden, name):
    g = golden("forward")
    dtype = DT[str(g[f"{name}/dtype"])]
    F, P, d, pre, post = (int(x) for x in g[f"{name}/meta"])
    cost, thr, lb = (float(x) for x in g[f"{name}/params"])
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    L = h.shape[1]
    ff = ffa.FrameFusion(cost, thr, lb)
    ff.prepare(dev(pt), P, pre, pre + F * P, F * P, L)
    pos = make_pos(str(g[f"{name}/pos_kind"]), L, dtype)
    pos_in = [p.clone() for p in pos] if isinstance(pos, list) else pos.clone()
    pos = [dev(p) for p in pos] if isinstance(pos, list) else dev(pos)
    mask = None
    if bool(g[f"{name}/mask"]):
        mask = torch.zeros(1, 1, L, L, dtype=dtype).masked_fill_(torch.ones(L, L, dtype=torch.bool).triu(1), float("-inf"))
    hd = dev(h)
    out, pos_out, mask_out = ff(hd, pos, None if mask is None else dev(mask))
    keep = torch.from_numpy(g[f"{name}/keep"])
    got_keep = torch.nonzero(ff.last_plan()["keep"].bool()).reshape(-1).cpu() if ff.last_call else torch.arange(L)
    assert np.array_equal(got_keep.numpy(), keep.numpy())
    assert same_bits(out[0].cpu(), from_bits(g[f"{name}/hidden_out"], dtype))
    assert np.array_equal(ff.patch_type[0].cpu().numpy(), g[f"{name}/patch_type_out"])
    assert [ff.finish_merging, ff.finish_pruning] == [bool(x) for x in g[f"{name}/flags"]]
    assert ff.sparsity_list == list(g[f"{name}/sparsity"])
    if isinstance(pos_in, list):
        assert pos_out is pos                                           # list mutated in place (main.py:146-170)
        for a, b in zip(pos_out, pos_in):
            assert same_bits(a.cpu().contiguous(), b.index_select(b.ndim - 2, keep))
    else:
        assert torch.equal(pos_out.cpu(), pos_in[:, keep])
    if mask is not None:
        assert same_bits(mask_out.cpu().contiguous(), mask[:, :, keep][:, :, :, keep])
    assert same_bits(hd.cpu(), h)                                        # the fast path leaves its input alone


@pytest.mark.parametrize("name", CAS_CASES)
def test_cascade_golden(golden, name):
    g = golden("cascade")
    dtype = DT[str(g[f"{name}/dtype"])]
    F, P, d, pre, post, layers, heads, num = (int(x) for x in g[f"{name}/meta"])
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    L = h.shape[1]
    log, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), dev(h), dev(pt), P,
                                 dev(torch.arange(L)[None]), None, layers, heads, num)
    assert [r["length"] for r in log] == [int(x) for x in g[f"{name}/lengths"]]
    for r, fl, ns in zip(log, g[f"{name}/flags"], g[f"{name}/n_sparsity"]):
        assert [r["finish_merging"], r["finish_pruning"]] == [bool(fl[0]), bool(fl[1])], r["tag"]
        assert len(r["sparsity"]) == int(ns)
        assert np.array_equal(r["pos"][0].cpu().numpy(), g[f"{name}/{r['tag']}/index"]), r["tag"]
        assert same_bits(r["hidden"][0].cpu(), from_bits(g[f"{name}/{r['tag']}/hidden"], dtype)), r["tag"]
    assert log[-1]["sparsity"] == list(g[f"{name}/sparsity"])


def test_cascade_with_rotary_containers(golden):
    """Same cascades with [cos, sin] containers (3-D and M-RoPE 4-D) against the oracle."""
    g = golden("cascade")
    for name, kind in (("c_thr_prune", "qwen2"), ("c_low_prune", "mrope")):
        dtype = DT[str(g[f"{name}/dtype"])]
        F, P, d, pre, post, layers, heads, num = (int(x) for x in g[f"{name}/meta"])
        h = from_bits(g[f"{name}/hidden"], dtype)[None]
        pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
        L = h.shape[1]
        want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P,
                                      make_pos(kind, L, dtype), None, layers, heads, num)
        got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), dev(h), dev(pt), P,
                                     [dev(p) for p in make_pos(kind, L, dtype)], None, layers, heads, num)
        for a, b in zip(got, want):
            assert a["length"] == b["length"]
            assert same_bits(a["hidden"].cpu(), b["hidden"])
            for x, y in zip(a["pos"], b["pos"]):
                assert same_bits(x.cpu().contiguous(), y.contiguous())


# ---------------------------------------------------------------------------------------------
# selection rules
# ---------------------------------------------------------------------------------------------
def run_plan(sim, thr, sub, lb, ftn=None):
    lib = _lib.load()
    nv = sim.numel()
    dtype = sim.dtype
    simd = dev(sim.contiguous())
    order = torch.arange(nv, dtype=torch.int32, device=DEV)
    out = [torch.empty(nv, dtype=torch.uint8, device=DEV), torch.empty(nv, dtype=torch.int32, device=DEV),
           torch.empty(nv, dtype=torch.uint8, device=DEV)]
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=DEV)
    stats[_lib.STAT_NV] = nv
    stats[_lib.STAT_FTN] = nv if ftn is None else ftn
    wsb = int(lib.ff_workspace_bytes(nv, 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    thr_t = float(torch.tensor(thr, dtype=dtype))
    _lib.check(lib.ff_plan_merge(simd.data_ptr(), _lib.DTYPE_CODE[dtype], order.data_ptr(), nv, thr_t, sub, lb,
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stats.data_ptr(),
                                 ws.data_ptr(), wsb, _lib.stream_ptr()), "ff_plan_merge")
    torch.cuda.synchronize()
    return [o.cpu() for o in out], stats.cpu()


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,seed", [(50, 0), (4099, 1), (70000, 2)])
def test_select_threshold_and_topk(dt, n, seed):
    dtype = DT[dt]
    g = torch.Generator().manual_seed(seed)
    sim = (torch.rand(n, generator=g) * 1.2 - 0.1).to(dtype)       # few distinct bf16 values: many ties
    sim[0] = -2
    sim[torch.rand(n, generator=g) < 0.05] = -2
    if n > 100:
        sim[7] = float("nan")
    for sub in (0.9, 0.3, 0.05, 0.0):
        (member, dst, keep), st = run_plan(sim, 0.6, sub, 0.1)
        count = int((sim >= 0.6).sum())
        assert int(st[_lib.STAT_COUNT]) == count
        ratio = count / n
        if ratio < sub:
            assert int(st[_lib.STAT_BRANCH]) == 0
            members = torch.nonzero(sim >= 0.6).reshape(-1)
            assert int(st[_lib.STAT_BELOW_LB]) == int(ratio < 0.1)
        else:
            assert int(st[_lib.STAT_BRANCH]) == 1
            k = int(sub * n)
            assert int(st[_lib.STAT_K]) == k
            members = orc.topk_lowest_index(sim, k)
        members = members[members != 0]                               # position 0 never folds
        want_keep = torch.ones(n, dtype=torch.uint8)
        want_keep[members] = 0
        assert torch.equal(keep, want_keep), sub
        assert int(st[_lib.STAT_LOUT]) == int(want_keep.sum())
        assert torch.equal(dst[want_keep.bool()], torch.arange(int(want_keep.sum()), dtype=torch.int32))
        assert torch.equal(member, 1 - want_keep), sub        # order = identity here


# ---------------------------------------------------------------------------------------------
# importance
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", IMP_CASES)
def test_importance_golden(golden, name):
    g = golden("importance")
    dtype = DT[str(g[f"{name}/dtype"])]
    H, Hk, S, dh, num, causal = (int(x) for x in g[f"{name}/meta"])
    q = from_bits(g[f"{name}/q"], dtype)[None]
    k = from_bits(g[f"{name}/k"], dtype)[None]
    w = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, is_causal=bool(causal), enable_gqa=H != Hk)
    want = from_bits(g[f"{name}/weights"], dtype)
    tol = 2 ** -7 if dtype != torch.float32 else 2e-6
    assert torch.allclose(w[0].cpu().float(), want.float(), rtol=tol, atol=1e-30)
    imp = ffa.last_query_importance(dev(q), dev(k), num=num, is_causal=bool(causal))
    assert imp.shape == (1, 1, 1, S)
    assert torch.allclose(imp.reshape(-1).cpu().float(), from_bits(g[f"{name}/importance"], dtype).float(),
                          rtol=tol, atol=1e-30)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dh,H,Hk,num,S", [(64, 8, 2, 1, 700), (128, 28, 4, 4, 1111), (256, 4, 4, 1, 300), (128, 40, 4, 4, 520)])
def test_importance_matrix_core_path(dtype, dh, H, Hk, num, S):
    """dh in {64, 128, 256}, 16-bit T: q . K^T on MFMA (rows of a GQA group padded to 32, more than 32 rows in
    passes).  Grid inputs make every product sum exact, so the scores are the oracle's and what is left is the
    softmax's ulp noise."""
    g = torch.Generator().manual_seed(dh + S)
    q = harness.snap(0.5 * torch.randn(1, H, num, dh, generator=g), dtype)
    k = harness.snap(0.5 * torch.randn(1, Hk, S, dh, generator=g), dtype)
    want = orc.last_query_attention(q, k, num=num, is_causal=True, enable_gqa=True)
    got = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, is_causal=True, enable_gqa=True)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    assert torch.allclose(got.cpu().float(), want.float(), rtol=tol, atol=1e-30)
    assert float((got.cpu().float() != want.float()).float().mean()) <= 2e-3
    imp = ffa.last_query_importance(dev(q), dev(k), num=num, is_causal=True)
    assert torch.allclose(imp.reshape(-1).cpu().float(), torch.mean(want, dim=(1, 2))[0].float(), rtol=tol, atol=1e-30)


def test_importance_odd_head_size_general_path():
    """dh * sizeof(T) / 16 not a power of two (dh = 24 bf16 -> 3 lanes per key): the general kernels"""
    g = torch.Generator().manual_seed(5)
    q = harness.snap(torch.randn(1, 6, 5, 24, generator=g), torch.bfloat16)
    k = harness.snap(torch.randn(1, 2, 300, 24, generator=g), torch.bfloat16)
    for num in (1, 4):
        want = orc.last_query_attention(q, k, num=num, is_causal=True, enable_gqa=True)
        got = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, is_causal=True, enable_gqa=True)
        assert torch.allclose(got.cpu().float(), want.float(), rtol=2 ** -7, atol=1e-30)
        assert float((got.cpu().float() != want.float()).float().mean()) <= 5e-3
        imp = ffa.last_query_importance(dev(q), dev(k), num=num, is_causal=True)
        assert torch.allclose(imp.reshape(-1).cpu().float(), torch.mean(want, dim=(1, 2))[0].float(), rtol=2 ** -7, atol=1e-30)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("H,num,S", [(28, 4, 5000), (1, 1, 64), (3, 11, 4097), (8, 16, 1003), (65, 8, 777), (4, 1, 36898)])
def test_head_mean_exact(dtype, H, num, S):
    """1, 33, 112, 128 and 520 rows (one, several, and several groups of 32-row passes), lengths that are and are not
    whole 16-byte words (the scalar-load path); the stub's sums are exact in fp32, so the mean is torch's to the bit."""
    w = harness.attention_stub(H, num, S, dtype)
    want = torch.mean(w, dim=(1, 2))[0]
    lib = _lib.load()
    wd = dev(w)
    imp = torch.empty(S, dtype=dtype, device=DEV)
    _lib.check(lib.ff_head_mean(wd.data_ptr(), _lib.DTYPE_CODE[dtype], H, num, S, imp.data_ptr(), _lib.stream_ptr()), "hm")
    assert same_bits(imp.cpu(), want)


# ---------------------------------------------------------------------------------------------
# errors and edge cases (reference behaviour)
# ---------------------------------------------------------------------------------------------
def test_edges_and_errors():
    f = ffa.FrameFusion()
    f.prepare(dev(torch.tensor([[-1]])), 4, 0, 0, 0, 1)
    tok = torch.zeros(1, 1, 8, device=DEV)
    r = f(tok, "pos", "mask")
    assert r[0] is tok and r[1] == "pos" and r[2] == "mask"
    f.prepare(dev(torch.tensor([[-1, -1, -1]])), 4, 0, 0, 0, 3)
    with pytest.raises(AssertionError):
        f(torch.zeros(1, 3, 8, device=DEV), [torch.zeros(1, 3, 2, device=DEV)] * 2, None)
    for bad in ((torch.zeros(1, 4, 2, device=DEV),) * 2, torch.zeros(1, 4, 2, device=DEV)):
        f.prepare(dev(torch.tensor([[0, 1, 0, 1]])), 2, 0, 4, 4, 4)
        with pytest.raises(NotImplementedError):
            f(torch.ones(1, 4, 8, device=DEV), bad, None)
    f.prepare(dev(torch.tensor([[0, 1, 0, 1]])), 2, 0, 4, 4, 4, sparsity_list=[0] * 10)
    with pytest.raises(ValueError, match="The cost is too small"):
        f(torch.ones(1, 4, 8, device=DEV), torch.arange(4, device=DEV)[None], None)
    g = ffa.FrameFusion()
    with pytest.raises(AttributeError):
        g(torch.zeros(1, 3, 8, device=DEV), None, None)
    with pytest.raises(ffa.FrameFusionHipError):                  # no CPU path in the product
        f.prepare(torch.tensor([[0, 1, 0, 1]]), 2, 0, 4, 4, 4)
        f(torch.ones(1, 4, 8), torch.arange(4)[None], None)


def test_order_maintenance():
    """The merge kernel hands the next call the by-patch order of the compacted sequence (K0 is then
    skipped): it must equal what K0 computes from the new patch_type, call after call."""
    h, pt = video_tokens(20, 37, 256, p_change=0.6, sigma=0.3, sigma_hi=1.6, seed=4, pre=6, post=9, grid=0.125)
    L = h.shape[1]
    f = ffa.FrameFusion(0.3, 0.6, 0.01)
    f.prepare(dev(pt), 37, 6, 6 + 20 * 37, 20 * 37, L)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.01)
    o.prepare(pt.clone(), 37, 6, 6 + 20 * 37, 20 * 37, L)
    hg, pg = dev(h), dev(torch.arange(L)[None])
    ho, po = h, torch.arange(L)[None]
    merges = 0
    for layer in range(4):
        if f.finish_merging:
            break
        hg, pg, _ = f(hg, pg, None)
        ho, po, _ = o.forward(ho, po, None)
        merges += 1
        assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
        sc = f.last_call["scratch"]
        n = hg.shape[1]
        assert sc.order_valid_for == (f._ptype_gen, n)
        fresh, stats = hip_order(f.patch_type.cpu(), 37)
        assert torch.equal(sc.order[:n].cpu().long(), fresh)
        assert int(sc.stats[_lib.STAT_NV]) == int(stats[_lib.STAT_NV])
        assert int(sc.stats[_lib.STAT_FTN]) == int(stats[_lib.STAT_FTN])
        hg, ho = harness.layer_stub(hg, layer), harness.layer_stub(ho, layer)
    assert merges >= 2          # at least one call ran on a maintained order


def test_recovery_after_a_failed_call():
    """A call that dies between the two halves of the step (bad position container) must not poison
    the next one: the select tables accumulated by its similarity pass are reset."""
    h, pt = video_tokens(10, 20, 128, p_change=0.2, seed=8, pre=2, post=3, grid=0.125)
    L = h.shape[1]
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), 20, 2, 202, 200, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    for attempt in range(3):
        f.prepare(dev(pt), 20, 2, 202, 200, L)
        with pytest.raises(NotImplementedError):
            f(dev(h), torch.zeros(1, L, 2, device=DEV), None)          # 3-D tensor container: rejected
        f.prepare(dev(pt), 20, 2, 202, 200, L)
        hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
        assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho), attempt


def test_determinism():
    h, pt = video_tokens(16, 48, 1024, p_change=0.3, sigma_hi=1.5, seed=9, pre=5, post=5)
    outs = []
    for _ in range(3):
        f = ffa.FrameFusion(0.3, 0.6, 0.1)
        f.prepare(dev(pt), 48, 5, 5 + 16 * 48, 16 * 48, h.shape[1])
        o, p, _ = f(dev(h), dev(torch.arange(h.shape[1])[None]), None)
        outs.append((o.cpu(), p.cpu()))
    for o, p in outs[1:]:
        assert same_bits(o, outs[0][0]) and torch.equal(p, outs[0][1])


# ---------------------------------------------------------------------------------------------
# layout hint: closed-form by-patch order inside the similarity kernel, verified on the device
# ---------------------------------------------------------------------------------------------
def one_call(pt, P, start, length, h, cost=0.3, thr=0.6, lb=0.1):
    L = h.shape[1]
    f = ffa.FrameFusion(cost, thr, lb)
    f.prepare(dev(pt), P, start, start + length - 1, length, L)
    o = orc.OracleFrameFusion(cost, thr, lb)
    o.prepare(pt.clone(), P, start, start + length - 1, length, L)
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    return f, o, (hg, pg), (ho, po)


@pytest.mark.parametrize("F,P,pre,post", [(9, 16, 3, 5), (5, 33, 0, 0), (64, 7, 11, 0), (3, 300, 0, 2), (1, 8, 2, 2)])
def test_layout_hint_gives_the_order_k0_builds(F, P, pre, post):
    h, pt = video_tokens(F, P, 64, p_change=0.4, sigma=0.3, seed=21, pre=pre, post=post, grid=0.125)
    f, o, (hg, pg), (ho, po) = one_call(pt, P, pre, F * P, h)
    assert f._layout_hint == (pre, F)
    assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
    assert (f.finish_merging, f.finish_pruning, f.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list)
    L = h.shape[1]
    fresh, stats = hip_order(pt, P)                       # what K0 computes for the same row
    sc = f.last_call["scratch"]
    assert torch.equal(f.last_call["order"][:L].cpu().long(), fresh)
    assert f.last_call["nv"] == F * P == int(stats[_lib.STAT_NV]) and f.last_call["ftn"] == int(stats[_lib.STAT_FTN])
    assert torch.equal(f.last_plan()["sim"].cpu().float(), o.last_sim[0].float())
    assert int(sc.stats[_lib.STAT_ERROR]) == 0


def test_layout_hint_mismatch_falls_back_to_k0():
    """prepare() scalars that suggest whole frames while patch_type says otherwise: the device flags
    the mismatch, the call is repeated through K0 and the result is the reference's."""
    F, P, d = 6, 10, 64
    h, pt = video_tokens(F, P, d, p_change=0.4, sigma=0.3, seed=8, pre=4, post=6, grid=0.125)
    L = h.shape[1]
    variants = {}
    a = pt.clone(); a[0, 4 + 2 * P + 3] = -1                      # a text token inside a frame
    variants["hole"] = a
    b = pt.clone(); b[0, 1] = 5                                   # a visual token in front of the hinted range
    variants["stray_before"] = b
    c = pt.clone(); c[0, L - 2] = 0                               # ... and behind it
    variants["stray_after"] = c
    e = pt.clone(); e[0, 4:4 + P] = torch.arange(P).flip(0)       # first frame typed backwards
    variants["permuted_frame"] = e
    g = pt.clone(); g[0, 4 + P] = P + 3                           # an out-of-range type (non-text, non-visual)
    variants["foreign_type"] = g
    for name, row in variants.items():
        f, o, (hg, pg), (ho, po) = one_call(row, P, 4, F * P, h)
        assert f._layout_hint is None, name                        # stopped hinting for this prefill
        assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho), name
        assert torch.equal(f.patch_type.cpu(), o.patch_type), name
        fresh, stats = hip_order(row, P)
        assert f.last_call["nv"] == int(stats[_lib.STAT_NV]) and f.last_call["ftn"] == int(stats[_lib.STAT_FTN]), name
        sc = f.last_call["scratch"]
        assert int(sc.stats[_lib.STAT_ERROR]) == 0, name           # published and cleared
        # the instance keeps working: a second prefill with a correct row is hinted again
        f.prepare(dev(pt), P, 4, 4 + F * P - 1, F * P, L)
        assert f._layout_hint == (4, F)
        hg2, pg2, _ = f(dev(h), dev(torch.arange(L)[None]), None)
        o2 = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        o2.prepare(pt.clone(), P, 4, 4 + F * P - 1, F * P, L)
        ho2, po2, _ = o2.forward(h, torch.arange(L)[None], None)
        assert torch.equal(pg2.cpu(), po2) and same_bits(hg2.cpu(), ho2), name


def test_layout_hint_from_the_scalars_the_packers_pass():
    """The reference's packers hand prepare() 0-d / 1-element DEVICE tensors
    (llava_video/modeling_llava_video.py:332, qwenvl/modeling_qwen2_vl.py:123): they are read back once, in
    prepare(), and the hinted path runs - the path bench.py measures is the one real packers reach."""
    F, P = 5, 12
    h, pt = video_tokens(F, P, 64, p_change=0.4, sigma=0.3, seed=2, pre=3, post=3, grid=0.125)
    L = h.shape[1]
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    start_dev = torch.tensor([3], device=DEV)
    f.prepare(dev(pt), P, start_dev, torch.tensor([3 + F * P - 1], device=DEV), F * P, L)
    assert f._layout_hint == (3, F)
    assert f.image_token_start_index is start_dev                  # the attribute stays what the caller passed
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), P, 3, 3 + F * P - 1, F * P, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
    assert f.last_call["scratch"].ctx.swaps == 1 and not f.last_call["scratch"].dirty
    f.prepare(dev(pt), P, torch.tensor(3.5, device=DEV), 0, F * P, L)
    assert f._layout_hint is None                                  # not an index
    f.prepare(dev(pt), P, torch.tensor([3]), torch.tensor(3 + F * P - 1), torch.tensor(F * P), L)
    assert f._layout_hint == (3, F)                                # 0-d / 1-element CPU tensors are fine
    f.prepare(dev(pt), float(P), 3, 3 + F * P - 1, F * P, L)
    assert f._layout_hint == (3, F)                                # nvila hands patch_num over as a float
    f.prepare(dev(pt), P, 3, 0, F * P + 1, L)
    assert f._layout_hint is None                                  # not whole frames
    f.prepare(dev(pt), P, 3, 0, F * P, L)
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), P, 3, 0, F * P, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
    # a hint that does not fit the sequence is dropped on the host
    f.prepare(dev(pt), P, 3, 0, (F + 2) * P, L)
    assert f._layout_hint == (3, F + 2)
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho)
