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
from tests.test_oracle_golden import make_pos, SIM_CASES, FWD_CASES, CAS_CASES, IMP_CASES, MASK_CASES, load_mask

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


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_merge_arbitrary_bit_patterns(dtype, seed):
    """The fold on values the grid fixtures never contain: every exponent (subnormal quotients, overflowing sums), -0,
    infinities, runs of 1..20+ members (divisors that are not powers of two).  The bf16 flush multiplies by one IEEE
    reciprocal instead of dividing (csrc/ff_merge.hip, proven exhaustively on the CPU in tests/test_host_logic.py): the
    result must still be the oracle's `rows / (n + 1)` bit for bit."""
    g = torch.Generator().manual_seed(seed)
    F, P, d = 48, 8, 64
    bits = torch.randint(0, 65536, (F * P, d), generator=g, dtype=torch.int32)
    expo_mask, expo_max = (0x7f80, 0x7f00) if dtype == torch.bfloat16 else (0x7c00, 0x7800)
    is_special = (bits & expo_mask) == expo_mask                      # no NaN / inf from the generator (payloads differ by device)
    bits = torch.where(is_special, (bits & ~expo_mask) | expo_max, bits)
    if seed == 1:                                                       # small magnitudes: quotients in the subnormal range
        bits = bits & (0x80ff if dtype == torch.bfloat16 else 0x83ff) | (torch.randint(0, 3, bits.shape, generator=g, dtype=torch.int32) << (7 if dtype == torch.bfloat16 else 10))
    h = bits.to(torch.int16).view(dtype)[None].clone()
    h[0, 5] = 0.0
    h[0, 6] = -0.0
    h[0, 7, :8] = float("inf")
    order, _ = orc.by_patch_order(torch.arange(P).repeat(F)[None], P)
    nv = F * P
    flags = torch.rand(nv, generator=g) < 0.75
    flags[torch.arange(P) * F] = False                                  # a chain's first slot has no predecessor
    midx = torch.nonzero(flags).reshape(-1)
    want, keep_want = orc.merge_rows(h.clone(), order, midx)
    sim = torch.zeros(1, nv, dtype=dtype)
    got, keep = ffa.FrameFusion.merge_tokens_and_get_mask(dev(h.clone()), dev(sim), dev(order.reshape(1, -1)), dev(midx))
    assert torch.equal(keep.cpu(), keep_want)
    a, b = got[0].cpu()[keep_want[0]], want[0][keep_want[0]]
    same = (a.view(torch.int16) == b.view(torch.int16)) | (torch.isnan(a.float()) & torch.isnan(b.float()))
    assert bool(same.all()), f"{int((~same).sum())} elements differ"
    lens = orc.run_lengths(flags[None].to(torch.long))[0]
    assert int(lens.max()) >= 9                                         # long runs are really in there


@pytest.mark.parametrize("pattern", ["all_but_first", "alternating", "period_9_10_11", "period_20_21", "dense_0.95", "none",
                                     "one_giant_run"])
@pytest.mark.parametrize("F,P", [(200, 24), (64, 576)])
def test_merge_run_patterns_across_stream_boundaries(pattern, F, P):
    """The merge kernel cuts the by-patch order into streams at the non-member slot nearest to a multiple of its slot count
    (csrc/ff_merge.hip `boundary`: look-back of up to 10 slots, else the first non-member at or after the nominal boundary,
    a forward scan when the window holds none).  Member patterns that put runs of every length across those boundaries -
    runs longer than a window, runs of about the look-back length, no members at all - must fold exactly like the oracle."""
    d = 64
    g = torch.Generator().manual_seed(F * 1000 + P)
    h = (torch.randint(-8, 9, (1, F * P, d), generator=g).to(torch.float32) * 0.125).to(torch.bfloat16)     # dyadic grid: exact sums
    order, _ = orc.by_patch_order(torch.arange(P).repeat(F)[None], P)
    nv = F * P
    f = torch.arange(nv) % F                                    # frame of a by-patch slot (slot = patch * F + frame)
    if pattern == "all_but_first":
        flags = f != 0
    elif pattern == "alternating":
        flags = (f % 2) == 1
    elif pattern == "period_9_10_11":
        flags = ((torch.arange(nv) % 31) != 0) & ((torch.arange(nv) % 31) != 10) & ((torch.arange(nv) % 31) != 21) & (f != 0)
    elif pattern == "period_20_21":
        flags = ((torch.arange(nv) % 43) != 0) & ((torch.arange(nv) % 43) != 21) & (f != 0)
    elif pattern == "dense_0.95":
        flags = (torch.rand(nv, generator=g) < 0.95) & (f != 0)
    elif pattern == "none":
        flags = torch.zeros(nv, dtype=torch.bool)
    else:                                                       # every slot but the very first: one run over all patches
        flags = torch.ones(nv, dtype=torch.bool)
        flags[0] = False
    midx = torch.nonzero(flags).reshape(-1)
    stock = orc.run_lengths
    orc.run_lengths = lambda fl: stock(fl.to(torch.long))       # exact integer run lengths (deviation 2)
    try:
        want, keep_want = orc.merge_rows(h.clone(), order, midx)
    finally:
        orc.run_lengths = stock
    sim = torch.zeros(1, nv, dtype=h.dtype)
    got, keep = ffa.FrameFusion.merge_tokens_and_get_mask(dev(h.clone()), dev(sim), dev(order.reshape(1, -1)), dev(midx))
    assert torch.equal(keep.cpu(), keep_want)
    kept = keep_want[0]
    assert same_bits(got[0].cpu()[kept], want[0][kept])


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


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dh,H,Hk,num,S", [(128, 64, 8, 1, 4099), (128, 28, 4, 1, 3001), (64, 16, 2, 1, 1000), (128, 8, 4, 2, 777), (64, 4, 2, 4, 130),
                                           (128, 8, 1, 1, 9), (128, 5, 5, 1, 1), (64, 24, 3, 1, 65), (128, 16, 2, 1, 20000)])
def test_importance_dot_path(dtype, dh, H, Hk, num, S):
    """At most 8 query rows per kv head and dh in {64, 128} (num = 1 of LLaVA-Video 7B / 72B): the v_dot2c kernel - 8 lanes
    per key, whole cache lines, transposed 8-lane reduction, one statistics entry per workgroup.  Grid inputs make every
    product sum exact (any summation order gives the oracle's scores); what is left is the softmax's ulp noise.  Lengths
    around the 8-key group, the 64-key step and the workgroup's run; 1 to 8 rows."""
    g = torch.Generator().manual_seed(dh + S + H)
    q = harness.snap(0.5 * torch.randn(1, H, num, dh, generator=g), dtype)
    k = harness.snap(0.5 * torch.randn(1, Hk, S, dh, generator=g), dtype)
    want = orc.last_query_attention(q, k, num=num, is_causal=True, enable_gqa=True)
    got = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, is_causal=True, enable_gqa=True)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10

    def ulps(a, b):       # non-negative 16-bit values: neighbouring bit patterns are neighbouring values (fp16 subnormals included)
        return int((a.cpu().view(torch.int16).int() - b.cpu().view(torch.int16).int()).abs().max())
    assert got.shape == want.shape
    assert ulps(got, want) <= 1
    # (fp16's 11-bit mantissa sees the fp32 order of the row's exponentials more often: the soak's bound, tests/soak_gpu.py)
    assert float((got.cpu().float() != want.float()).float().mean()) <= (2e-3 if dtype == torch.bfloat16 else 1.6e-2)
    imp = ffa.last_query_importance(dev(q), dev(k), num=num, is_causal=True).reshape(-1)
    assert ulps(imp, torch.mean(got, dim=(1, 2))[0]) <= 1 and ulps(imp, torch.mean(want, dim=(1, 2))[0]) <= 2
    # gaussian (off-grid) inputs: the fp32 order of a score's 128 products differs from the oracle's, so a score now and
    # then rounds to the neighbouring T value (and its weight moves by more than an ulp): rare, and bounded
    q2 = torch.randn(1, H, num, dh, generator=g).to(dtype)
    k2 = torch.randn(1, Hk, S, dh, generator=g).to(dtype)
    want2 = orc.last_query_attention(q2, k2, num=num, is_causal=True, enable_gqa=True).float()
    got2 = ffa.scaled_dot_product_attention(dev(q2), dev(k2), None, num=num, is_causal=True, enable_gqa=True).cpu().float()
    off = ((got2 - want2).abs() > 2 * tol * want2.abs()).float().mean()
    assert float(off) <= 5e-3, float(off)
    assert torch.allclose(got2, want2, rtol=0.1, atol=1e-30)


def test_importance_owner_timeout_same_bits():
    """The finish kernel's row statistics come from owner workgroups through tagged granules; a workgroup that gets no
    answer folds the rows itself after a timeout.  FF_LQ_TEST_NO_PUBLISH silences the owners (a child process: the
    switch is read once per process): weights and importance must be the same BITS as with the exchange, at shapes with
    one owner workgroup, several, more rows than one polling pass (H * num = 448) and a one-workgroup grid."""
    import os, subprocess, sys, tempfile
    code = r'''
import sys, torch
import framefusion_amd as ffa
out = {}
for dh, H, Hk, num, S, dt in [(128, 28, 4, 4, 1111, torch.bfloat16), (64, 8, 2, 1, 700, torch.float16), (128, 64, 8, 1, 9000, torch.bfloat16),
                              (128, 28, 4, 16, 40, torch.bfloat16), (32, 6, 3, 1, 333, torch.float32)]:
    g = torch.Generator().manual_seed(dh + S)
    q = torch.randn(1, H, num, dh, generator=g).to(dt).cuda()
    k = torch.randn(1, Hk, S, dh, generator=g).to(dt).cuda()
    w = ffa.scaled_dot_product_attention(q, k, None, num=num, is_causal=True, enable_gqa=True)
    imp = ffa.last_query_importance(q, k, num=num, is_causal=True)
    out[f"w{S}"] = w.float().cpu()
    out[f"i{S}"] = imp.float().cpu()
torch.save(out, sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, extra in enumerate(({}, {"FF_LQ_TEST_NO_PUBLISH": "1"})):
            env = {k: v for k, v in os.environ.items() if k != "FF_LQ_TEST_NO_PUBLISH"}
            env.update(extra)
            env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
            path = os.path.join(tmp, f"o{i}.pt")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=root, timeout=600)
            res.append(torch.load(path))
    assert res[0].keys() == res[1].keys() and len(res[0]) == 10
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), key
        assert torch.isfinite(a).all() and float(a.abs().sum()) > 0, key


@pytest.mark.parametrize("name", MASK_CASES)
def test_importance_attn_mask_golden(golden, name):
    """scaled_dot_product_attention(attn_mask=...) (utils.py:40-44) against the real reference's outputs."""
    g = golden("importance_mask")
    dtype = DT[str(g[f"{name}/dtype"])]
    H, Hk, S, dh, num = (int(x) for x in g[f"{name}/meta"])
    q = from_bits(g[f"{name}/q"], dtype)[None]
    k = from_bits(g[f"{name}/k"], dtype)[None]
    mask = load_mask(g, name)
    w = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, attn_mask=dev(mask), enable_gqa=H != Hk)
    want = from_bits(g[f"{name}/weights"], dtype)
    tol = 2 ** -7 if dtype != torch.float32 else 2e-6
    assert torch.allclose(w[0].cpu().float(), want.float(), rtol=tol, atol=1e-30)
    with pytest.raises(AssertionError):
        ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, attn_mask=dev(mask), is_causal=True)


@pytest.mark.parametrize("kind", ["bool", "add"])
@pytest.mark.parametrize("dh,H,Hk,num,S", [(128, 28, 4, 4, 1111), (64, 8, 2, 1, 700), (128, 64, 8, 1, 5000)])
def test_importance_attn_mask_matrix_core_path(kind, dh, H, Hk, num, S):
    """The same on the MFMA score kernel (16-bit T, dh 64 / 128) at sizes with several tiles per row - the rows'
    statistics folded by the last workgroup of every kv head - against the oracle on grid inputs."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(dh + S + num)
    q = harness.snap(0.5 * torch.randn(1, H, num, dh, generator=g), dtype)
    k = harness.snap(0.5 * torch.randn(1, Hk, S, dh, generator=g), dtype)
    if kind == "bool":
        mask = torch.rand(num, S, generator=g) < 0.6
        mask[:, S // 2] = True
    else:
        mask = harness.snap(torch.randn(1, S, generator=g), dtype)
        mask[:, 5] = float("-inf")
    want = orc.last_query_attention(q, k, num=num, enable_gqa=True, attn_mask=mask)
    for rep in range(3):                                    # the workspace (and its arrival counters) is reused call after call
        got = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, attn_mask=dev(mask), enable_gqa=True)
        assert torch.allclose(got.cpu().float(), want.float(), rtol=2 ** -7, atol=1e-30), rep
        assert float((got.cpu().float() != want.float()).float().mean()) <= 2e-3, rep


@pytest.mark.parametrize("dtype,dh,H,Hk,num,S", [(torch.bfloat16, 128, 28, 4, 1, 3001), (torch.float16, 64, 8, 2, 4, 777),
                                               (torch.float32, 32, 4, 2, 1, 300), (torch.bfloat16, 24, 6, 2, 1, 200)])
def test_importance_reads_transposed_keys_in_place(dtype, dh, H, Hk, num, S, monkeypatch):
    """What an attention module hands over during prefill is the transposed VIEW of its k_proj output ([S, H_kv, dh] in
    memory, strides (dh, H_kv * dh, 1)): every K5 kernel reads it through its strides - same numbers as from a contiguous
    copy, and no copy of K is made on the way."""
    g = torch.Generator().manual_seed(S)
    q = harness.snap(0.5 * torch.randn(1, H, num, dh, generator=g), dtype)
    k_mem = harness.snap(0.5 * torch.randn(1, S, Hk, dh, generator=g), dtype)           # [1, S, H_kv, dh] as projected
    k_view = dev(k_mem).transpose(1, 2)                                                 # [1, H_kv, S, dh], not contiguous
    assert not k_view.is_contiguous()
    want = ffa.scaled_dot_product_attention(dev(q), k_view.contiguous(), None, num=num, is_causal=True, enable_gqa=True)
    copies = []
    real = torch.Tensor.contiguous
    monkeypatch.setattr(torch.Tensor, "contiguous", lambda t, *a, **kw: (copies.append(tuple(t.shape)) if not t.is_contiguous() else None, real(t, *a, **kw))[1])
    got = ffa.scaled_dot_product_attention(dev(q), k_view, None, num=num, is_causal=True, enable_gqa=True)
    imp = ffa.last_query_importance(dev(q), k_view, num=num, is_causal=True)
    monkeypatch.undo()
    assert all(len(shape) != 3 or shape[1] != S for shape in copies), copies                # K itself was never copied
    assert same_bits(got.cpu(), want.cpu())
    ref = orc.last_query_attention(q, k_mem.transpose(1, 2), num=num, is_causal=True, enable_gqa=True)
    tol = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10, torch.float32: 2e-6}[dtype]
    assert torch.allclose(got.cpu().float(), ref.float(), rtol=tol, atol=1e-30)
    assert torch.allclose(imp.reshape(-1).cpu().float(), torch.mean(ref, dim=(1, 2))[0].float(), rtol=tol, atol=1e-30)


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


def test_importance_more_than_4096_query_rows_takes_the_general_path_with_its_own_workspace():
    """H * num > 4096 with a power-of-two head size (28 heads x 256 queries): launch_lq takes the general kernels, whose
    workspace is 2 * H * num * S floats - ff_last_query_workspace_bytes must size for THAT path (it used to answer with the
    tiled path's size: an out-of-bounds device write).  The bytes are checked through the C ABI, the values against the
    oracle."""
    H, Hk, num, S, dh = 28, 4, 256, 300, 128
    lib = _lib.load()
    assert lib.ff_last_query_workspace_bytes(_lib.FF_BF16, H, num, S, dh) == 2 * H * num * S * 4
    assert lib.ff_last_query_workspace_bytes(_lib.FF_BF16, H, 4, S, dh) < 2 * H * 4 * S * 4          # (tiled: 2-byte scores)
    g = torch.Generator().manual_seed(11)
    q = harness.snap(torch.randn(1, H, num, dh, generator=g), torch.bfloat16)
    k = harness.snap(torch.randn(1, Hk, S, dh, generator=g), torch.bfloat16)
    want = orc.last_query_attention(q, k, num=num, is_causal=True, enable_gqa=True)
    guard = torch.full((1 << 22,), 7, dtype=torch.uint8, device=DEV)        # (whatever the allocator puts behind the workspace)
    got = ffa.scaled_dot_product_attention(dev(q), dev(k), None, num=num, is_causal=True, enable_gqa=True)
    imp = ffa.last_query_importance(dev(q), dev(k), num=num, is_causal=True)
    torch.cuda.synchronize()
    assert bool((guard == 7).all())
    assert torch.allclose(got.cpu().float(), want.float(), rtol=2 ** -7, atol=1e-30)
    assert float((got.cpu().float() != want.float()).float().mean()) <= 5e-3
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


@pytest.mark.parametrize("container", ["ids", "rotary", "mrope"])
def test_compact_outputs_returns_exactly_sized_tensors(container):
    """FrameFusion(compact_outputs=True), the default: the tensors a merge call returns own exactly their bytes (the reference's
    `hidden_states[token_mask, :]`, main.py:132-138); compact_outputs=False returns views of input-length buffers; same values."""
    from framefusion_amd.synth import rotary_tables
    F, P, d = 16, 48, 1024
    h, pt = video_tokens(F, P, d, p_change=0.2, sigma_hi=1.5, seed=9, pre=5, post=5, grid=0.125)
    L = h.shape[1]

    def positions():
        if container == "ids":
            return torch.arange(L)[None]
        return list(rotary_tables(L, 64, torch.bfloat16, mrope=(container == "mrope")))

    got = {}
    for compact in (False, True):
        f = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=compact)
        f.prepare(dev(pt), P, 5, 5 + F * P - 1, F * P, L)
        pe = positions()
        pe = [dev(t) for t in pe] if isinstance(pe, list) else dev(pe)
        out, pos, _ = f(dev(h), pe, None)
        got[compact] = (out, pos, f.patch_type)
    out_v, pos_v, pt_v = got[False]
    out_c, pos_c, pt_c = got[True]
    L_out = out_c.shape[1]
    assert 2 * L_out < L                                                       # (the top-k regime of this sample: compaction applies)
    tensors_c = [out_c, pt_c] + (pos_c if isinstance(pos_c, list) else [pos_c])
    tensors_v = [out_v, pt_v] + (pos_v if isinstance(pos_v, list) else [pos_v])
    for c, v in zip(tensors_c, tensors_v):
        assert c.shape == v.shape and torch.equal(c.cpu().view(torch.uint8) if c.dtype == torch.bfloat16 else c.cpu(),
                                                  v.cpu().view(torch.uint8) if v.dtype == torch.bfloat16 else v.cpu())
        assert c.is_contiguous()
    # no byte is held beyond the results: every storage is exactly as large as the tensors that live in it (cos and sin share one)
    by_storage = {}
    for c in tensors_c:
        by_storage.setdefault(c.untyped_storage().data_ptr(), [c.untyped_storage().nbytes(), 0])[1] += c.numel() * c.element_size()
    assert all(total == used for total, used in by_storage.values()), by_storage
    assert out_v.untyped_storage().nbytes() == L * d * 2                        # compact_outputs=False: a view of an L-row buffer
    assert ffa.FrameFusion().compact_outputs is True                           # exactly sized outputs are the default (round 5)


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
        assert f.last_call["unhinted"], name                       # (the library repeated the call through K0)
        # the instance keeps working - and has learnt: scalars that described a wrong layout once are not hinted again (a packer
        # that puts separators between its frames does so for every prompt; each wrong hint costs a whole wasted call, a
        # missing one only K0's closed form), other scalars are
        f.prepare(dev(pt), P, 4, 4 + F * P - 1, F * P, L)
        assert f._layout_hint is None
        hg2, pg2, _ = f(dev(h), dev(torch.arange(L)[None]), None)
        assert not f.last_call["unhinted"]
        o2 = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        o2.prepare(pt.clone(), P, 4, 4 + F * P - 1, F * P, L)
        ho2, po2, _ = o2.forward(h, torch.arange(L)[None], None)
        assert torch.equal(pg2.cpu(), po2) and same_bits(hg2.cpu(), ho2), name
        h3, pt3 = video_tokens(F + 1, P, d, p_change=0.4, sigma=0.3, seed=8, pre=4, post=6, grid=0.125)
        f.prepare(dev(pt3), P, 4, 4 + (F + 1) * P - 1, (F + 1) * P, h3.shape[1])
        assert f._layout_hint == (4, F + 1), name


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


# ---------------------------------------------------------------------------------------------
# the reference's public position handlers (main.py:142-178) and the attention-mask gather
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["qwen2", "mrope", "ids"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_position_handlers_match_torch_indexing(kind, dtype):
    """position_embedding_handler_at_pruning / _at_merging are plain index / boolean-mask gathers along the token axis
    (main.py:142-178): bit-identical to torch's own indexing, list containers mutated in place."""
    L = 1237
    g = torch.Generator().manual_seed(L)
    if kind == "ids":
        pe = torch.randint(0, 50000, (1, L), generator=g)
    else:
        shape = (3, 1, L, 48) if kind == "mrope" else (1, L, 48)
        pe = [torch.randn(shape, generator=g).to(dtype), torch.randn(shape, generator=g).to(dtype)]
    keep = torch.sort(torch.randperm(L, generator=g)[:400]).values
    mask = torch.zeros(1, L, dtype=torch.bool)
    mask[0, keep] = True
    want = orc.gather_position_embeddings([t.clone() for t in pe] if isinstance(pe, list) else pe.clone(), keep)
    f = ffa.FrameFusion()
    for how in ("pruning", "merging"):
        arg = [t.to(DEV) for t in pe] if isinstance(pe, list) else pe.to(DEV)
        if how == "pruning":
            got = f.position_embedding_handler_at_pruning(arg, keep.to(DEV))
        else:
            got = f.position_embedding_handler_at_merging(arg, mask.to(DEV))
        if isinstance(pe, list):
            assert got is arg                                              # the list is mutated in place (main.py:146-150)
            for a, b in zip(got, want):
                assert a.shape == b.shape and same_bits(a.cpu().contiguous(), b.contiguous()), (how, kind)
        else:
            assert torch.equal(got.cpu(), want), (how, kind)
    # at_pruning follows the index tensor literally (any order, repeats, negative indices)
    odd = torch.tensor([5, 5, L - 1, -1, 0, 17])
    arg = [t.to(DEV) for t in pe] if isinstance(pe, list) else pe.to(DEV)
    got = f.position_embedding_handler_at_pruning(arg, odd.to(DEV))
    want = orc.gather_position_embeddings([t.clone() for t in pe] if isinstance(pe, list) else pe.clone(), odd)
    for a, b in zip(got if isinstance(got, list) else [got], want if isinstance(want, list) else [want]):
        assert torch.equal(a.cpu().float(), b.float())
    # ... and an index outside [-L, L) raises, as torch's indexing does (the kernel used to skip the row: uninitialised output)
    for bad in (torch.tensor([3, L]), torch.tensor([-L - 1, 0])):
        arg = [t.to(DEV) for t in pe] if isinstance(pe, list) else pe.to(DEV)
        with pytest.raises(IndexError):
            f.position_embedding_handler_at_pruning(arg, bad.to(DEV))


@pytest.mark.parametrize("pdtype", [torch.uint8, torch.bool, torch.int16, torch.int32])
def test_position_handlers_take_one_byte_position_tensors(pdtype):
    """a 2-D position tensor of 1-byte elements (odd row size): the reference's `pe[:, keep]` accepts any dtype"""
    L = 515
    g = torch.Generator().manual_seed(3)
    pe = torch.randint(0, 2 if pdtype == torch.bool else 120, (2, L), generator=g).to(pdtype)
    keep = torch.sort(torch.randperm(L, generator=g)[:77]).values
    mask = torch.zeros(1, L, dtype=torch.bool)
    mask[0, keep] = True
    f = ffa.FrameFusion()
    assert torch.equal(f.position_embedding_handler_at_pruning(pe.to(DEV), keep.to(DEV)).cpu(), pe[:, keep])
    assert torch.equal(f.position_embedding_handler_at_merging(pe.to(DEV), mask.to(DEV)).cpu(), pe[:, mask[0]])


@pytest.mark.parametrize("mdtype", [torch.bfloat16, torch.float32, torch.bool, torch.float64])
@pytest.mark.parametrize("L", [97, 640, 1531])
@pytest.mark.parametrize("through_call", [False, True])
def test_attention_mask_is_gathered_like_the_reference(mdtype, L, through_call):
    """main.py:137-138 / 99-100: attention_mask[:, :, keep, :][:, :, :, keep] through the two-level gather kernel, every
    element size, lengths whose output rows are and are not whole 16-byte words; merge call and prune call."""
    F, P, d = (L - 7) // 10, 10, 64
    h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.6, seed=L, pre=3, post=L - 3 - F * P, grid=0.125)
    assert h.shape[1] == L
    g = torch.Generator().manual_seed(L + 1)
    if mdtype == torch.bool:
        mask = torch.rand(1, 1, L, L, generator=g) < 0.5
    else:
        mask = torch.randn(1, 1, L, L, generator=g).to(mdtype)
    want, mo = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P, torch.arange(L)[None], mask.clone(),
                                   layers=3, heads=2, num=1, start=3, n_visual=F * P)
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    # through_call: the mask and an [L, L] capacity buffer travel in the call block and ff_ctx_merge_finish gathers (the C
    # ABI's form for hosts that allocate up front); else FrameFusion.forward's own: an exactly sized buffer, ff_ctx_gather_mask
    f.mask_through_call = through_call
    got, mg = harness.run_cascade(f, dev(h), dev(pt), P, dev(torch.arange(L)[None]), dev(mask),
                                  layers=3, heads=2, num=1, start=3, n_visual=F * P)
    assert [r["length"] for r in got] == [r["length"] for r in want] and got[-1]["finish_pruning"]
    assert mg.shape == mo.shape and mg.dtype == mo.dtype
    assert torch.equal(mg.cpu().to(torch.float64), mo.to(torch.float64))


# ---------------------------------------------------------------------------------------------
# the two documented deviations from the reference, pinned: where exactly the HIP path and the
# oracle (which mirrors the reference) part, and that nothing else does
# ---------------------------------------------------------------------------------------------
def test_deviation_run_lengths_beyond_256_are_exact_here():
    """main.py:269-276 stores run lengths in the activation dtype: a run of 291 folded frames becomes bf16(291) = 292,
    the reference then anchors the run one slot early.  The HIP path derives runs from the member flags (exact at any
    length): it equals the oracle with exact integer run lengths bit for bit, has the reference's keep mask, and
    differs from the stock oracle (which mirrors the reference) only in rows of patches that have a run longer than 256."""
    F, P, d = 300, 4, 64
    h, pt = video_tokens(F, P, d, p_change=0.0, sigma=0.3, seed=11, pre=2, post=2, grid=0.125)
    h = h.clone()
    for fr in range(40, F, 50):                              # patch 3 breaks every 50 frames: no long run there
        h[0, 2 + fr * P + 3] = -h[0, 2 + fr * P + 3]
    L = h.shape[1]

    def run_oracle():
        o = orc.OracleFrameFusion(0.02, 0.6, 0.0)
        o.prepare(pt.clone(), P, 2, 2 + F * P - 1, F * P, L)
        out = o.forward(h.clone(), torch.arange(L)[None], None)
        flags = torch.zeros(F * P, dtype=torch.long)
        flags[o.last_merge_idx] = 1
        return o, out, flags
    o_ref, (h_ref, p_ref, _), flags = run_oracle()
    stock = orc.run_lengths
    lens = stock(flags[None])[0]
    order = o_ref.last_order.reshape(-1)
    long_patches = set(int(pt[0, order[j]]) for j in torch.nonzero(lens > 256).reshape(-1).tolist())
    assert long_patches and 3 not in long_patches            # the case is what it claims to be
    orc.run_lengths = lambda fl: stock(fl.to(torch.long))                # exact integer run lengths
    try:
        o_exact, (h_exact, p_exact, _), _ = run_oracle()
    finally:
        orc.run_lengths = stock
    f = ffa.FrameFusion(0.02, 0.6, 0.0)
    f.prepare(dev(pt), P, 2, 2 + F * P - 1, F * P, L)
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    assert torch.equal(pg.cpu(), p_exact) and torch.equal(pg.cpu(), p_ref)          # the keep mask never depended on the lengths
    assert same_bits(hg.cpu(), h_exact)
    differ = (hg.cpu().view(torch.int16) != h_ref.view(torch.int16)).any(dim=-1)[0]
    assert bool(differ.any())                                                       # the reference really is off here
    differing_patches = set(int(pt[0, i]) for i in pg.cpu()[0][differ].tolist())
    assert differing_patches <= long_patches                                        # ... and only there (text = -1 and patch 3 agree)


def test_deviation_by_patch_slot_zero_never_folds():
    """main.py:290: with a top-k larger than the number of valid pairs the reference selects IGNORE (-2) entries, slot 0
    among them, whose "anchor" index -1 wraps to the LAST by-patch slot.  Only reachable when the threshold itself is
    below -2 (count/ftn >= sub forces k <= count otherwise).  Here slot 0 is never a member: the first by-patch token
    survives, everything else is the reference's."""
    F, P, d = 2, 4, 64
    h, pt = video_tokens(F, P, d, p_change=0.0, sigma=0.3, seed=3, pre=2, post=1, grid=0.125)
    L = h.shape[1]
    o = orc.OracleFrameFusion(0.3, -5.0, 0.0)
    o.prepare(pt.clone(), P, 2, 2 + F * P - 1, F * P, L)
    ho, po, _ = o.forward(h.clone(), torch.arange(L)[None], None)
    assert o.last_merge_idx.tolist()[0] == 0 and o.last_merge_idx.numel() == 5          # k = 5 > 4 valid pairs: slot 0 selected
    f = ffa.FrameFusion(0.3, -5.0, 0.0)
    f.prepare(dev(pt), P, 2, 2 + F * P - 1, F * P, L)
    hg, pg, _ = f(dev(h), dev(torch.arange(L)[None]), None)
    first = int(o.last_order.reshape(-1)[0])                                             # sequence position of by-patch slot 0
    assert f.last_call["branch"] == 1 and f.last_call["k"] == 5
    assert sorted(pg.cpu()[0].tolist()) == sorted(po[0].tolist() + [first])              # one more survivor: that token
    # every row the two have in common is bit-identical except the run the reference folded slot 0 into
    last = int(o.last_order.reshape(-1)[-1])
    for i in po[0].tolist():
        a = hg.cpu()[0, int((pg.cpu()[0] == i).nonzero()[0])]
        b = ho[0, int((po[0] == i).nonzero()[0])]
        anchor_of_last_run = int(o.last_order.reshape(-1)[-2])
        if i == anchor_of_last_run:
            continue
        assert same_bits(a[None, None], b[None, None]), i
    assert last not in po[0].tolist()


@pytest.mark.parametrize("dtype,H,Hk,num,S,d", [(torch.bfloat16, 8, 2, 1, 3000, 128), (torch.float16, 12, 4, 4, 5555, 64),
                                                (torch.bfloat16, 4, 4, 1, 700, 256)])
def test_hook_importance_feeds_the_prune_with_tables(dtype, H, Hk, num, S, d):
    """last_query_importance(..., framefusion=ff) + the prune call that consumes it: the hook's kernel leaves the select
    tables in the instance's workspace (tables_ready = 1).  Compared with the stand-alone route (weights -> head mean ->
    tables -> plan) fed the SAME importance: identical kept sets and rows."""
    dh = 128
    g = torch.Generator().manual_seed(S)
    q = harness.snap(torch.randn(1, H, num, dh, generator=g), dtype).to(DEV)
    k = harness.snap(torch.randn(1, Hk, S, dh, generator=g), dtype).to(DEV)
    h = torch.randn(1, S, d, generator=g).to(dtype).to(DEV)
    start, n_img = 5, S - 12
    if True:
        def fresh():
            f = ffa.FrameFusion(0.3, 0.6, 0.1)
            f.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, n_img, S,
                      finish_merging=True, finish_pruning=False, sparsity_list=[0.4, 0.2])
            return f
        f1 = fresh()
        imp = ffa.last_query_importance(q, k, num=num, is_causal=True, framefusion=f1)
        sc = f1._scratch[("cuda", 0)]
        assert sc.dirty                      # (tables announced: a prune that never comes is cleaned up by the next call)
        out1, pos1, _ = f1(h, torch.arange(S, device=DEV)[None], None, imp)
        assert f1.finish_pruning and not sc.dirty
        # reference route: the same importance as a plain [1, 1, 1, S] tensor (no token: head mean of one row, tables, plan)
        f2 = fresh()
        out2, pos2, _ = f2(h, torch.arange(S, device=DEV)[None], None, imp.clone())
        assert torch.equal(pos1, pos2) and same_bits(out1.cpu(), out2.cpu())
        # a second prefill on the same instance (tables cleared, flags reusable)
        f1.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, n_img, S,
                   finish_merging=True, finish_pruning=False, sparsity_list=[0.4, 0.2])
        imp_b = ffa.last_query_importance(q, k, num=num, is_causal=True, framefusion=f1)
        out3, pos3, _ = f1(h, torch.arange(S, device=DEV)[None], None, imp_b)
        assert torch.equal(pos3, pos1) and same_bits(out3.cpu(), out1.cpu())
        # the budget changes between hook and prune: the token no longer matches, the prune recomputes everything
        f3 = fresh()
        imp_c = ffa.last_query_importance(q, k, num=num, is_causal=True, framefusion=f3)
        f3.sparsity_list = [0.4, 0.3]
        out4, pos4, _ = f3(h, torch.arange(S, device=DEV)[None], None, imp_c)
        f4 = fresh()
        f4.sparsity_list = [0.4, 0.3]
        out5, pos5, _ = f4(h, torch.arange(S, device=DEV)[None], None, imp_c.clone())
        assert torch.equal(pos4, pos5) and same_bits(out4.cpu(), out5.cpu()) and pos4.shape != pos1.shape


@pytest.mark.parametrize("dtype,H,Hk,num,S,d,dh,container,with_res,with_mask", [
    (torch.bfloat16, 8, 2, 1, 3000, 128, 128, "qwen2", False, False),
    (torch.float16, 12, 4, 4, 5555, 64, 64, "mrope", True, False),
    (torch.float32, 4, 4, 2, 900, 96, 32, "ids", False, True),
    (torch.bfloat16, 6, 3, 4, 1300, 256, 24, "qwen2", True, True),          # odd head size: the general importance path
])
def test_prune_from_qk_equals_hook_then_prune(dtype, H, Hk, num, S, d, dh, container, with_res, with_mask):
    """One host call (ff_ctx_prune_from_qk: importance -> plan -> gather) against the two-call form
    (last_query_importance(..., framefusion=ff) then forward): same kept set, same rows, same gathered containers, a clean
    workspace afterwards - for every dtype, both importance paths, every position container, with the residual add and an
    attention mask."""
    g = torch.Generator().manual_seed(S + dh)
    q = harness.snap(torch.randn(1, H, num + 3, dh, generator=g), dtype).to(DEV)
    k = harness.snap(torch.randn(1, Hk, S, dh, generator=g), dtype).to(DEV)
    h = torch.randn(1, S, d, generator=g).to(dtype).to(DEV)
    res = torch.randn(1, S, d, generator=g).to(dtype).to(DEV) if with_res else None
    mask = torch.randn(1, 1, S, S, generator=g).to(dtype).to(DEV) if with_mask else None
    start, n_img = 7, S - 19

    def pos():
        if container == "ids":
            return torch.arange(S, device=DEV)[None]
        return [t.to(DEV) for t in rotary_tables(S, 16, dtype, mrope=(container == "mrope"))]

    def fresh():
        f = ffa.FrameFusion(0.3, 0.6, 0.1)
        f.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, n_img, S,
                  finish_merging=True, finish_pruning=False, sparsity_list=[0.4, 0.2])
        return f
    f1 = fresh()
    imp = ffa.last_query_importance(q, k, num=num, is_causal=True, framefusion=f1)
    o1, p1, m1 = f1(h, pos(), mask, imp, residual=res)
    f2 = fresh()
    o2, p2, m2 = f2.prune_from_qk(h, pos(), mask, q, k, num=num, is_causal=True, residual=res)
    assert f2.finish_pruning and not f2._scratch[("cuda", 0)].dirty
    assert o1.shape == o2.shape and same_bits(o1.cpu(), o2.cpu())
    if container == "ids":
        assert torch.equal(p1, p2)
    else:
        assert all(same_bits(a.cpu(), b.cpu()) for a, b in zip(p1, p2))
    assert (m1 is None) == (m2 is None) and (m1 is None or same_bits(m1.cpu(), m2.cpu()))
    # the instance is reusable: a second prefill through the same scratch (grow-only importance workspace) gives the same bits
    f2.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, n_img, S,
               finish_merging=True, finish_pruning=False, sparsity_list=[0.4, 0.2])
    o3, _, _ = f2(h, pos(), mask, ffa.last_query_importance(q, k, num=num, is_causal=True, framefusion=f2, defer=True), residual=res)
    assert same_bits(o3.cpu(), o2.cpu())
    # the single-crossing form of the C ABI (ff_ctx_prune_from_qk: importance, plan and gather enqueued by ONE call; the Python
    # host launches the importance early instead and uses two)
    f4 = fresh()
    f4.prune_in_one_crossing = True
    o4, p4, m4 = f4.prune_from_qk(h, pos(), mask, q, k, num=num, is_causal=True, residual=res)
    assert same_bits(o4.cpu(), o2.cpu()) and not f4._scratch[("cuda", 0)].dirty
    assert (m4 is None) == (m2 is None) and (m4 is None or same_bits(m4.cpu(), m2.cpu()))
    # a handle for another length is refused before anything is enqueued
    f3 = fresh()
    with pytest.raises(ffa.FrameFusionHipError):
        f3(h[:, :-1], pos(), None, ffa.last_query_importance(q, k, num=num, is_causal=True, defer=True))


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 100), (torch.float16, 36), (torch.float32, 7)])
def test_rows_that_are_not_whole_16_byte_words(dtype, d):
    """hidden_size * itemsize % 16 != 0 (no model of the reference has such rows; its torch path does not care): the host pads
    zero columns up to the next word - neutral for dot products, norms and folds - and cuts them off again.  Whole cascade
    (merge calls + the prune) against the oracle, bit for bit; the static entry points too."""
    F, P, pre, post = 9, 21, 3, 4
    h, pt = video_tokens(F, P, d, p_change=0.4, sigma=0.3, sigma_hi=1.4, seed=d, pre=pre, post=post, dtype=dtype, grid=0.125)
    L = h.shape[1]
    want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P, torch.arange(L)[None], None, 3)
    got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), dev(h), dev(pt), P, dev(torch.arange(L)[None]), None, 3)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert (a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
               (b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"]), a["tag"]
        assert a["hidden"].shape[-1] == d and a["hidden"].is_contiguous() and same_bits(a["hidden"].cpu(), b["hidden"]), a["tag"]
        assert torch.equal(a["pos"].cpu(), b["pos"])
    sim_g, ord_g = ffa.FrameFusion.compute_similarity_and_token_index_by_patch(dev(h), dev(pt), P)
    sim_o, ord_o = orc.pair_similarity(h, pt, P)
    assert torch.equal(ord_g.cpu(), ord_o) and same_bits(sim_g.cpu(), sim_o)
    a, b = h[0, :50], h[0, 50:100]
    assert same_bits(ffa.cosine_similarity(dev(a), dev(b)).cpu(), orc.staged_cosine(a, b))
