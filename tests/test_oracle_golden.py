"""CPU: the oracle (oracle/ff_oracle.py) against the golden vectors captured from the real reference
by oracle/make_golden.py.  This is what pins the oracle wherever the repo travels."""
import numpy as np
import pytest
import torch

from oracle import ff_oracle as orc
from framefusion_amd.synth import rotary_tables
from tests import harness
from tests.conftest import DT, from_bits, same_bits, Golden


def test_run_lengths_kat_and_random(golden):
    g = golden("primitives")
    kat = torch.tensor([[0, 1, 1, 1, 0, 0, 1, 1]])
    assert orc.run_lengths(kat).tolist() == [[0, 0, 0, 3, 0, 0, 0, 2]]      # reference main.py:361-363
    got = orc.run_lengths(torch.from_numpy(g["runs_in"]))
    assert np.array_equal(got.numpy(), g["runs_out"])


def test_budget(golden):
    g = golden("primitives")
    for lst, cost, want in zip(g["budget_lists"], g["budget_costs"], g["budget_vals"]):
        sl = [float(x) for x in str(lst).split(",") if x]
        assert float(orc.budget(sl, float(cost))) == float(want)
    with pytest.raises(ValueError, match="The cost is too small"):
        orc.budget([0] * 10, 0.3)


SIM_CASES = Golden("similarity_merge").cases()


@pytest.mark.parametrize("name", [c for c in SIM_CASES if c != "hand"])
def test_similarity_and_merge(golden, name):
    g = golden("similarity_merge")
    dtype = DT[str(g[f"{name}/dtype"])]
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    P = int(g[f"{name}/patch_num"])
    sim, order = orc.pair_similarity(h, pt, P)
    assert np.array_equal(order[0].numpy(), g[f"{name}/order"])
    want = from_bits(g[f"{name}/sim"], dtype)
    if bool(g[f"{name}/exact"]):
        assert same_bits(sim[0], want)          # dyadic-grid inputs: independent of summation order
    else:
        # gaussian inputs: another CPU may sum in another order; <= 1 ulp on a tiny fraction
        diff = (sim[0].float() - want.float()).abs()
        assert float((diff > 0).float().mean()) <= 0.02
        assert torch.allclose(sim[0].float(), want.float(), rtol=2 ** -7, atol=0)
    golden_sim = want[None]
    for sname in ("thr", "all", "alt", "none"):
        midx = torch.from_numpy(g[f"{name}/merge_{sname}/idx"])
        merged, keep = orc.merge_rows(h, order, midx)
        assert np.array_equal(keep[0].numpy(), g[f"{name}/merge_{sname}/keep"])
        assert same_bits(merged[0], from_bits(g[f"{name}/merge_{sname}/hidden"], dtype))
    del golden_sim


def test_hand_example(golden):
    g = golden("similarity_merge")
    h = torch.from_numpy(g["hand/hidden"])[None]
    pt = torch.from_numpy(g["hand/patch_type"])[None]
    sim, order = orc.pair_similarity(h, pt, 2)
    assert order.tolist() == [[1, 3, 5, 2, 4, 6]]
    assert np.allclose(sim[0].numpy(), g["hand/sim"], rtol=1e-6)


def make_pos(kind, L, dtype):
    if kind == "qwen2":
        return rotary_tables(L, 16, dtype)
    if kind == "mrope":
        return rotary_tables(L, 16, dtype, mrope=True)
    return torch.arange(L)[None] * 3


FWD_CASES = Golden("forward").cases()


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_single_call(golden, name):
    g = golden("forward")
    dtype = DT[str(g[f"{name}/dtype"])]
    F, P, d, pre, post = (int(x) for x in g[f"{name}/meta"])
    cost, thr, lb = (float(x) for x in g[f"{name}/params"])
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    L = h.shape[1]
    ff = orc.OracleFrameFusion(cost, thr, lb)
    ff.prepare(pt.clone(), P, pre, pre + F * P, F * P, L)
    pos = make_pos(str(g[f"{name}/pos_kind"]), L, dtype)
    pos_in = [p.clone() for p in pos] if isinstance(pos, list) else pos.clone()
    mask = None
    if bool(g[f"{name}/mask"]):
        mask = torch.zeros(1, 1, L, L, dtype=dtype).masked_fill_(torch.ones(L, L, dtype=torch.bool).triu(1), float("-inf"))
    out, pos_out, mask_out = ff.forward(h.clone(), pos, None if mask is None else mask.clone())
    keep = torch.from_numpy(g[f"{name}/keep"])
    assert np.array_equal(ff.last_keep.numpy(), keep.numpy())
    assert same_bits(out[0], from_bits(g[f"{name}/hidden_out"], dtype))
    assert np.array_equal(ff.patch_type[0].numpy(), g[f"{name}/patch_type_out"])
    assert [ff.finish_merging, ff.finish_pruning] == [bool(x) for x in g[f"{name}/flags"]]
    assert ff.sparsity_list == list(g[f"{name}/sparsity"])
    if isinstance(pos_in, list):
        for a, b in zip(pos_out, pos_in):
            assert same_bits(a, b.index_select(b.ndim - 2, keep))
    else:
        assert torch.equal(pos_out, pos_in[:, keep])
    if mask is not None:
        assert same_bits(mask_out, mask[:, :, keep][:, :, :, keep])


CAS_CASES = Golden("cascade").cases()


@pytest.mark.parametrize("name", CAS_CASES)
def test_cascade(golden, name):
    g = golden("cascade")
    dtype = DT[str(g[f"{name}/dtype"])]
    F, P, d, pre, post, layers, heads, num = (int(x) for x in g[f"{name}/meta"])
    h = from_bits(g[f"{name}/hidden"], dtype)[None]
    pt = torch.from_numpy(g[f"{name}/patch_type"])[None]
    L = h.shape[1]
    log, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P,
                                 torch.arange(L)[None], None, layers, heads, num)
    assert [r["tag"] for r in log] == [str(t) for t in g[f"{name}/tags"]]
    assert [r["length"] for r in log] == [int(x) for x in g[f"{name}/lengths"]]
    for r, fl, ns in zip(log, g[f"{name}/flags"], g[f"{name}/n_sparsity"]):
        assert [r["finish_merging"], r["finish_pruning"]] == [bool(fl[0]), bool(fl[1])]
        assert len(r["sparsity"]) == int(ns)
        assert same_bits(r["hidden"][0], from_bits(g[f"{name}/{r['tag']}/hidden"], dtype))
        assert np.array_equal(r["pos"][0].numpy(), g[f"{name}/{r['tag']}/index"])
    assert log[-1]["sparsity"] == list(g[f"{name}/sparsity"])


IMP_CASES = Golden("importance").cases()


@pytest.mark.parametrize("name", IMP_CASES)
def test_importance(golden, name):
    g = golden("importance")
    dtype = DT[str(g[f"{name}/dtype"])]
    H, Hk, S, dh, num, causal = (int(x) for x in g[f"{name}/meta"])
    q = from_bits(g[f"{name}/q"], dtype)[None]
    k = from_bits(g[f"{name}/k"], dtype)[None]
    w = orc.last_query_attention(q, k, num=num, is_causal=bool(causal), enable_gqa=H != Hk)
    want = from_bits(g[f"{name}/weights"], dtype)
    # exp / matmul summation order may differ by CPU: allow 1 ulp on a small fraction
    assert torch.allclose(w[0].float(), want.float(), rtol=2 ** -7, atol=1e-30)
    assert float((w[0].float() != want.float()).float().mean()) <= 0.02
    imp = torch.mean(w, dim=(1, 2))[0]
    assert torch.allclose(imp.float(), from_bits(g[f"{name}/importance"], dtype).float(), rtol=2 ** -7, atol=1e-30)


MASK_CASES = Golden("importance_mask").cases()


def load_mask(g, name):
    kind = str(g[f"{name}/mask_dtype"])
    raw = g[f"{name}/mask"]
    if kind == "bool":
        return torch.from_numpy(raw.copy())
    if kind == "fp32":
        return torch.from_numpy(raw.copy())
    return from_bits(raw, DT[kind])


@pytest.mark.parametrize("name", MASK_CASES)
def test_importance_with_attn_mask(golden, name):
    """utils.py:40-44 (boolean / additive attn_mask): the oracle against what the real reference produced
    (oracle/make_golden_mask.py)."""
    g = golden("importance_mask")
    dtype = DT[str(g[f"{name}/dtype"])]
    H, Hk, S, dh, num = (int(x) for x in g[f"{name}/meta"])
    q = from_bits(g[f"{name}/q"], dtype)[None]
    k = from_bits(g[f"{name}/k"], dtype)[None]
    w = orc.last_query_attention(q, k, num=num, enable_gqa=H != Hk, attn_mask=load_mask(g, name))
    want = from_bits(g[f"{name}/weights"], dtype)
    assert torch.allclose(w[0].float(), want.float(), rtol=2 ** -7, atol=1e-30)
    assert float((w[0].float() != want.float()).float().mean()) <= 0.02
    with pytest.raises(AssertionError):                                      # utils.py:35
        orc.last_query_attention(q, k, num=num, is_causal=True, attn_mask=load_mask(g, name))


def test_topk_tie_rule():
    x = torch.tensor([.5, .75, .75, .75, .25, .75])
    assert orc.topk_lowest_index(x, 3).tolist() == [1, 2, 3]
    assert orc.topk_lowest_index(x, 0).tolist() == []
    y = torch.tensor([1.0, float("nan"), 3.0, 2.0])
    assert orc.topk_lowest_index(y, 2).tolist() == [1, 2]      # NaN ranks highest, like torch.topk


def test_edge_cases():
    f = orc.OracleFrameFusion()
    f.prepare(torch.tensor([[-1]]), 4, 0, 0, 0, 1)
    tok = torch.zeros(1, 1, 8)
    r = f.forward(tok, "pos", "mask")
    assert r[0] is tok and r[1] == "pos" and r[2] == "mask"                # q_len == 1: untouched
    f.prepare(torch.tensor([[-1, -1, -1]]), 4, 0, 0, 0, 3)
    with pytest.raises(AssertionError):                                      # no visual tokens, main.py:240
        f.forward(torch.zeros(1, 3, 8), [torch.zeros(1, 3, 2)] * 2, None)
    g = orc.OracleFrameFusion()
    with pytest.raises(AttributeError):                                      # never prepared
        g.forward(torch.zeros(1, 3, 8), None, None)
