"""Fixed-sparsity merging baseline (SURVEY.md §8(f) rank 4;
framefusion/models/qwen2/modeling_qwen2_baseline.py:26-43, 860-1203).

tests/golden/baseline.npz holds what the reference's own merging block produced on seeded inputs
(oracle/make_golden_baseline.py).  CPU: the oracle restatement against those vectors.  GPU: the HIP
path (ff_ctx_merge_begin + ff_ctx_merge_finish with force_k and FF_FOLD_MEAN, through framefusion_amd.baseline)
against the vectors, against the oracle over a multi-layer schedule, at full size, and inside a
random-weight transformers Qwen2.
"""
import math

import numpy as np
import pytest
import torch

from framefusion_amd.synth import video_tokens, rotary_tables
from oracle import ff_oracle as orc
from tests.conftest import DT, Golden, from_bits, same_bits

DEV = "cuda:0"
CASES = ["bf16_grid_s10", "bf16_grid_s45", "bf16_gauss_s30", "fp32_gauss_s25", "fp16_grid_s20", "bf16_tiny_s0"]


def case_inputs(g, name):
    F, P, d, pre, post, seed, grid, k = (int(v) for v in g[f"{name}/cfg"])
    pc, sp = (float(v) for v in g[f"{name}/fcfg"])
    dt = DT[name.split("_")[0]]
    hidden, ptype = video_tokens(F, P, d, p_change=pc, sigma=0.3, seed=seed, pre=pre, post=post, dtype=dt,
                                 grid=0.125 if grid else None)
    return hidden, ptype, P, sp, k, dt, bool(grid)


@pytest.mark.parametrize("name", CASES)
def test_oracle_baseline_matches_reference_vectors(name):
    g = Golden("baseline")
    hidden, ptype, P, sp, k, dt, _ = case_inputs(g, name)
    o = orc.fixed_sparsity_merge(hidden, ptype, P, sp)
    assert o["prune_num"] == k
    assert same_bits(o["hidden"], from_bits(g[f"{name}/hidden"], dt))
    assert torch.equal(o["patch_type"], torch.from_numpy(g[f"{name}/patch_type"]))
    mask = torch.from_numpy(g[f"{name}/mask"])
    if k:
        assert torch.equal(o["token_mask"], mask)
        assert same_bits(o["sim"], from_bits(g[f"{name}/sim"], dt))
    else:
        assert o["token_mask"] is None and bool(mask.all())


def test_oracle_density_overhead_matches_reference_vectors():
    g = Golden("baseline")
    i = 0
    while g.has(f"density/{i}/in"):
        got = orc.density_overhead(g[f"density/{i}/in"].tolist())
        assert got == tuple(g[f"density/{i}/out"].tolist())
        i += 1
    assert i == 4


def test_density_overhead_host_function():
    from framefusion_amd.baseline import compute_density_overhead
    g = Golden("baseline")
    for i in range(4):
        assert compute_density_overhead(g[f"density/{i}/in"].tolist()) == tuple(g[f"density/{i}/out"].tolist())


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def hip_merger(sparsity, ptype, P):
    from framefusion_amd.baseline import FixedSparsityMerging
    m = FixedSparsityMerging(sparsity)
    m.prepare(ptype.to(DEV), P)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_baseline_matches_reference_vectors(name):
    g = Golden("baseline")
    hidden, ptype, P, sp, k, dt, grid = case_inputs(g, name)
    m = hip_merger([sp], ptype, P)
    out, mask, _ = m.merge(0, hidden.to(DEV))
    want_h = from_bits(g[f"{name}/hidden"], dt)
    want_mask = torch.from_numpy(g[f"{name}/mask"])
    if k == 0:
        assert mask is None and same_bits(out.cpu(), want_h)
        return
    assert torch.equal(mask.cpu(), want_mask)                       # unique top-k cut: index-exact
    assert torch.equal(m.patch_type.cpu(), torch.from_numpy(g[f"{name}/patch_type"]))
    if grid:
        assert same_bits(out.cpu(), want_h)                          # sums exact in fp32: bit-exact
        assert same_bits(m.last_plan()["sim"].cpu()[None], from_bits(g[f"{name}/sim"], dt))
    else:
        tol = 1e-3 if dt != torch.float32 else 1e-5                  # north-star tolerance (fp32 sum order)
        torch.testing.assert_close(out.cpu().float(), want_h.float(), rtol=tol, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_hip_baseline_schedule_vs_oracle(dt):
    """Five layers of the schedule on the same evolving tensors: normed activations, residual, cos/sin
    and patch_type compacted by every layer (order maintained on the device from layer to layer)."""
    F, P, d, pre, post = 16, 24, 128, 5, 7
    sparsity = [0.10, 0.0, 0.25, 0.05, 0.30]
    seed = 300           # ties at a cut are fine here: oracle and HIP share the lowest-index rule
    hidden, ptype = video_tokens(F, P, d, p_change=0.3, sigma=0.3, seed=seed, pre=pre, post=post, dtype=dt, grid=0.125)
    residual, _ = video_tokens(F, P, d, p_change=0.9, sigma=0.3, seed=seed + 1000, pre=pre, post=post, dtype=dt, grid=0.125)
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, 32, dt)
    h, r, pt, pos = hidden, residual, ptype, [cos, sin]
    trace = []
    for layer, sp in enumerate(sparsity):
        o = orc.fixed_sparsity_merge(h, pt, P, sp, pos, r)
        h, r, pt, pos = o["hidden"], o["residual"], o["patch_type"], o["position_embeddings"]
        trace.append(o)
    m = hip_merger(sparsity, ptype, P)
    h, r, pos = hidden.to(DEV), residual.to(DEV), [cos.to(DEV), sin.to(DEV)]
    for layer, o in enumerate(trace):
        h, mask, r = m.merge(layer, h, pos, r)
        if o["token_mask"] is None:
            assert mask is None
        else:
            assert torch.equal(mask.cpu(), o["token_mask"]), layer
        assert same_bits(h.cpu(), o["hidden"]), layer
        assert same_bits(r.cpu(), o["residual"]), layer
        assert torch.equal(m.patch_type.cpu(), o["patch_type"]), layer
        assert same_bits(pos[0].cpu(), o["position_embeddings"][0]) and same_bits(pos[1].cpu(), o["position_embeddings"][1])
    assert m.last_call["L_out"] == trace[-1]["hidden"].shape[1] < hidden.shape[1]
    # a new prefill starts from the full row again
    m.prepare(ptype.to(DEV), P)
    h2, mask2, _ = m.merge(0, hidden.to(DEV))
    assert torch.equal(mask2.cpu(), trace[0]["token_mask"])


@pytest.mark.gpu
def test_hip_baseline_full_size():
    """64 x 576 x 4096 bf16 (BASELINE.json configs[1]), sparsity 0.3: size-independent properties
    on gaussian data + exact comparison with the oracle on a row sample."""
    F, P, d = 64, 576, 4096
    hidden, ptype = video_tokens(F, P, d, p_change=0.3, sigma=0.3, seed=77, dtype=torch.bfloat16)
    L = hidden.shape[1]
    k = math.floor(0.3 * L)
    m = hip_merger([0.3], ptype, P)
    hd = hidden.to(DEV)
    out, mask, res = m.merge(0, hd, None, hd)
    assert mask.shape == (1, L) and int((~mask).sum()) == k and out.shape[1] == L - k == res.shape[1]
    assert torch.equal(res, hd[:, mask[0]])                          # the residual is only compacted
    plan = m.last_plan()
    sim, order, member = plan["sim"].float().cpu(), plan["order"].long().cpu(), plan["member"].bool().cpu()[:L]
    # the k selected similarities are the k largest (ties aside): min(selected) >= max(unselected)
    assert int(member.sum()) == k and float(sim[member].min()) >= float(sim[~member].max())
    assert not bool(member[0])
    # runs: compare some anchors with the oracle's fp32 mean of the same rows
    anchors = torch.nonzero(~member[:-1] & member[1:]).reshape(-1)[:: max(1, k // 400)]
    dstmap = (torch.cumsum(mask[0].cpu().long(), 0) - 1)
    outc = out[0].cpu()
    for a in anchors.tolist():
        n = 1
        while a + n < L and member[a + n]:
            n += 1
        rows = hidden[0, order[a:a + n]]
        want = rows.mean(dim=0)
        got = outc[dstmap[order[a]]]
        torch.testing.assert_close(got.float(), want.float(), rtol=1e-3, atol=1e-3)
    # unmerged tokens pass through untouched
    single = torch.nonzero(~member & ~torch.cat((member[1:], torch.tensor([False])))).reshape(-1)[::97]
    assert torch.equal(outc[dstmap[order[single]]], hidden[0, order[single]])


@pytest.mark.gpu
def test_hip_baseline_edges():
    from framefusion_amd.baseline import FixedSparsityMerging
    from framefusion_amd._lib import FrameFusionHipError
    hidden, ptype = video_tokens(4, 4, 32, seed=3, pre=1, post=1, dtype=torch.bfloat16, grid=0.125)
    m = FixedSparsityMerging([0.5])
    with pytest.raises(AttributeError):
        m.merge(0, hidden.to(DEV))                                    # never prepared
    m.prepare(ptype.to(DEV), 4)
    with pytest.raises(FrameFusionHipError):
        m.merge(0, hidden)                                            # CPU tensor: no eager fallback
    one = hidden[:, :1].to(DEV)
    assert m.merge(0, one)[0] is one                                  # decode step: untouched
    with pytest.raises(FrameFusionHipError):
        m.merge(0, hidden[:, :-1].to(DEV))                            # patch_type length mismatch
    with pytest.raises(NotImplementedError):
        m.merge(0, hidden.to(DEV), tuple(rotary_tables(hidden.shape[1], 8, torch.bfloat16)))
    out, mask, _ = m.merge(0, hidden.to(DEV))
    o = orc.fixed_sparsity_merge(hidden, ptype, 4, 0.5)
    assert out.shape == o["hidden"].shape and int((~mask).sum()) == o["prune_num"] == 8


@pytest.mark.gpu
def test_hf_qwen2_with_merging_baseline():
    """transformers' Qwen2ForCausalLM (random weights) patched with replace_qwen2_merging: every
    layer's merge is replayed on the CPU oracle with the tensors the model really produced."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from transformers.cache_utils import DynamicCache
    from framefusion_amd.models.qwen2_merging import replace_qwen2_merging
    torch.manual_seed(0)
    cfg = Qwen2Config(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
                      num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=4096)
    cfg._attn_implementation = "sdpa"
    model = Qwen2ForCausalLM(cfg).to(DEV).to(torch.bfloat16).eval()
    sparsity = [0.2, 0.1, 0.0, 0.15]
    merger = replace_qwen2_merging(model, sparsity)
    log = []
    inner = merger.merge

    def shadow(layer_idx, hidden, pos=None, residual=None):
        o = orc.fixed_sparsity_merge(hidden.cpu(), merger.patch_type.cpu(), merger.patch_num, sparsity[layer_idx])
        out = inner(layer_idx, hidden, pos, residual)
        if o["token_mask"] is not None:
            sym = int((out[1].cpu() != o["token_mask"]).sum())
            both = out[1][0].cpu() & o["token_mask"][0]
            a = out[0][0].float().cpu()[both[out[1][0].cpu()]]
            b = o["hidden"][0].float()[both[o["token_mask"][0]]]
            log.append(dict(layer=layer_idx, sym=sym, rel=float(((a - b).abs() / b.abs().clamp_min(1e-2)).max())))
        return out
    merger.merge = shadow
    F_, P, pre, post = 12, 24, 5, 7
    emb, pt = video_tokens(F_, P, 256, p_change=0.5, sigma=0.3, sigma_hi=1.2, seed=12, pre=pre, post=post)
    L = emb.shape[1]
    merger.prepare(pt.to(DEV), P)
    cache = DynamicCache(config=cfg)
    with torch.no_grad():
        out = model.model(inputs_embeds=emb.to(DEV), past_key_values=cache, use_cache=True)
    lengths = model.model.merging_lengths
    ftn, want = F_ * P, []
    cur = L
    for sp in sparsity:
        k = math.floor(sp * ftn)
        ftn -= k
        cur -= k
        want.append(cur)
    assert lengths == want and out.last_hidden_state.shape[1] == want[-1]
    assert [cache.layers[i].keys.shape[2] for i in range(4)] == want
    assert len(log) == 3 and all(r["sym"] <= 2 and r["rel"] <= 2e-2 for r in log), log
    with torch.no_grad():                                              # decode continues on the ragged cache
        step = model.model(input_ids=torch.tensor([[3]], device=DEV), past_key_values=cache, use_cache=True,
                           position_ids=torch.tensor([[L]], device=DEV))
    assert step.last_hidden_state.shape[1] == 1
    assert [cache.layers[i].keys.shape[2] for i in range(4)] == [n + 1 for n in want]
