"""GPU: the whole hook protocol on a tiny Qwen2-style decoder patched with apply_framefusion, with a
shadow CPU oracle checking every FrameFusion.forward call on the activations the model really
produces (gaussian-like, not on a grid)."""
import numpy as np
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens
from oracle import ff_oracle as orc
from tests import tiny_decoder as td

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Shadow:
    """Wraps the HIP FrameFusion.forward: replays each call on the CPU oracle (same inputs, same
    state) and records the comparison."""

    def __init__(self, ff):
        self.ff, self.inner, self.log = ff, ff.forward, []

    def __call__(self, hidden, pos, mask, attn_w=None, residual=None):
        ff = self.ff
        if hidden.shape[1] <= 1:
            return self.inner(hidden, pos, mask, attn_w, residual)
        fused_in = hidden
        if residual is not None:                # the oracle sees what the reference would: the eager sum
            hidden = residual + hidden
        o = orc.OracleFrameFusion(ff.cost, ff.similarity_lower_bound, ff.ratio_lower_bound)
        o.prepare(ff.patch_type.cpu(), ff.patch_num, ff.image_token_start_index, ff.image_token_end_index,
                  ff.image_token_length, ff.original_length, ff.finish_merging, ff.finish_pruning, list(ff.sparsity_list))
        L = hidden.shape[1]
        # (the adapters hand the prune a LastQuery handle - q_last + K - instead of weights: the oracle gets the weights it stands for)
        w_ref = attn_w.weights() if isinstance(attn_w, ffa.utils.LastQuery) else attn_w
        ho, po, _ = o.forward(hidden.cpu(), torch.arange(L)[None], None, None if w_ref is None else w_ref.cpu())
        was_active = (not ff.finish_merging) or (not ff.finish_pruning)
        out = self.inner(fused_in, pos, mask, attn_w, residual)
        if was_active and (ff.last_call is not None):
            kind = ff.last_call["kind"]
            keep_g = torch.nonzero(ff.last_plan()["keep"][:L].bool()).reshape(-1).cpu().numpy()
            keep_o = po[0].numpy()
            sym = np.setxor1d(keep_g, keep_o).size
            common, ig, io = np.intersect1d(keep_g, keep_o, return_indices=True)
            a, b = out[0][0].float().cpu()[ig], ho[0].float()[io]
            rel = float(((a - b).abs() / b.abs().clamp_min(1e-2)).max()) if len(ig) else 0.0
            self.log.append(dict(kind=kind, L_in=L, L_out=out[0].shape[1], L_oracle=ho.shape[1], sym=sym, rel=rel,
                                 flags=(ff.finish_merging, ff.finish_pruning) == (o.finish_merging, o.finish_pruning)))
        return out


@pytest.mark.parametrize("dtype,p_change", [(torch.bfloat16, 0.5), (torch.bfloat16, 0.95), (torch.float32, 0.3)])
def test_tiny_decoder_prefill_with_shadow_oracle(dtype, p_change):
    torch.manual_seed(0)
    td.register()
    F_, P, d, pre, post = 12, 24, 256, 5, 7
    model = td.TinyVLM(d=d, heads=8, kv_heads=2, layers=4).to(DEV).to(dtype)
    ffa.apply_framefusion(model, cost=0.3, similarity_lower_bound=0.6, ratio_lower_bound=0.1)
    ff = model.framefusion
    assert all(layer.framefusion is ff and layer.self_attn.framefusion is ff for layer in model.model.layers)
    shadow = Shadow(ff)
    ff.forward = shadow                     # nn.Module.__call__ dispatches to .forward
    emb, pt = video_tokens(F_, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.2, seed=11, pre=pre, post=post, dtype=dtype)
    L = emb.shape[1]
    model.prepare_visual(pt.to(DEV), P, pre, F_ * P, L)
    with torch.no_grad():
        out, lengths = model.model(emb.to(DEV))
        dense = td.TinyVLM(d=d, heads=8, kv_heads=2, layers=4)        # sanity: the unpatched stack keeps its length
    assert out.shape[1] == lengths[-1] < L and torch.isfinite(out.float()).all()
    assert ff.finish_merging and ff.finish_pruning                     # merged, then pruned once
    kinds = [r["kind"] for r in shadow.log]
    assert "merge" in kinds and kinds.count("prune") == 1
    assert lengths == sorted(lengths, reverse=True)                     # the sequence only ever shrinks
    for r in shadow.log:
        assert r["flags"], r
        assert r["L_out"] == r["L_oracle"] or r["sym"] <= 2, r          # kept sets agree (ulp flips allowed)
        assert r["sym"] <= 2, r
        assert r["rel"] <= (2e-2 if dtype == torch.bfloat16 else 1e-4), r
    # the budget was honoured: sum of per-layer lengths <= cost * layers * L (28-layer constant aside)
    assert lengths[-1] <= 0.6 * L
    # decode step: q_len == 1 is a no-op
    tok = torch.zeros(1, 1, d, dtype=dtype, device=DEV)
    assert ff(tok, "pos", None)[0] is tok


def test_hf_qwen2_prefill_and_decode():
    """The installed transformers' Qwen2ForCausalLM (random weights) patched with apply_framefusion:
    prefill shrinks the sequence layer by layer, per-layer KV lengths follow, decode is untouched."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from transformers.cache_utils import DynamicCache
    from framefusion_amd.models.qwen2 import register_hf_qwen2
    torch.manual_seed(0)
    cfg = Qwen2Config(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
                      num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=4096)
    cfg._attn_implementation = "sdpa"
    model = Qwen2ForCausalLM(cfg).to(DEV).to(torch.bfloat16).eval()
    register_hf_qwen2()
    ffa.apply_framefusion(model, cost=0.3, similarity_lower_bound=0.6, ratio_lower_bound=0.1)
    ff = model.framefusion
    shadow = Shadow(ff)
    ff.forward = shadow
    F_, P, pre, post = 12, 24, 5, 7
    emb, pt = video_tokens(F_, P, 256, p_change=0.5, sigma=0.3, sigma_hi=1.2, seed=11, pre=pre, post=post)
    L = emb.shape[1]
    ff.prepare(pt.to(DEV), P, pre, pre + F_ * P, F_ * P, L)
    cache = DynamicCache(config=cfg)
    with torch.no_grad():
        out = model.model(inputs_embeds=emb.to(DEV), past_key_values=cache, use_cache=True)
    lengths = model.model.framefusion_lengths
    assert out.last_hidden_state.shape[1] == lengths[-1] < L
    assert lengths == sorted(lengths, reverse=True) and ff.finish_merging and ff.finish_pruning
    kv = [cache.layers[i].keys.shape[2] for i in range(4)]
    assert kv[0] <= L and kv == sorted(kv, reverse=True) and kv[-1] >= lengths[-1]   # per-layer KV lengths differ
    assert [r["kind"] for r in shadow.log].count("prune") == 1
    for r in shadow.log:
        assert r["flags"] and r["sym"] <= 2, r
    # decode step: one new token, FrameFusion is a no-op, the cache keeps its per-layer lengths + 1
    with torch.no_grad():
        step = model.model(inputs_embeds=torch.zeros(1, 1, 256, dtype=torch.bfloat16, device=DEV),
                           past_key_values=cache, use_cache=True,
                           position_ids=torch.tensor([[L]], device=DEV))
    assert step.last_hidden_state.shape[1] == 1
    assert [cache.layers[i].keys.shape[2] for i in range(4)] == [n + 1 for n in kv]


def test_hf_qwen2_vl_text_prefill():
    """transformers' Qwen2VLTextModel (random weights, M-RoPE [3, 1, L, dh] position embeddings,
    num=4 importance queries) patched with apply_framefusion."""
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextModel
    from transformers.cache_utils import DynamicCache
    from framefusion_amd.models.qwen2_vl import register_hf_qwen2_vl
    torch.manual_seed(0)
    cfg = Qwen2VLTextConfig(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
                            num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=4096,
                            rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [4, 6, 6]})
    cfg._attn_implementation = "sdpa"

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Qwen2VLTextModel(cfg)
    m = Holder().to(DEV).to(torch.bfloat16).eval()
    register_hf_qwen2_vl()
    ffa.apply_framefusion(m, cost=0.3, similarity_lower_bound=0.5, ratio_lower_bound=0.1)
    ff = m.framefusion
    shadow = Shadow(ff)
    ff.forward = shadow
    F_, P, pre, post = 12, 24, 5, 7
    emb, pt = video_tokens(F_, P, 256, p_change=0.5, sigma=0.3, sigma_hi=1.2, seed=12, pre=pre, post=post)
    L = emb.shape[1]
    ff.prepare(pt.to(DEV), P, pre, pre + F_ * P, F_ * P, L)
    with torch.no_grad():
        out = m.model(inputs_embeds=emb.to(DEV), past_key_values=DynamicCache(config=cfg), use_cache=True)
    lengths = m.model.framefusion_lengths
    assert out.last_hidden_state.shape[1] == lengths[-1] < L
    assert lengths == sorted(lengths, reverse=True) and ff.finish_merging and ff.finish_pruning
    assert [r["kind"] for r in shadow.log].count("prune") == 1
    for r in shadow.log:
        assert r["flags"] and r["sym"] <= 2, r


def test_hf_qwen2_vl_full_model_builds_its_own_layout():
    """transformers' Qwen2VLForConditionalGeneration with a (tiny, random) vision tower: the packer
    hook derives patch_type from input_ids + video_grid_thw (qwenvl/modeling_qwen2_vl.py:117-138)
    and calls prepare; nobody prepares FrameFusion by hand."""
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    from transformers.cache_utils import DynamicCache
    from framefusion_amd.models.qwen2_vl import register_hf_qwen2_vl
    from oracle import layout_oracle as lay
    torch.manual_seed(0)
    VID = 151
    cfg = Qwen2VLConfig(
        text_config=dict(vocab_size=160, hidden_size=256, intermediate_size=512, num_hidden_layers=4,
                         num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=4096,
                         bos_token_id=None, eos_token_id=None,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [4, 6, 6]}),
        vision_config=dict(depth=1, embed_dim=32, hidden_size=256, num_heads=2, mlp_ratio=2, in_channels=3,
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
        image_token_id=150, video_token_id=VID, vision_start_token_id=152, vision_end_token_id=153)
    cfg._attn_implementation = "sdpa"
    m = Qwen2VLForConditionalGeneration(cfg).to(DEV).to(torch.bfloat16).eval()
    register_hf_qwen2_vl()
    ffa.apply_framefusion(m, cost=0.3, similarity_lower_bound=0.5, ratio_lower_bound=0.1)
    ff = m.framefusion
    assert m.model.language_model.framefusion is ff
    shadow = Shadow(ff)
    ff.forward = shadow
    T, H, W = 12, 8, 12                                    # 12 temporal grids of 8x12 patches -> P = 24
    n = T * H * W // 4
    ids = [1, 2, 3, 4, 152] + [VID] * n + [153, 5, 6, 7, 8, 9, 10]
    frame = torch.randn(H * W, 3 * 2 * 14 * 14)
    frames = [frame]
    for t in range(1, T):                                   # slowly drifting frames, two scene cuts
        frames.append(torch.randn_like(frame) if t in (5, 9) else frames[-1] + 0.05 * torch.randn_like(frame))
    pixels = torch.cat(frames).to(DEV).to(torch.bfloat16)
    input_ids = torch.tensor([ids], device=DEV)
    with torch.no_grad():
        out = m.model(input_ids=input_ids, pixel_values_videos=pixels, video_grid_thw=torch.tensor([[T, H, W]], device=DEV),
                      mm_token_type_ids=(input_ids == VID).int() * 2,
                      past_key_values=DynamicCache(config=cfg.text_config), use_cache=True)
    want = lay.qwen2_vl(ids, VID, H, W, 2, n, len(ids))
    assert (ff.patch_num, int(ff.image_token_start_index), int(ff.image_token_end_index), int(ff.image_token_length),
            ff.original_length) == tuple(want[1:])
    assert ff.patch_num == 24 and ff.original_length == len(ids)
    lengths = m.model.language_model.framefusion_lengths
    assert out.last_hidden_state.shape[1] == lengths[-1] < len(ids)
    assert lengths == sorted(lengths, reverse=True) and ff.finish_merging and ff.finish_pruning
    kinds = [r["kind"] for r in shadow.log]
    assert "merge" in kinds and kinds.count("prune") <= 1          # the top-k branch ends both phases at once
    for r in shadow.log:
        assert r["flags"] and r["sym"] <= 2, r
    # a second prefill re-derives the layout (state reset by prepare, main.py:27-38) ...
    with torch.no_grad():
        m.model(input_ids=input_ids, pixel_values_videos=pixels, video_grid_thw=torch.tensor([[T, H, W]], device=DEV),
                mm_token_type_ids=(input_ids == VID).int() * 2,
                past_key_values=DynamicCache(config=cfg.text_config), use_cache=True)
    assert m.model.language_model.framefusion_lengths == lengths
    # ... and the first one left the full-length row behind for inspection
    first_row = lay.qwen2_vl(ids, VID, H, W, 2, n, len(ids))[0]
    from framefusion_amd.layout import qwen2_vl_layout
    assert qwen2_vl_layout(input_ids, VID, (T, H, W), 2).patch_type[0].tolist() == first_row
