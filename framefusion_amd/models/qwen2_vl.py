"""Qwen2-VL text decoder for transformers 5.x.

Same protocol as the reference's ``framefusion/models/qwen2/modeling_qwen2_vl.py`` (decoder
:143-220, calls :178-179 and :200; attention :223-317 with ``num=4`` importance queries, :296; model
loop :16-140): M-RoPE position embeddings are a mutable ``[cos, sin]`` list of ``[3, 1, L, dh]``
tensors, gathered along the token axis by FrameFusion (main.py:145-147,165-167).

The packer block that builds ``patch_type`` (``framefusion/models/qwenvl/modeling_qwen2_vl.py:117-138``) sits in
the reference inside a re-written ``Qwen2VLForConditionalGeneration.forward``; here it is a forward
pre-hook on transformers' ``Qwen2VLModel`` (the module that owns the vision tower and receives
``input_ids`` + ``video_grid_thw``): on a prefill with a video it derives the layout with
``framefusion_amd.layout.qwen2_vl_layout`` and calls ``framefusion.prepare``.  Bare text-decoder
wrappers have no such inputs: their callers invoke ``model.framefusion.prepare(...)`` themselves.
"""
from __future__ import annotations

import torch

from ..interface import Family, register_family
from ..main import call_b_with_residual
from ..utils import last_query_importance

NUM_IMPORTANCE_QUERIES = 4           # modeling_qwen2_vl.py:296


def qwen2vl_attention_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                              output_attentions=False, use_cache=False, position_embeddings=None, **kwargs):
    from transformers.models.qwen2_vl.modeling_qwen2_vl import (ALL_ATTENTION_FUNCTIONS, apply_multimodal_rotary_pos_emb,
                                                                 eager_attention_forward)
    bsz, q_len, _ = hidden_states.size()
    query_states = self.q_proj(hidden_states).view(bsz, q_len, -1, self.head_dim).transpose(1, 2)
    key_states = self.k_proj(hidden_states).view(bsz, q_len, -1, self.head_dim).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(bsz, q_len, -1, self.head_dim).transpose(1, 2)
    cos, sin = position_embeddings
    query_states, key_states = apply_multimodal_rotary_pos_emb(query_states, key_states, cos, sin,
                                                               self.config.rope_parameters["mrope_section"])
    if past_key_values is not None:
        key_states, value_states = past_key_values.update(key_states, value_states, self.layer_idx)

    importance = None
    ff = self.framefusion
    if q_len > 1 and ff.finish_merging and not ff.finish_pruning:            # modeling_qwen2_vl.py:291
        importance = last_query_importance(query_states, key_states, num=NUM_IMPORTANCE_QUERIES,
                                           is_causal=attention_mask is None, scale=self.scaling, framefusion=ff, defer=True)

    attention_interface = ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, eager_attention_forward)
    attn_output, _ = attention_interface(self, query_states, key_states, value_states, attention_mask,
                                         dropout=0.0 if not self.training else self.attention_dropout,
                                         scaling=self.scaling, sliding_window=self.sliding_window,
                                         position_ids=position_ids, **kwargs)
    attn_output = self.o_proj(attn_output.reshape(bsz, q_len, -1).contiguous())
    return attn_output, importance


def qwen2vl_decoder_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                            use_cache=False, position_embeddings=None, **kwargs):
    if self.self_attn.layer_idx == 0:                                        # modeling_qwen2_vl.py:178-179
        hidden_states, position_embeddings, attention_mask = self.framefusion(
            hidden_states, position_embeddings, attention_mask)
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)
    hidden_states, importance = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask,
                                               position_ids=position_ids, past_key_values=past_key_values,
                                               use_cache=use_cache, position_embeddings=position_embeddings, **kwargs)
    hidden_states, position_embeddings, attention_mask = call_b_with_residual(    # modeling_qwen2_vl.py:197-200
        self.framefusion, residual, hidden_states, position_embeddings, attention_mask, importance)
    residual = hidden_states
    hidden_states = self.mlp(self.post_attention_layernorm(hidden_states))
    hidden_states = residual + hidden_states
    return hidden_states, position_embeddings, attention_mask                 # modeling_qwen2_vl.py:218


def qwen2vl_text_model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                               inputs_embeds=None, use_cache=None, **kwargs):
    from transformers.cache_utils import DynamicCache
    from transformers.masking_utils import create_causal_mask
    from transformers.modeling_outputs import BaseModelOutputWithPast
    if (input_ids is None) ^ (inputs_embeds is not None):
        raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    if use_cache and past_key_values is None:
        past_key_values = DynamicCache(config=self.config)
    if position_ids is None:
        seen = past_key_values.get_seq_length() if past_key_values is not None else 0
        position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + seen)
        position_ids = position_ids.view(1, 1, -1).expand(3, inputs_embeds.shape[0], -1)
    elif position_ids.ndim == 2:
        position_ids = position_ids[None, ...].expand(3, position_ids.shape[0], -1)
    text_position_ids = None
    if position_ids.ndim == 3 and position_ids.shape[0] == 4:
        text_position_ids, position_ids = position_ids[0], position_ids[1:]
    causal_mask = create_causal_mask(config=self.config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                     past_key_values=past_key_values, position_ids=text_position_ids)
    hidden_states = inputs_embeds
    position_embeddings = list(self.rotary_emb(hidden_states, position_ids))  # [cos, sin], each [3, 1, L, dh]
    lengths = []
    for decoder_layer in self.layers:
        hidden_states, position_embeddings, causal_mask = decoder_layer(
            hidden_states, attention_mask=causal_mask, position_embeddings=position_embeddings,
            position_ids=None, past_key_values=past_key_values, use_cache=use_cache, **kwargs)
        lengths.append(hidden_states.shape[1])
    self.framefusion_lengths = lengths
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=past_key_values)


def attach_qwen2vl_prepare(model) -> None:
    """Forward pre-hook on `model.model` (Qwen2VLModel): qwenvl/modeling_qwen2_vl.py:117-138."""
    from ..layout import qwen2_vl_layout
    inner = model.model
    old = getattr(inner, "_framefusion_prepare_handle", None)
    if old is not None:
        old.remove()

    def prepare_from_inputs(module, args, kwargs):
        input_ids = kwargs.get("input_ids", args[0] if args else None)
        grid = kwargs.get("video_grid_thw")
        if input_ids is None or grid is None or input_ids.shape[1] == 1:     # :118 (prefill only)
            return None
        assert input_ids.shape[0] == 1, "FrameFusion handles one sample per prefill (main.py:203)"
        cfg = module.config
        merge = cfg.vision_config.spatial_merge_size
        qwen2_vl_layout(input_ids, cfg.video_token_id, grid, merge).prepare(model.framefusion)
        return None

    inner._framefusion_prepare_handle = inner.register_forward_pre_hook(prepare_from_inputs, with_kwargs=True)


def register_hf_qwen2_vl() -> None:
    """Make apply_framefusion() accept transformers' Qwen2VLForConditionalGeneration (text decoder at
    ``model.language_model``) and bare wrappers exposing a Qwen2VLTextModel as ``.model``."""
    def is_full(model):
        try:
            from transformers import Qwen2VLForConditionalGeneration
        except Exception:
            return False
        return isinstance(model, Qwen2VLForConditionalGeneration)

    def is_text_wrapper(model):
        try:
            from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
        except Exception:
            return False
        return isinstance(getattr(model, "model", None), Qwen2VLTextModel)

    register_family(Family("hf_qwen2_vl", is_full, qwen2vl_text_model_forward, qwen2vl_decoder_forward,
                           qwen2vl_attention_forward, "model.language_model", "layers", "self_attn",
                           attach=attach_qwen2vl_prepare))
    register_family(Family("hf_qwen2_vl_text", is_text_wrapper, qwen2vl_text_model_forward, qwen2vl_decoder_forward,
                           qwen2vl_attention_forward, "model", "layers", "self_attn"))
