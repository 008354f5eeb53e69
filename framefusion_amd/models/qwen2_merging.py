"""Fixed-sparsity merging baseline wired into transformers 5.x Qwen2
(reference: ``replace_Qwen2_merging``, framefusion/models/qwen2/modeling_qwen2_baseline.py:860-874,
attention :876-1126, decoder layer :1128-1203, model loop :1205-1336).

In the reference the merging block sits at the top of the attention forward and hands its mask back
so that the decoder layer can compact the residual (:1180-1185).  Here the decoder layer calls the
merger on the normed activations itself - the same tensors, one call earlier in the same layer -
and the stock attention runs on the shortened sequence.  As in the reference (:1098-1100 slices
only the key axis of a 4-D mask) the prefill must run with ``attention_mask=None`` (causal SDPA).
"""
from __future__ import annotations

from types import MethodType
from typing import Sequence

import torch

from ..baseline import FixedSparsityMerging


def qwen2_merging_decoder_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                                  use_cache=False, position_embeddings=None, **kwargs):
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)                       # :1163-1165
    hidden_states, token_mask, residual = self.merging.merge(                 # :905-1053, :1180-1185
        self.self_attn.layer_idx, hidden_states, position_embeddings, residual)
    if token_mask is not None and attention_mask is not None:
        raise NotImplementedError("the merging baseline runs its prefill with attention_mask=None (is_causal SDPA)")
    hidden_states, _ = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask,
                                      position_ids=position_ids, past_key_values=past_key_values,
                                      use_cache=use_cache, position_embeddings=tuple(position_embeddings), **kwargs)
    hidden_states = residual + hidden_states                                  # :1187
    residual = hidden_states
    hidden_states = self.mlp(self.post_attention_layernorm(hidden_states))    # :1190-1194
    return residual + hidden_states


def qwen2_merging_model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                                inputs_embeds=None, use_cache=None, **kwargs):
    from transformers.cache_utils import DynamicCache
    from transformers.modeling_outputs import BaseModelOutputWithPast
    if (input_ids is None) ^ (inputs_embeds is not None):
        raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    if attention_mask is not None and inputs_embeds.shape[1] > 1:
        raise NotImplementedError("the merging baseline runs its prefill with attention_mask=None")
    if use_cache and past_key_values is None:
        past_key_values = DynamicCache(config=self.config)
    if position_ids is None:
        seen = past_key_values.get_seq_length() if past_key_values is not None else 0
        position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + seen).unsqueeze(0)
    hidden_states = inputs_embeds
    position_embeddings = list(self.rotary_emb(hidden_states, position_ids))  # :1270-1275 (a list, compacted in place)
    lengths = []
    for decoder_layer in self.layers[: self.config.num_hidden_layers]:
        hidden_states = decoder_layer(hidden_states, attention_mask=None, position_embeddings=position_embeddings,
                                      position_ids=position_ids, past_key_values=past_key_values, use_cache=use_cache,
                                      **kwargs)
        lengths.append(hidden_states.shape[1])
    self.merging_lengths = lengths
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states,
                                   past_key_values=past_key_values if use_cache else None)


def replace_qwen2_merging(model, sparsity: Sequence[float] = (0.1,) * 28) -> FixedSparsityMerging:
    """``replace_Qwen2_merging`` (:860-874) for transformers' Qwen2ForCausalLM: one shared merger on
    the wrapper, the model and every decoder layer.  Callers invoke
    ``model.merging.prepare(patch_type, patch_num)`` before each prefill (the reference reads
    ``model.patch_type`` / ``model.patch_num`` set by the packer)."""
    merger = FixedSparsityMerging(sparsity)
    if len(merger.sparsity) < len(model.model.layers):
        raise ValueError(f"{len(model.model.layers)} layers need as many sparsity entries")
    model.merging = merger
    model.sparsity = merger.sparsity
    model.model.merging = merger
    model.model.forward = MethodType(qwen2_merging_model_forward, model.model)
    for layer in model.model.layers:
        layer.merging = merger
        layer.forward = MethodType(qwen2_merging_decoder_forward, layer)
    return merger
