"""Qwen2 (the LLM of LLaVA-Video / LLaVA-NeXT-Video / MiniCPM-V / NVILA) for transformers 5.x.

Same protocol as the reference's ``framefusion/models/qwen2/modeling_qwen2.py``:
  * decoder layer: FrameFusion call A before attention at layer 0 (:45-46), call B after attention +
    residual at every layer (:67); the layer returns ``(hidden, position_embeddings, mask)`` (:85);
  * attention: importance of the last query captured only while
    ``finish_merging and not finish_pruning`` (:166-178) - here as a ``LastQuery`` handle (q_last + the
    un-repeated GQA keys, which stay in the KV cache): the prune call of the same layer computes the
    head-averaged importance itself, in the same host call as its select and gather;
  * model loop: position embeddings as a mutable ``[cos, sin]`` list threaded back from every layer
    (:263-266, :304-305); per-layer KV lengths differ after a reduction, which DynamicCache allows.

The multimodal packer that builds ``patch_type`` lives outside the LLM (llava_video.py:321-338);
callers invoke ``model.framefusion.prepare(...)`` themselves before the prefill.
"""
from __future__ import annotations

import torch

from ..interface import Family, register_family
from ..main import call_b_with_residual
from ..utils import last_query_importance


def qwen2_attention_forward(self, hidden_states, position_embeddings, attention_mask, past_key_values=None, **kwargs):
    from transformers.models.qwen2.modeling_qwen2 import (ALL_ATTENTION_FUNCTIONS, apply_rotary_pos_emb,
                                                           eager_attention_forward)
    input_shape = hidden_states.shape[:-1]
    hidden_shape = (*input_shape, -1, self.head_dim)
    query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    cos, sin = position_embeddings
    query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)
    if past_key_values is not None:
        key_states, value_states = past_key_values.update(key_states, value_states, self.layer_idx)

    importance = None
    q_len = query_states.shape[2]
    ff = self.framefusion
    if q_len > 1 and ff.finish_merging and not ff.finish_pruning:            # modeling_qwen2.py:168
        importance = last_query_importance(query_states, key_states, num=1, is_causal=attention_mask is None,
                                           scale=self.scaling, framefusion=ff, defer=True)

    attention_interface = ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, eager_attention_forward)
    attn_output, _ = attention_interface(self, query_states, key_states, value_states, attention_mask,
                                         dropout=0.0 if not self.training else self.attention_dropout,
                                         scaling=self.scaling, sliding_window=self.sliding_window, **kwargs)
    attn_output = self.o_proj(attn_output.reshape(*input_shape, -1).contiguous())
    return attn_output, importance


def qwen2_decoder_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                          use_cache=False, position_embeddings=None, **kwargs):
    if self.self_attn.layer_idx == 0:                                        # modeling_qwen2.py:45-46
        hidden_states, position_embeddings, attention_mask = self.framefusion(
            hidden_states, position_embeddings, attention_mask)
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)
    hidden_states, importance = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask,
                                               position_ids=position_ids, past_key_values=past_key_values,
                                               use_cache=use_cache, position_embeddings=position_embeddings, **kwargs)
    # modeling_qwen2.py:64-67: hidden = residual + attention output, then call B - here the add is formed
    # inside the reduction's two streaming passes (a plain add when no reduction is due)
    hidden_states, position_embeddings, attention_mask = call_b_with_residual(
        self.framefusion, residual, hidden_states, position_embeddings, attention_mask, importance)
    residual = hidden_states
    hidden_states = self.mlp(self.post_attention_layernorm(hidden_states))
    hidden_states = residual + hidden_states
    return hidden_states, position_embeddings, attention_mask                 # modeling_qwen2.py:85


def qwen2_model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
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
        position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + seen).unsqueeze(0)
    causal_mask = create_causal_mask(config=self.config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                     past_key_values=past_key_values, position_ids=position_ids)
    hidden_states = inputs_embeds
    position_embeddings = list(self.rotary_emb(hidden_states, position_ids))  # modeling_qwen2.py:263-266
    lengths = []
    for decoder_layer in self.layers[: self.config.num_hidden_layers]:
        hidden_states, position_embeddings, causal_mask = decoder_layer(      # modeling_qwen2.py:304-305
            hidden_states, attention_mask=causal_mask, position_embeddings=position_embeddings,
            position_ids=position_ids, past_key_values=past_key_values, use_cache=use_cache, **kwargs)
        lengths.append(hidden_states.shape[1])
    self.framefusion_lengths = lengths
    hidden_states = self.norm(hidden_states)
    return BaseModelOutputWithPast(last_hidden_state=hidden_states,
                                   past_key_values=past_key_values if use_cache else None)


def register_hf_qwen2() -> None:
    """Make apply_framefusion() accept transformers' Qwen2ForCausalLM / Qwen2Model wrappers."""
    def matches(model):
        try:
            from transformers import Qwen2ForCausalLM
        except Exception:
            return False
        return isinstance(model, Qwen2ForCausalLM)
    register_family(Family("hf_qwen2", matches, qwen2_model_forward, qwen2_decoder_forward, qwen2_attention_forward,
                           "model", "layers", "self_attn"))
