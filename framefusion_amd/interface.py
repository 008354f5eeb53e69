"""Public patcher with the reference's signatures (``framefusion/interface.py:47,140,169``).

``replace_framefusion_forward`` is the attach protocol itself and is kept as is: ONE shared
``FrameFusion`` instance is set as ``.framefusion`` on the wrapper module, the LLM, every decoder
layer and every attention module, and their ``forward`` is re-bound with ``types.MethodType``
(accelerate ``_hf_hook`` wrappers are preserved).  The model-family dispatch of the reference
(``apply_framefusion``, interface.py:58-124) depends on adapter code written against
transformers 4.45/4.51 internals plus the LLaVA-NeXT / VILA packages, which is outside this
build's scope; here families are *registered* (``register_family``) and
``apply_framefusion`` / ``get_token_type`` dispatch over the registry, raising
``NotImplementedError`` for unknown models exactly like the reference.
"""
from __future__ import annotations

from types import MethodType
from typing import Callable, Dict, List, NamedTuple, Optional

import torch.nn as nn

from .main import FrameFusion
from .utils import get_attr_by_name


class Family(NamedTuple):
    name: str
    matches: Callable[[object], bool]
    llm_forward: Callable
    decoder_forward: Callable
    attention_forward: Callable
    llm_key: str = "model"
    decoder_key: str = "layers"
    attention_key: str = "self_attn"
    # (attribute name, function) re-bound on the top-level model so that it builds patch_type and
    # calls framefusion.prepare(...) - e.g. ("prepare_inputs_labels_for_multimodal", fn) for
    # LLaVA-Video (interface.py:69-70)
    prepare_hook: Optional[tuple] = None
    # attach(model): alternative to prepare_hook for packers that cannot be re-bound as one method
    # (e.g. a forward pre-hook that derives the layout from input_ids and calls framefusion.prepare)
    attach: Optional[Callable] = None


_FAMILIES: List[Family] = []


def register_family(family: Family) -> None:
    _FAMILIES[:] = [f for f in _FAMILIES if f.name != family.name] + [family]


def registered_families() -> Dict[str, Family]:
    return {f.name: f for f in _FAMILIES}


def _lookup(model) -> Family:
    for fam in _FAMILIES:
        if fam.matches(model):
            return fam
    print("Model not supported")
    print(f"Model type: {type(model)}")
    raise NotImplementedError


def _bind_prepare(model, fam: Family) -> None:
    if fam.prepare_hook is not None:
        attr, fn = fam.prepare_hook
        setattr(model, attr, MethodType(fn, model))
    if fam.attach is not None:
        fam.attach(model)


def apply_framefusion(model, cost, similarity_lower_bound, ratio_lower_bound):
    """Apply FrameFusion to `model` (interface.py:47-137)."""
    fam = _lookup(model)
    _bind_prepare(model, fam)
    replace_framefusion_forward(
        model, cost=cost, similarity_lower_bound=similarity_lower_bound, ratio_lower_bound=ratio_lower_bound,
        llm_forward=fam.llm_forward, decoder_forward=fam.decoder_forward, attention_forward=fam.attention_forward,
        llm_key=fam.llm_key, decoder_key=fam.decoder_key, attention_key=fam.attention_key)


def get_token_type(model):
    """Only re-bind the family's patch_type builder (interface.py:140-166)."""
    _bind_prepare(model, _lookup(model))


def replace_framefusion_forward(module: nn.Module, cost: float, similarity_lower_bound: float,
                                ratio_lower_bound: float, llm_forward: Callable, decoder_forward: Callable,
                                attention_forward: Callable, llm_key: str = "model",
                                decoder_key: str = "layers", attention_key: str = "self_attn"):
    """interface.py:169-214.  Keys are dotted paths resolved hierarchically:
    module.<llm_key>.<decoder_key>[i].<attention_key>."""
    framefusion = FrameFusion(cost, similarity_lower_bound, ratio_lower_bound)
    module.framefusion = framefusion

    llm = get_attr_by_name(module, llm_key)
    assert isinstance(llm, nn.Module), f"{llm_key} is not a nn.Module"
    llm.framefusion = framefusion
    llm.forward = MethodType(llm_forward, llm)

    layers = get_attr_by_name(llm, decoder_key)
    for i, layer in enumerate(layers):
        assert isinstance(layer, nn.Module), f"{decoder_key}[{i}] is not a nn.Module"
        layer.framefusion = framefusion
        layer.forward = MethodType(decoder_forward, layer)
        if hasattr(layer, "_hf_hook"):       # keep accelerate's device-placement hook (interface.py:204-207)
            from accelerate.hooks import add_hook_to_module
            layer._old_forward = MethodType(decoder_forward, layer)
            add_hook_to_module(layer, layer._hf_hook)
        attention = get_attr_by_name(layer, attention_key)
        assert isinstance(attention, nn.Module), f"{decoder_key}[{i}].{attention_key} is not a nn.Module"
        attention.framefusion = framefusion
        attention.forward = MethodType(attention_forward, attention)
    return framefusion
