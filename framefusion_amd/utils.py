"""Boundary helpers with the reference's names (``framefusion/utils.py``).

``scaled_dot_product_attention`` is the importance input of the prune step: the attention
probabilities of the last ``num`` queries (utils.py:27-57).  Here it runs on the gfx950 kernel
behind ``ff_last_query_attention`` and reads the un-repeated GQA keys directly.
"""
from __future__ import annotations

import math
from typing import Any

import torch

from . import _lib
from ._lib import FrameFusionHipError

TEXT_TOKEN = -1
IGNORE_TOKEN = -2


def get_attr_by_name(obj: Any, name: str) -> Any:
    """Dotted attribute lookup with integer components indexing sequences, e.g.
    ``get_attr_by_name(model, "llm.model.layers.0")`` (reference utils.py:13-25)."""
    node = obj
    for part in name.split("."):
        node = node[int(part)] if part.isdigit() else getattr(node, part)
    return node


class LastQuery:
    """What the attention hook knows about the last `num` queries, prepared for the kernels: q_last [H, num, dh] contiguous,
    the keys read in place where their layout allows (`k`, element strides `sh` / `ss`), sizes, dtype code, scale.  Handed to
    ``FrameFusion.forward(..., self_attn_weights=LastQuery)`` by ``last_query_importance(..., defer=True)``: the prune call then
    computes the importance itself, in the same host call as its plan and gather (``ff_ctx_prune_from_qk``)."""
    __slots__ = ("q_last", "k", "sh", "ss", "H", "H_kv", "num", "S", "dh", "code", "factor", "is_causal", "bias", "dtype", "device")

    def __init__(self, query, key, num, is_causal, scale, bias=None):
        _lib.require_gpu(query, "scaled_dot_product_attention")
        if query.ndim != 4 or key.ndim != 4 or query.shape[0] != 1 or key.shape[0] != 1:
            raise FrameFusionHipError("expected query [1, H, L, dh] and key [1, H_kv, S, dh]")
        H, dh = query.shape[1], query.shape[3]
        H_kv, S = key.shape[1], key.shape[2]
        num = min(num, query.shape[2])
        if H % H_kv:
            raise FrameFusionHipError(f"{H} query heads are not a multiple of {H_kv} kv heads")
        code = _lib.DTYPE_CODE.get(query.dtype)
        if code is None or key.dtype != query.dtype:
            raise FrameFusionHipError(f"unsupported dtypes {query.dtype} / {key.dtype}")
        self.q_last = query[0, :, -num:, :].contiguous()
        # the keys as they are: [H_kv, S, dh] contiguous, or the transposed view of a k_proj output ([S, H_kv, dh] in
        # memory: what transformers' attention hands over before the cache concatenates anything) - any layout with
        # contiguous rows and 16-byte aligned strides is read in place, a copy of K is the exception
        k = key[0]
        esz = k.element_size()
        sh, ss, sd = k.stride()
        if (sd != 1 or (sh * esz) & 15 or (ss * esz) & 15 or ss < dh or k.data_ptr() & 15
                or ((S - 1) * ss + dh) * esz >= (1 << 32) or (H_kv > 1 and sh < dh)):
            k = k.contiguous()
            sh, ss = S * dh, dh
        self.k, self.sh, self.ss = k, sh, ss
        self.H, self.H_kv, self.num, self.S, self.dh, self.code = H, H_kv, num, S, dh, code
        self.factor = float(1 / math.sqrt(dh) if scale is None else scale)
        self.is_causal, self.bias = bool(is_causal), bias
        self.dtype, self.device = query.dtype, query.device

    # (what FrameFusion._prune asks of a weights tensor)
    @property
    def shape(self):
        return (1, self.H, self.num, self.S)

    def weights(self) -> torch.Tensor:
        """The [1, H, num, S] attention weights the handle stands for (utils.py:27-57), materialised - for code that wants to
        look at them (tests, the oracle): the prune itself never needs the tensor."""
        w, _ = _launch_last_query(self.q_last[None], self.k[None], self.num, self.is_causal, self.factor, True, False, bias=self.bias)
        return w


def _launch_last_query(query, key, num, is_causal, scale, want_weights, want_importance, plan=None, bias=None):
    lq = LastQuery(query, key, num, is_causal, scale, bias)
    lib = _lib.load()
    q_last, k, sh, ss = lq.q_last, lq.k, lq.sh, lq.ss
    H, H_kv, num, S, dh, code = lq.H, lq.H_kv, lq.num, lq.S, lq.dh, lq.code
    factor = 1 / math.sqrt(dh) if scale is None else scale
    dev = query.device
    weights = torch.empty(1, H, num, S, dtype=query.dtype, device=dev) if want_weights else None
    importance = torch.empty(S, dtype=query.dtype, device=dev) if want_importance else None
    ws_bytes = int(lib.ff_last_query_workspace_bytes(code, H, num, S, dh))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)          # (scores + statistics: dead when the call returns)
    if plan is not None:
        # the attention hook of an instance whose prune call comes next: importance + the select tables of that prune
        ctx_ptr, start, n_img, k_keep, stream = plan
        rc = lib.ff_ctx_last_query_importance(ctx_ptr, q_last.data_ptr(), k.data_ptr(), code, H, H_kv, num, S, dh, sh, ss,
                                              float(factor), 1 if is_causal else 0, bias.data_ptr() if bias is not None else None,
                                              importance.data_ptr(), start, n_img, k_keep, ws.data_ptr(), ws_bytes, stream or 0)
        _lib.check(rc, "ff_ctx_last_query_importance")
        return weights, importance
    rc = lib.ff_last_query_attention(q_last.data_ptr(), k.data_ptr(), code, H, H_kv, num, S, dh, sh, ss, float(factor),
                                     1 if is_causal else 0, bias.data_ptr() if bias is not None else None,
                                     weights.data_ptr() if want_weights else None,
                                     importance.data_ptr() if want_importance else None,
                                     0, 0, None, 0,
                                     ws.data_ptr(), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "ff_last_query_attention")
    return weights, importance


def _bias_from_mask(attn_mask, num, S, dtype, device):
    """utils.py:32,40-44: attn_bias = zeros(L, S, dtype=T); bool mask -> -inf where False, else += mask (T(mask))."""
    bias = torch.zeros(num, S, dtype=dtype, device=device)
    if attn_mask.dtype == torch.bool:
        bias.masked_fill_(attn_mask.to(device).logical_not(), float("-inf"))
    else:
        bias += attn_mask.to(device)
    return bias


def scaled_dot_product_attention(query, key, value, num=1, attn_mask=None, dropout_p=0.0,
                                 is_causal=False, scale=None, enable_gqa=False) -> torch.Tensor:
    """Reference utils.py:27-57: softmax attention weights [1, H, num, S] of the last ``num``
    queries.  ``key`` may carry H_kv < H heads whether or not ``enable_gqa`` is set (head h reads
    kv head h // (H // H_kv), exactly what repeat_kv / repeat_interleave produce).  ``attn_mask``
    (boolean or additive, broadcastable to [num, S], utils.py:40-44) becomes the T-valued bias the
    score kernel adds; like the reference it cannot be combined with ``is_causal`` (utils.py:35)."""
    if dropout_p != 0.0:
        raise FrameFusionHipError("dropout_p must be 0 (prefill-time importance scoring)")
    bias = None
    if attn_mask is not None:
        assert not is_causal                                                        # utils.py:35
        bias = _bias_from_mask(attn_mask, min(num, query.shape[2]), key.shape[2], query.dtype, query.device)
    weights, _ = _launch_last_query(query, key, num, is_causal, scale, True, False, bias=bias)
    return weights


def last_query_importance(query, key, num=1, is_causal=True, scale=None, framefusion=None, defer=False):
    """Fused form for the attention hook: the head/query mean of the weights
    above, shaped [1, 1, 1, S] so that FrameFusion.forward's own mean (main.py:70) is the identity.
    With `framefusion` (the instance whose prune call will consume the result) the importance kernel
    also accumulates the select tables of that call in the instance's workspace (ff_ctx_last_query_importance):
    the prune call then goes straight to its plan.  `defer=True` returns a `LastQuery` handle instead of a tensor and launches
    nothing: passed to ``FrameFusion.forward`` as ``self_attn_weights`` (what the adapters in models/ do) it makes hook +
    prune ONE host call - `FrameFusion.prune_from_qk` is the same thing by name."""
    if defer and getattr(framefusion, "accepts_last_query", framefusion is None):
        # (an object on `.framefusion` that is not this build's FrameFusion - the reference's module, an eager stand-in - gets
        # the tensor below instead)
        # nothing is launched here: the prune call that receives this handle enqueues importance + plan + gather in ONE
        # crossing of the C ABI (K stays where it is - the layer's KV cache - until then)
        return LastQuery(query, key, num, is_causal, scale)
    plan, token = None, None
    if framefusion is not None and query.is_cuda:
        plan, token = framefusion._expect_importance(key.shape[2], query.dtype, query.device)
    _, imp = _launch_last_query(query, key, num, is_causal, scale, False, True, plan)
    out = imp[None, None, None, :]
    if token is not None:
        out._ff_tables = token
    return out
