"""Fixed-sparsity token merging - the reference's ablation baseline - on the same HIP kernels.

Reference: ``framefusion/models/qwen2/modeling_qwen2_baseline.py``
  * ``compute_density_overhead`` (:26-43);
  * ``replace_Qwen2_merging(model, sparsity=[0.1] * 28)`` (:860-874): every layer ℓ merges
    ``floor(sparsity[ℓ] * ftn)`` visual tokens during prefill;
  * the merging block at the top of the attention forward (:905-1053): by-patch cosine similarity
    of the (normed) activations, ``torch.topk`` with that k, run detection, every run replaced by
    the ``.mean()`` of its rows at the run's anchor, compaction of activations / ``patch_type`` and
    (:1081-1085) cos/sin; the decoder layer compacts the residual with the same mask (:1180-1185).

Differences from FrameFusion.forward (main.py): k is given instead of derived from a threshold and
a budget, and a run is averaged in fp32 with one rounding (``FF_FOLD_MEAN``) instead of the
per-add rounding of ``index_add_``.  Everything else - K0 order (maintained across layers), K1
similarities, the radix select, the scan, K4 with its aux gathers - is shared
(``ff_ctx_merge_begin`` / ``ff_ctx_merge_finish`` with ``force_k`` and ``FF_FOLD_MEAN``).  top-k ties go to the lowest by-patch index, as in
the main path.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import FrameFusionHipError
from .main import FrameFusion, TEXT_TOKEN, _PACK_I64, _PACK_PTR, _Scratch, _dtype_code, _fail


def compute_density_overhead(sparsity_list) -> tuple:
    """What a per-layer sparsity schedule costs (modeling_qwen2_baseline.py:26-43): the fraction of tokens alive after
    layer i is the running product of (1 - sparsity) - returns (its mean over the layers, its last value).  Python floats,
    folded left to right, so the pair is bit-identical to the reference's."""
    alive = 1.0
    alive_per_layer = []
    for sparsity in sparsity_list:
        alive *= 1 - sparsity
        alive_per_layer.append(alive)
    total = 0.0
    for a in alive_per_layer:
        total += a
    return total / len(alive_per_layer), alive


class FixedSparsityMerging:
    """The state the reference keeps on ``model`` (``model.sparsity``, ``model.patch_type``,
    ``model.patch_num``, :861, :913, :939) plus the per-device scratch of the kernels."""

    def __init__(self, sparsity: Sequence[float] = (0.1,) * 28):
        self.sparsity = list(sparsity)
        self._ptype_gen = 0
        self.patch_type = None
        self.patch_num: Optional[int] = None
        self._ftn: Optional[int] = None
        self._scratch = {}
        self.last_call = None

    @property
    def patch_type(self) -> Optional[torch.Tensor]:
        return self._patch_type

    @patch_type.setter
    def patch_type(self, value):
        # any assignment invalidates the by-patch order cached in the scratch (keyed on the generation)
        self._patch_type = value
        self._ptype_gen += 1

    def prepare(self, patch_type: torch.Tensor, patch_num: int):
        """New prefill: the full-length patch_type row of the packer."""
        self.patch_type = patch_type
        self.patch_num = patch_num
        self._ftn = None

    def _scratch_for(self, device, L):
        key = (device.type, device.index)
        s = self._scratch.get(key)
        if s is None:
            s = self._scratch[key] = _Scratch(device)
        ptr = _lib.stream_ptr()
        if s.last_stream_ptr != ptr:
            cur = torch.cuda.current_stream(device)
            last = s.last_stream
            if last is not None and last != cur:
                cur.wait_stream(last)
            s.last_stream, s.last_stream_ptr = cur, ptr
        return s.ensure(L), ptr

    def merge(self, layer_idx: int, hidden_states: torch.Tensor, position_embeddings: Optional[List[torch.Tensor]] = None,
              residual: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """One layer's merging block (:916-1052).  hidden_states [1, L, d] are the activations the
        attention projects (after input_layernorm); `position_embeddings` (a [cos, sin] list of
        [1, L, dh]) is compacted in place (:1081-1085); `residual` [1, L, d] with the same mask
        (:1180-1185).  Returns (hidden_states, token_mask [1, L] bool or None, residual)."""
        _lib.require_gpu(hidden_states, "FixedSparsityMerging.merge")
        bsz, L, d = hidden_states.size()
        if L <= 1:                                                                   # :916 decode
            return hidden_states, None, residual
        assert bsz == 1, "Only support batch size 1"                                # :936
        device, dtype = hidden_states.device, hidden_states.dtype
        ptype = self.patch_type
        if ptype is None:
            raise AttributeError("FixedSparsityMerging.prepare() has not been called")
        if ptype.device != device or ptype.dtype != torch.int64 or not ptype.is_contiguous():
            ptype = ptype.to(device=device, dtype=torch.int64).contiguous()
            self.patch_type = ptype
        if ptype.numel() != L:
            raise FrameFusionHipError(f"patch_type has {ptype.numel()} entries for a sequence of {L}")
        if self._ftn is None:                                                        # :919, one readback per prefill:
            self._ftn = int((ptype != TEXT_TOKEN).sum().item())                      # later layers know it from k
        ftn = self._ftn
        prune_num = math.floor(self.sparsity[layer_idx] * ftn)                       # :918-920
        if prune_num <= 0:                                                           # :922
            self.last_call = dict(kind="skip", L_in=L, L_out=L, k=0)
            return hidden_states, None, residual

        lib = _lib.load()
        code = _dtype_code(hidden_states)
        hidden = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        sc, stream = self._scratch_for(device, L)
        order_valid = 1 if sc.order_gen == self._ptype_gen else 0
        call = sc.call
        _lib.MERGE_CALL_HEAD.pack_into(call, 0, hidden.data_ptr(), 0, 0, ptype.data_ptr(), code, L, d, L, int(self.patch_num),
                                       order_valid, 0.0, 0.0, 0.0, prune_num, _lib.FOLD_MEAN, 0, 0, stream or 0, 0)
        sc.order_gen = None
        rc = lib.ff_ctx_merge_begin(sc.ctx_ptr, sc.call_ptr)
        if rc:
            _fail(rc, "merge")

        out = torch.empty(1, L, d, dtype=dtype, device=device)
        ptype_out = torch.empty(1, L, dtype=torch.int64, device=device)
        srcs, outs = [ptype.view(1, L)], [ptype_out]
        rebuild = None
        if position_embeddings is not None:
            if type(position_embeddings) != list:
                raise NotImplementedError("position_embeddings must be the mutable [cos, sin] list (:1273)")
            s2, o2, rebuild = FrameFusion._aux_for_positions(position_embeddings, L, L)
            srcs += s2
            outs += o2
        res_out = None
        if residual is not None:
            if residual.shape != hidden_states.shape or residual.device != device:
                raise FrameFusionHipError("residual must have the shape and device of hidden_states")
            res = residual.contiguous()
            res_out = torch.empty_like(res)
            srcs.append(res)
            outs.append(res_out)
        n_aux = sc.put_aux(call, _lib.MERGE_CALL_AUX_OFFSET, zip(srcs, outs), L)
        _PACK_PTR.pack_into(call, 16, out.data_ptr())
        _PACK_I64.pack_into(call, _lib.MERGE_CALL_AUX_OFFSET - 8, n_aux)
        _lib.MASK_TRIPLE.pack_into(call, _lib.MERGE_CALL_MASK_OFFSET, 0, 0, 0)
        rc = lib.ff_ctx_merge_finish(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
        nv, _ftn, _count, _branch, k_used, L_out, err, _unhinted, _wait, _applied = _lib.MERGE_RESULT.unpack_from(sc.res)
        if rc:
            _fail(rc, "merge", err)
        sc.sync_views()
        token_mask = sc.keep[:L].bool().view(1, L)          # a copy: the scratch is reused by the next layer
        if nv <= 0:
            raise ValueError("No token in this patch")                               # :982-983
        if L_out == L:
            # the forced top-k folded nothing (its only candidate was by-patch slot 0, which has no
            # predecessor): the merge kernel wrote nothing - the caller's tensors ARE the result and the
            # by-patch order in the scratch still describes them
            self.last_call = dict(kind="merge", L_in=L, L_out=L, nv=nv, ftn=ftn, k=k_used,
                                  scratch=sc, dtype=dtype, order=sc.order)
            sc.order_gen = self._ptype_gen
            return hidden_states, token_mask, residual
        self._ftn = ftn - (L - L_out)                       # every dropped token was a visual one
        self.last_call = dict(kind="merge", L_in=L, L_out=L_out, nv=nv, ftn=ftn, k=k_used,
                              scratch=sc, dtype=dtype, order=sc.order_next)
        self.patch_type = ptype_out[:, :L_out]                                       # :1051
        sc.order_gen = self._ptype_gen
        if rebuild is not None:
            rebuild(L_out)
        return out[:, :L_out], token_mask, (res_out[:, :L_out] if res_out is not None else None)

    def last_plan(self):
        c = self.last_call
        sc, L, nv = c["scratch"], c["L_in"], c["nv"]
        return dict(keep=sc.keep[:L], member=sc.member[:L], sim=sc.sim(c["dtype"], nv), order=c["order"][:nv])
