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
(``ff_merge_begin`` + ``ff_merge_finish_topk``).  top-k ties go to the lowest by-patch index, as in
the main path.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import FFAux, FrameFusionHipError
from .main import FrameFusion, TEXT_TOKEN, _Scratch, _dtype_code


def compute_density_overhead(sparsity_list) -> tuple:
    """(mean cumulative density, final density) of a per-layer sparsity schedule (:26-43)."""
    density_list = [1 - s for s in sparsity_list]
    cost = 0.0
    remaining_density = 1.0
    for density in density_list:
        remaining_density *= density
        cost += remaining_density
    norm_cost = cost / len(density_list)
    return norm_cost, remaining_density


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
        for s in self._scratch.values():
            s.order_valid_for = None

    def _scratch_for(self, device, L, dtype) -> _Scratch:
        key = (device.type, device.index)
        s = self._scratch.get(key)
        if s is None:
            s = self._scratch[key] = _Scratch(device)
        cur = torch.cuda.current_stream(device)
        last = getattr(s, "last_stream", None)
        if last is not None and last != cur:
            cur.wait_stream(last)
        s.last_stream = cur
        return s.ensure(L, dtype)

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
        sc = self._scratch_for(device, L, dtype)
        stream = _lib.stream_ptr()
        order_valid = 1 if sc.order_valid_for == (self._ptype_gen, L) else 0
        if sc.dirty:
            sc.ws.zero_()
            sc.stats.zero_()
            sc.dirty = False
        sc.seq += 1
        seq = sc.seq
        sc.dirty = True
        _lib.check(lib.ff_merge_begin(hidden.data_ptr(), None, code, L, d, ptype.data_ptr(), int(self.patch_num), order_valid,
                                      0.0, sc.order.data_ptr(), sc.inv.data_ptr(), sc.sim32.data_ptr(), sc.stats.data_ptr(),
                                      seq, 0, 0,
                                      sc.ws.data_ptr(), sc.ws_bytes, stream), "ff_merge_begin")

        out = torch.empty(1, L, d, dtype=dtype, device=device)
        ptype_out = torch.empty(1, L, dtype=torch.int64, device=device)
        aux = (FFAux * _lib.MAX_AUX)()
        n_aux = FrameFusion._fill_aux(aux, 0, [ptype.view(1, L)], [ptype_out], L)
        rebuild = None
        if position_embeddings is not None:
            if type(position_embeddings) != list:
                raise NotImplementedError("position_embeddings must be the mutable [cos, sin] list (:1273)")
            srcs, outs, rebuild = FrameFusion._aux_for_positions(position_embeddings, L, L)
            n_aux = FrameFusion._fill_aux(aux, n_aux, srcs, outs, L)
        res_out = None
        if residual is not None:
            if residual.shape != hidden_states.shape or residual.device != device:
                raise FrameFusionHipError("residual must have the shape and device of hidden_states")
            res = residual.contiguous()
            res_out = torch.empty_like(res)
            n_aux = FrameFusion._fill_aux(aux, n_aux, [res], [res_out], L)
        _lib.check(lib.ff_merge_finish_topk(hidden.data_ptr(), out.data_ptr(), code, L, d, L, prune_num, _lib.FOLD_MEAN,
                                            sc.order.data_ptr(), sc.inv.data_ptr(), sc.sim32.data_ptr(),
                                            sc.member.data_ptr(), sc.dst.data_ptr(), sc.keep.data_ptr(),
                                            sc.stats.data_ptr(), sc.stats_host_ptr, seq, aux, n_aux,
                                            sc.order_next.data_ptr(), sc.inv_next.data_ptr(),
                                            sc.ws.data_ptr(), sc.ws_bytes, stream), "ff_merge_finish_topk")
        sc.dirty = False
        token_mask = sc.keep[:L].bool().view(1, L)          # a copy: the scratch is reused by the next layer

        st = sc.wait_stats(seq)
        if int(st[_lib.STAT_ERROR]):
            sc.dirty = True
            sc.order_valid_for = None
            raise FrameFusionHipError(f"device-side check failed in the merge call (error bits {int(st[_lib.STAT_ERROR]):#x})")
        nv, L_out = int(st[_lib.STAT_NV]), int(st[_lib.STAT_LOUT])
        if nv <= 0:
            raise ValueError("No token in this patch")                               # :982-983
        if L_out == L:
            # the forced top-k folded nothing (its only candidate was by-patch slot 0, which has no
            # predecessor): the merge kernel wrote nothing - the caller's tensors ARE the result and the
            # by-patch order in the scratch still describes them
            self.last_call = dict(kind="merge", L_in=L, L_out=L, nv=nv, ftn=ftn, k=int(st[_lib.STAT_K]),
                                  scratch=sc, dtype=dtype, order=sc.order)
            sc.order_valid_for = (self._ptype_gen, L)
            return hidden_states, token_mask, residual
        self._ftn = ftn - (L - L_out)                       # every dropped token was a visual one
        self.patch_type = ptype_out[:, :L_out]                                       # :1051
        self.last_call = dict(kind="merge", L_in=L, L_out=L_out, nv=nv, ftn=ftn, k=int(st[_lib.STAT_K]),
                              scratch=sc, dtype=dtype, order=sc.order)
        sc.order, sc.order_next = sc.order_next, sc.order
        sc.inv, sc.inv_next = sc.inv_next, sc.inv
        sc.order_valid_for = (self._ptype_gen, L_out)
        if rebuild is not None:
            rebuild(L_out)
        return out[:, :L_out], token_mask, (res_out[:, :L_out] if res_out is not None else None)

    def last_plan(self):
        c = self.last_call
        sc, L, nv = c["scratch"], c["L_in"], c["nv"]
        return dict(keep=sc.keep[:L], member=sc.member[:L], sim=sc.sim(c["dtype"], nv), order=c["order"][:nv])
