"""FrameFusion token reduction for MI355X - host side.

Same class, method and attribute surface as the reference's ``framefusion/main.py`` (``FrameFusion``
:8-343, ``cosine_similarity`` :345, ``find_contigious_latter_index`` :351, ``TEXT_TOKEN`` /
``IGNORE_TOKEN`` :5-6) so the reference's patched decoder/attention forwards can call it
unchanged; every tensor-sized operation behind it is a hand-written gfx950 kernel reached through
the C ABI of ``include/framefusion_hip.h``.  The host keeps only the state machine and the python
float budget arithmetic, and reads ONE 256-byte result block back per merge call (the reference
performs 17 host syncs per call).  The prune call needs no readback at all.

There is no eager/CPU fallback: CPU tensors raise ``FrameFusionHipError``.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import List, Optional

import torch
from torch import nn

from . import _lib
from ._lib import FFAux, FrameFusionHipError

TEXT_TOKEN = -1
IGNORE_TOKEN = -2


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _lib.DTYPE_CODE[t.dtype]
    except KeyError:
        raise FrameFusionHipError(f"unsupported activation dtype {t.dtype} (fp32 / bf16 / fp16 only)")


def _round_to(value: float, dtype: torch.dtype) -> float:
    """T(value): a python comparison scalar is cast to the tensor dtype by torch
    (``sim >= 0.6`` compares against bf16(0.6) = 0.6015625)."""
    return float(torch.tensor(value, dtype=dtype))


def _to_int(x) -> int:
    return int(x.item()) if isinstance(x, torch.Tensor) else int(x)


class _Scratch:
    """Per-device integer scratch reused across calls (order, member, dst, keep, stats)."""

    def __init__(self, device):
        self.device = device
        self.cap = 0
        self.order_valid_for = None     # (patch_type generation, L) the cached by-patch order belongs to

    def ensure(self, L: int, sim_dtype):
        if L > self.cap:
            cap = max(L, 1024)
            dev = self.device
            self.order = torch.empty(cap, dtype=torch.int32, device=dev)
            self.order_next = torch.empty(cap, dtype=torch.int32, device=dev)
            self.inv = torch.empty(cap, dtype=torch.int32, device=dev)          # inverse of order (slot of a position)
            self.inv_next = torch.empty(cap, dtype=torch.int32, device=dev)
            self.member = torch.empty(cap, dtype=torch.uint8, device=dev)
            self.dst = torch.empty(cap, dtype=torch.int32, device=dev)
            self.keep = torch.empty(cap, dtype=torch.uint8, device=dev)
            self.sim32 = torch.empty(cap, dtype=torch.float32, device=dev)   # viewed as T
            self.ws_bytes = int(_lib.load().ff_workspace_bytes(cap, 1))
            self.ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=dev)   # select tables start clean
            self.stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=dev)
            self.stats_host = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64).pin_memory()
            self.stats_host_ptr = self.stats_host.data_ptr()     # device-visible (pinned, UVA)
            self.stats_np = self.stats_host.numpy()              # shares the pinned memory
            self.seq = 0
            self.dirty = False          # a call died between begin and finish: the select tables need a reset
            self.cap = cap
            self.order_valid_for = None
        return self

    def sim(self, dtype, n):
        return self.sim32.view(dtype)[:n]

    def wait_stats(self, seq, spins=4_000_000):
        """Poll the pinned result block until the device has published call `seq`."""
        view = self.stats_np
        word = _lib.STAT_SEQ
        n = 0
        while view[word] != seq:
            n += 1
            if (n & 1023) == 0:
                time.sleep(0)                    # let other Python threads (other replicas) run
            if n > spins:
                torch.cuda.current_stream().synchronize()
                if view[word] != seq:
                    raise FrameFusionHipError("the device never published the result block of this call")
                break
        return view


class FrameFusion(nn.Module):
    def __init__(self, cost=0.3, similarity_lower_bound=0.6, ratio_lower_bound=0.1):
        super().__init__()
        self.cost = cost
        self.similarity_lower_bound = similarity_lower_bound
        self.ratio_lower_bound = ratio_lower_bound
        self._scratch = {}
        self._ptype_gen = 0       # bumped whenever patch_type is (re)assigned: keys the cached by-patch order
        self.last_call = None     # diagnostics of the most recent reduction (tests / bench)

    _PLAIN = (bool, int, float, str, list, tuple, dict, type(None), torch.Tensor)

    def __setattr__(self, name, value):
        # The module has no parameters, buffers or submodules; its attributes are per-prefill state
        # (flags, scalars, the patch_type tensor) rewritten on every call.  nn.Module.__setattr__
        # spends ~2 us per assignment on parameter/buffer/submodule bookkeeping before it ends up in
        # __dict__ as well, so plain values go there directly.
        if name == "patch_type":
            # ANY assignment (prepare(), the compaction of a merge call, a caller writing the attribute
            # as the reference allows) invalidates the by-patch order kept in the scratch
            self.__dict__["_ptype_gen"] = self.__dict__.get("_ptype_gen", 0) + 1
        if type(value) in FrameFusion._PLAIN:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def __getstate__(self):
        # The scratch holds device buffers and a raw pointer into pinned host memory that the device
        # publishes results into: a copy.deepcopy / pickle of a used instance must not share it.
        state = self.__dict__.copy()
        state["_scratch"] = {}
        state["last_call"] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # ---- reference main.py:15-38 -----------------------------------------------------------
    def prepare(self, patch_type: torch.Tensor, patch_num: int, image_token_start_index,
                image_token_end_index, image_token_length, original_length: int,
                finish_merging: bool = False, finish_pruning: bool = False,
                sparsity_list: Optional[List[float]] = None):
        self.patch_type = patch_type
        self.patch_num = patch_num
        self.image_token_start_index = image_token_start_index
        self.image_token_end_index = image_token_end_index
        self.image_token_length = image_token_length
        self.original_length = original_length
        self.finish_merging = finish_merging
        self.finish_pruning = finish_pruning
        self.sparsity_list = [] if sparsity_list is None else sparsity_list
        for s in self._scratch.values():
            s.order_valid_for = None
        self._layout_hint = self._frame_major_hint(patch_num, image_token_start_index, image_token_length)

    @staticmethod
    def _frame_major_hint(patch_num, start, length):
        """(pre, frames) if the scalars describe whole frames of `patch_num` tokens starting at `start`
        (the layout every packer of the reference builds), else None.  Only host values are looked at
        (no device read-back); the similarity kernel verifies the hint against patch_type."""
        def host_int(x):
            if isinstance(x, torch.Tensor):
                if x.is_cuda or x.numel() != 1:
                    return None
                x = x.reshape(-1)[0].item()
            try:
                return int(x) if float(x) == int(x) else None
            except (TypeError, ValueError):
                return None
        P, pre, n = host_int(patch_num), host_int(start), host_int(length)
        if P is None or pre is None or n is None or P < 1 or pre < 0 or n < P or n % P:
            return None
        return pre, n // P

    # ---- reference main.py:40-140 ------------------------------------------------------------
    def forward(self, hidden_states, position_embeddings, attention_mask, self_attn_weights=None, residual=None):
        """Reference signature (main.py:40-42) plus `residual`: when given, the sequence that is reduced
        is ``residual + hidden_states`` - the add the decoder performs right before call B
        (models/qwen2/modeling_qwen2.py:64-67) - formed inside the two streaming passes instead of by an
        eager add whose result would be written once and read twice."""
        dev = hidden_states.device
        if dev.type == "cuda" and dev.index != torch.cuda.current_device():
            # the kernels are launched through ctypes on the CURRENT device's stream: follow the
            # tensors (several replicas on several GPUs in one process, as in the reference's demo)
            with torch.cuda.device(dev):
                return self.forward(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)
        bsz, q_len, hidden_size = hidden_states.size()
        prune_now = q_len > 1 and self.finish_merging == True and self.finish_pruning == False
        merge_now = q_len > 1 and (not self.finish_merging)
        if residual is not None:
            if residual.shape != hidden_states.shape or residual.dtype != hidden_states.dtype or residual.device != dev:
                raise FrameFusionHipError("residual must have the shape, dtype and device of hidden_states")
            if not (prune_now or merge_now):
                return residual + hidden_states, position_embeddings, attention_mask
        if prune_now:
            hidden_states, position_embeddings, attention_mask = self._prune(
                hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)
            residual = None                      # (folded into the pruned rows)
        if merge_now:
            hidden_states, position_embeddings, attention_mask = self._merge(
                hidden_states, position_embeddings, attention_mask, residual)
        return hidden_states, position_embeddings, attention_mask

    def forward_residual(self, residual, hidden_states, position_embeddings, attention_mask, self_attn_weights=None):
        """Call B of the decoder layer with its residual add fused in:
        ``forward(residual + hidden_states, ...)`` without materialising the sum."""
        return self.forward(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual=residual)

    # ---- helpers -------------------------------------------------------------------------------
    def _threshold_for(self, dtype) -> float:
        key = (self.similarity_lower_bound, dtype)
        if getattr(self, "_thr_cache_key", None) != key:
            self._thr_cache_key, self._thr_cache = key, _round_to(self.similarity_lower_bound, dtype)
        return self._thr_cache

    def _scratch_for(self, device, L, dtype) -> _Scratch:
        key = (device.type, device.index)
        s = self._scratch.get(key)
        if s is None:
            s = self._scratch[key] = _Scratch(device)
        # the scratch is reused call after call: if the caller switched streams, order the new
        # stream behind the one that last touched it
        ptr = _lib.stream_ptr()
        if getattr(s, "last_stream_ptr", None) != ptr:
            cur = torch.cuda.current_stream(device)
            last = getattr(s, "last_stream", None)
            if last is not None and last != cur:
                cur.wait_stream(last)
            s.last_stream, s.last_stream_ptr = cur, ptr
        return s.ensure(L, dtype)

    @staticmethod
    def _aux_for_positions(position_embeddings, L: int, L_cap: int):
        """Describe the position container for K4 (main.py:142-178): returns
        (aux descriptors, outputs, rebuild(outs, L_out) -> new container)."""
        if type(position_embeddings) == list:
            assert len(position_embeddings) == 2
            srcs, outs = [], []
            for t in position_embeddings:
                if t.ndim not in (3, 4) or t.shape[-2] != L:
                    raise FrameFusionHipError(f"position embedding of shape {tuple(t.shape)} does not have "
                                              f"{L} tokens on its second-to-last axis")
                t = t.contiguous()
                srcs.append(t)
                outs.append(torch.empty(t.shape[:-2] + (L_cap, t.shape[-1]), dtype=t.dtype, device=t.device))

            def rebuild(L_out):
                for i in range(2):
                    position_embeddings[i] = outs[i].narrow(outs[i].ndim - 2, 0, L_out)
                return position_embeddings
            return srcs, outs, rebuild
        if type(position_embeddings) == torch.Tensor:
            if position_embeddings.ndim != 2:
                raise NotImplementedError("Only support 2D position embeddings")
            t = position_embeddings.contiguous()
            out = torch.empty(t.shape[0], L_cap, dtype=t.dtype, device=t.device)
            return [t], [out], (lambda L_out: out.narrow(1, 0, L_out))
        raise NotImplementedError("Only support list or tensor for position embeddings")

    @staticmethod
    def _fill_aux(arr, n0, srcs, outs, L):
        n = n0
        for s, o in zip(srcs, outs):
            if n >= _lib.MAX_AUX:
                raise FrameFusionHipError("too many auxiliary tensors")
            if s.ndim == 2:         # [B, L] ids
                row, outer = s.element_size(), s.shape[0]
            else:                   # [..., L, dh]
                row = s.shape[-1] * s.element_size()
                outer = s.numel() // (L * s.shape[-1])
            arr[n] = FFAux(s.data_ptr(), o.data_ptr(), row, outer)
            n += 1
        return n

    def _gather_mask(self, attention_mask, L, L_cap, dst, stream):
        m = attention_mask
        if m.ndim != 4 or m.shape[0] != 1 or m.shape[1] != 1 or m.shape[2] != L or m.shape[3] != L:
            raise FrameFusionHipError(f"attention mask of shape {tuple(m.shape)} is not [1, 1, {L}, {L}]")
        m = m.contiguous()
        out = torch.empty(1, 1, L_cap, L_cap, dtype=m.dtype, device=m.device)
        _lib.check(_lib.load().ff_gather_mask(m.data_ptr(), out.data_ptr(), m.element_size(), L, L_cap,
                                              dst.data_ptr(), stream), "ff_gather_mask")
        return out

    # ---- merge call: main.py:104-138 -------------------------------------------------------------
    def _merge(self, hidden_states, position_embeddings, attention_mask, residual=None):
        return self._merge_complete(self._merge_launch(hidden_states, position_embeddings, attention_mask, residual=residual))

    def _merge_launch(self, hidden_states, position_embeddings, attention_mask, use_hint=True, residual=None):
        """Enqueue the whole merge call on the current stream and return without waiting: the
        state machine is advanced by _merge_complete."""
        _lib.require_gpu(hidden_states, "FrameFusion.forward")
        lib = _lib.load()
        bsz, L, d = hidden_states.size()
        assert bsz == 1, "Only support batch size 1"                                # main.py:203
        device = hidden_states.device
        dtype = hidden_states.dtype
        code = _dtype_code(hidden_states)
        hidden = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        addend = None
        if residual is not None:
            addend = residual if residual.is_contiguous() else residual.contiguous()
        addend_ptr = addend.data_ptr() if addend is not None else None

        ptype = self.patch_type
        if ptype.device != device or ptype.dtype != torch.int64 or not ptype.is_contiguous():
            ptype = ptype.to(device=device, dtype=torch.int64).contiguous()
            self.patch_type = ptype
        if ptype.numel() != L:
            raise FrameFusionHipError(f"patch_type has {ptype.numel()} entries for a sequence of {L}")

        sub = self._compute_pruning_ratio(self.sparsity_list, self.cost)           # main.py:109
        sc = self._scratch_for(device, L, dtype)
        stream = _lib.stream_ptr()
        sim_ptr = sc.sim32.data_ptr()
        order_valid = 1 if sc.order_valid_for == (self._ptype_gen, L) else 0

        # first half (K0 + K1) goes out before any output tensor exists: the allocations below
        # overlap the similarity pass
        thr = self._threshold_for(dtype)
        if sc.dirty:                 # restore the workspace protocol (zeroed tables, fresh parity)
            sc.ws.zero_()
            sc.stats.zero_()
            sc.dirty = False
        sc.seq += 1
        seq = sc.seq
        sc.dirty = True              # cleared once ff_merge_finish has been enqueued
        # first call of a prefill: hand the frame-major layout the prepare() scalars describe to the
        # similarity kernel, which derives and verifies the by-patch order itself (no K0 launch)
        hint = getattr(self, "_layout_hint", None) if (use_hint and not order_valid) else None
        if hint is not None and hint[0] + hint[1] * int(self.patch_num) > L:
            hint = None
        hint_pre, hint_frames = hint if hint is not None else (0, 0)
        rc = lib.ff_merge_begin(hidden.data_ptr(), addend_ptr, code, L, d, ptype.data_ptr(), int(self.patch_num), order_valid,
                                thr, sc.order.data_ptr(), sc.inv.data_ptr(), sim_ptr, sc.stats.data_ptr(), seq, hint_pre,
                                hint_frames,
                                sc.ws.data_ptr(), sc.ws_bytes, stream)
        _lib.check(rc, "ff_merge_begin")

        L_cap = L
        out = torch.empty(1, L_cap, d, dtype=dtype, device=device)
        ptype_out = torch.empty(1, L_cap, dtype=torch.int64, device=device)
        srcs, outs, rebuild = self._aux_for_positions(position_embeddings, L, L_cap)
        aux = (FFAux * _lib.MAX_AUX)()
        n_aux = self._fill_aux(aux, 0, [ptype.view(1, L)], [ptype_out], L)
        n_aux = self._fill_aux(aux, n_aux, srcs, outs, L)
        rc = lib.ff_merge_finish(hidden.data_ptr(), addend_ptr, out.data_ptr(), code, L, d, L_cap,
                                 float(thr), float(sub), float(self.ratio_lower_bound),
                                 sc.order.data_ptr(), sc.inv.data_ptr(), sim_ptr, sc.member.data_ptr(),
                                 sc.dst.data_ptr(), sc.keep.data_ptr(), sc.stats.data_ptr(), sc.stats_host_ptr, seq,
                                 aux, n_aux, sc.order_next.data_ptr(), sc.inv_next.data_ptr(), sc.ws.data_ptr(),
                                 sc.ws_bytes, stream)
        _lib.check(rc, "ff_merge_finish")
        sc.dirty = False
        mask_out = None
        if attention_mask is not None:
            mask_out = self._gather_mask(attention_mask, L, L_cap, sc.dst, stream)

        return dict(sc=sc, seq=seq, L=L, dtype=dtype, out=out, ptype_out=ptype_out, rebuild=rebuild,
                    mask_out=mask_out, hinted=hint is not None, residual=residual,
                    inputs=(hidden_states, position_embeddings, attention_mask))

    def _merge_complete(self, pending):
        sc, seq, L, dtype = pending["sc"], pending["seq"], pending["L"], pending["dtype"]
        out, ptype_out, rebuild, mask_out = pending["out"], pending["ptype_out"], pending["rebuild"], pending["mask_out"]
        attention_mask = None
        # The one device->host hand-off of the call: the scan kernel stores the result block into
        # pinned host memory (sequence word last) BEFORE the merge kernel runs, so the host learns
        # L_out while the second streaming pass is still in flight and returns without waiting for it.
        st = sc.wait_stats(seq)
        err = int(st[_lib.STAT_ERROR])
        if err & ~_lib.ERR_BIT_LAYOUT:
            sc.dirty = True
            sc.order_valid_for = None
            raise FrameFusionHipError(f"device-side check failed in the merge call (error bits {err:#x})")
        if err & _lib.ERR_BIT_LAYOUT:
            # patch_type is not the frame-major layout the prepare() scalars suggested (e.g. text
            # between the frames): everything this call enqueued is void.  Repeat it through K0 and
            # stop hinting for this prefill.
            assert pending["hinted"]
            self._layout_hint = None
            sc.dirty = True
            sc.order_valid_for = None
            return self._merge_complete(self._merge_launch(*pending["inputs"], use_hint=False, residual=pending["residual"]))
        nv, ftn, count = int(st[_lib.STAT_NV]), int(st[_lib.STAT_FTN]), int(st[_lib.STAT_COUNT])
        L_out = int(st[_lib.STAT_LOUT])
        branch = int(st[_lib.STAT_BRANCH])
        assert nv > 0, "no visual tokens"                                          # main.py:240

        above_k_ratio = count / ftn                                                 # main.py:114
        if branch == 0:                                                             # main.py:116-120
            self.sparsity_list.append(above_k_ratio)
            if above_k_ratio < self.ratio_lower_bound:
                self.finish_merging = True
        else:                                                                       # main.py:121-127
            self.finish_merging = True
            self.finish_pruning = True

        if L_out == L:
            # nothing was folded (empty threshold set, main.py:264-266): the merge kernel saw that on
            # the device and wrote nothing - the reduced sequence is the input itself, and the order
            # in the scratch still describes the (unchanged) patch_type
            self.last_call = dict(kind="merge", L_in=L, L_out=L, nv=nv, ftn=ftn, count=count, branch=branch,
                                  k=int(st[_lib.STAT_K]), scratch=sc, dtype=dtype, order=sc.order)
            sc.order_valid_for = (self._ptype_gen, L)
            if pending["residual"] is not None:          # nothing folded, but the caller is owed the sum
                h_in, pe_in, mask_in = pending["inputs"]
                return pending["residual"] + h_in, pe_in, mask_in
            return pending["inputs"]

        self.patch_type = ptype_out[:, :L_out]                                      # main.py:132
        # order maintenance: the merge kernel also wrote the by-patch order of the compacted
        # sequence, so the next merge call of this prefill skips K0
        self.last_call = dict(kind="merge", L_in=L, L_out=L_out, nv=nv, ftn=ftn, count=count, branch=branch,
                              k=int(st[_lib.STAT_K]), scratch=sc, dtype=dtype, order=sc.order)
        sc.order, sc.order_next = sc.order_next, sc.order
        sc.inv, sc.inv_next = sc.inv_next, sc.inv
        sc.order_valid_for = (self._ptype_gen, L_out)
        hidden_states = out[:, :L_out]
        position_embeddings = rebuild(L_out)
        if mask_out is not None:
            attention_mask = mask_out[:, :, :L_out, :L_out]
        return hidden_states, position_embeddings, attention_mask

    def last_plan(self):
        """Diagnostics of the most recent merge / prune call (views into the reusable scratch: valid
        until the next call): keep mask by sequence position, similarities, by-patch order, member flags."""
        c = self.last_call
        sc, L, nv = c["scratch"], c["L_in"], c["nv"]
        out = dict(keep=sc.keep[:L], member=sc.member[:L])
        if c["kind"] == "merge":
            out.update(sim=sc.sim(c["dtype"], nv), order=c["order"][:nv])
        return out

    def _expect_importance(self, S: int, dtype, device):
        """The attention hook is about to compute the importance of a prune call over S tokens: hand out the
        workspace (as the `select` argument of utils._launch_last_query) in which the importance kernel
        accumulates the select tables, and a token _prune recognises the tensor by."""
        with torch.cuda.device(device):
            sc = self._scratch_for(device, S, dtype)
            start = _to_int(self.image_token_start_index)
            n_img = _to_int(self.image_token_length - (self.original_length - S))
            if n_img < 0 or start < 0 or start + n_img > S:
                return None, None
            if sc.dirty:
                sc.ws.zero_()
                sc.stats.zero_()
            sc.dirty = True                       # until the prune call has consumed (and cleared) the tables
            token = (id(sc), sc.seq, S, start, n_img, dtype)
            sc.tables_token = token
            return (start, start + n_img, sc.ws.data_ptr(), sc.ws_bytes), token

    # ---- prune call: main.py:61-101 ----------------------------------------------------------------
    def _prune(self, hidden_states, position_embeddings, attention_mask, self_attn_weights, residual=None):
        _lib.require_gpu(hidden_states, "FrameFusion.forward")
        lib = _lib.load()
        bsz, q_len, d = hidden_states.size()
        assert bsz == 1, "Only support batch size 1"
        device, dtype = hidden_states.device, hidden_states.dtype
        code = _dtype_code(hidden_states)
        hidden = hidden_states.contiguous()
        start = _to_int(self.image_token_start_index)
        n_img = _to_int(self.image_token_length - (self.original_length - q_len))
        stream = _lib.stream_ptr()
        sc = self._scratch_for(device, q_len, dtype)

        w = self_attn_weights                                                       # main.py:69-70
        _lib.require_gpu(w, "FrameFusion.forward(self_attn_weights)")
        if w.ndim != 4 or w.shape[0] != 1 or w.shape[-1] != q_len:
            raise FrameFusionHipError(f"self_attn_weights of shape {tuple(w.shape)} is not [1, H, num, {q_len}]")
        # the head mean and the top-k run in the WEIGHTS' dtype (main.py:69-76: torch.mean / topk of the
        # tensor the attention hook handed over), whatever the activation dtype is
        w_code = _lib.DTYPE_CODE.get(w.dtype)
        if w_code is None:
            raise FrameFusionHipError(f"unsupported attention-weight dtype {w.dtype} (fp32 / bf16 / fp16 only)")
        w = w.contiguous()
        if w.data_ptr() & 15:
            w = w.clone()
        pruning_ratio = self._compute_pruning_ratio(self.sparsity_list, self.cost)  # main.py:73
        k = round(n_img * (1 - pruning_ratio))                                      # main.py:76
        if k < 0 or k > n_img:
            raise RuntimeError("selected index k out of range")                     # torch.topk's error
        L_out = q_len - n_img + k
        # importance from last_query_importance(..., framefusion=self): its kernel has already filled the
        # select tables of exactly this call in the workspace
        token = getattr(w, "_ff_tables", None)
        tables_ready = int(token is not None and token == getattr(sc, "tables_token", None) and
                           token == (id(sc), sc.seq, q_len, start, n_img, w.dtype) and w.shape[1] * w.shape[2] == 1)
        sc.tables_token = None
        if sc.dirty and not tables_ready:   # a call died half-way: the select tables must start from zero
            sc.ws.zero_()
            sc.stats.zero_()
            sc.dirty = False
        # nothing is read back (L_out is known): head mean + select tables, plan, gather - one host call
        out = torch.empty(1, L_out, d, dtype=dtype, device=device)
        srcs, outs, rebuild = self._aux_for_positions(position_embeddings, q_len, L_out)
        aux = (FFAux * _lib.MAX_AUX)()
        n_aux = self._fill_aux(aux, 0, srcs, outs, q_len)
        imp = sc.sim32[:q_len]       # fp32-sized slots: room for any weight dtype
        sc.dirty = True
        addend = residual.contiguous() if residual is not None else None
        _lib.check(lib.ff_prune_step(hidden.data_ptr(), addend.data_ptr() if addend is not None else None,
                                     out.data_ptr(), code, q_len, d, L_out,
                                     w.data_ptr(), w_code, w.shape[1], w.shape[2], imp.data_ptr(), tables_ready,
                                     start, n_img, k, sc.member.data_ptr(), sc.dst.data_ptr(), sc.keep.data_ptr(),
                                     sc.stats.data_ptr(), aux, n_aux, sc.ws.data_ptr(), sc.ws_bytes, stream),
                   "ff_prune_step")
        sc.dirty = False
        if attention_mask is not None:
            attention_mask = self._gather_mask(attention_mask, q_len, L_out, sc.dst, stream)
        self.finish_pruning = True                                                  # main.py:101
        self.last_call = dict(kind="prune", L_in=q_len, L_out=L_out, k=k, nv=q_len, scratch=sc, dtype=w.dtype)
        return out, rebuild(L_out), attention_mask

    # ---- static parity entry points ---------------------------------------------------------------
    @staticmethod
    def compute_similarity_and_token_index_by_patch(hidden_states, token_patch_type, patch_num):
        """main.py:180-241 -> (similarity_by_patch [1, Nv] act dtype, token_index_by_patch [1, Nv] int64)."""
        _lib.require_gpu(hidden_states, "compute_similarity_and_token_index_by_patch")
        lib = _lib.load()
        bsz, L, d = hidden_states.size()
        assert bsz == 1, "Only support batch size 1"
        device, dtype = hidden_states.device, hidden_states.dtype
        code = _dtype_code(hidden_states)
        hidden = hidden_states.contiguous()
        ptype = token_patch_type.to(device=device, dtype=torch.int64).contiguous()
        order = torch.empty(L, dtype=torch.int32, device=device)
        sim = torch.empty(L, dtype=dtype, device=device)
        stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=device)
        ws_bytes = int(lib.ff_workspace_bytes(L, int(patch_num)))
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=device)
        stream = _lib.stream_ptr()
        _lib.check(lib.ff_build_order(ptype.data_ptr(), L, int(patch_num), order.data_ptr(), None, stats.data_ptr(),
                                      ws.data_ptr(), ws_bytes, stream), "ff_build_order")
        _lib.check(lib.ff_pair_similarity(hidden.data_ptr(), code, L, d, ptype.data_ptr(), order.data_ptr(),
                                          stats.data_ptr(), sim.data_ptr(), stream), "ff_pair_similarity")
        nv = int(stats[_lib.STAT_NV])
        assert nv > 0, "no visual tokens"
        return sim[None, :nv], order[None, :nv].long()

    @staticmethod
    def merge_tokens_and_get_mask(hidden_states, similarity_by_patch, token_index_by_patch, merge_index_by_patch):
        """main.py:243-319: folds the runs of `merge_index_by_patch` into their anchors IN PLACE
        (as the reference does) and returns (hidden_states, keep_mask [1, L] bool)."""
        _lib.require_gpu(hidden_states, "merge_tokens_and_get_mask")
        lib = _lib.load()
        bsz, L, d = hidden_states.size()
        device, dtype = hidden_states.device, hidden_states.dtype
        if merge_index_by_patch.shape[0] == 0:
            return hidden_states, torch.ones(hidden_states.shape[:-1], dtype=torch.bool, device=device)
        assert bsz == 1, "Only support batch size 1"
        code = _dtype_code(hidden_states)
        nv = token_index_by_patch.shape[1]
        order32 = token_index_by_patch[0].to(torch.int32)
        # complete the permutation with the positions that are not in the by-patch order
        rest = torch.ones(L, dtype=torch.bool, device=device)
        rest[token_index_by_patch[0]] = False
        order = torch.cat((order32, torch.nonzero(rest).reshape(-1).to(torch.int32))).contiguous()
        midx = merge_index_by_patch.to(device=device, dtype=torch.int64).contiguous()
        member = torch.empty(L, dtype=torch.uint8, device=device)
        dst = torch.empty(L, dtype=torch.int32, device=device)
        keep = torch.empty(L, dtype=torch.uint8, device=device)
        stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=device)
        stats[_lib.STAT_NV] = nv
        ws_bytes = int(lib.ff_workspace_bytes(L, 1))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        stream = _lib.stream_ptr()
        _lib.check(lib.ff_plan_from_index(midx.data_ptr(), midx.numel(), order.data_ptr(), L, member.data_ptr(),
                                          dst.data_ptr(), keep.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                          ws_bytes, stream), "ff_plan_from_index")
        hidden = hidden_states.contiguous()
        compact = torch.empty(L, d, dtype=dtype, device=device)
        _lib.check(lib.ff_merge_compact(hidden.data_ptr(), compact.data_ptr(), code, L, d, L, order.data_ptr(),
                                        member.data_ptr(), 1, dst.data_ptr(), None, None, 0, stream),
                   "ff_merge_compact")
        keep_b = keep.bool()
        kept = torch.nonzero(keep_b).reshape(-1)
        hidden_states[0, kept] = compact[: kept.numel()]      # members keep their old rows, like the reference
        return hidden_states, keep_b[None, :]

    @staticmethod
    def _compute_pruning_ratio(sparsity_list, cost, num_layers=28):
        """main.py:321-343 - python floats, same operation order (bit-identical budget)."""
        kept = 1
        spent = 0
        for sp in sparsity_list:
            kept *= (1 - sp)
            spent += kept
        remaining = num_layers * cost - spent
        if remaining < 0:
            raise ValueError("The cost is too small")
        share = remaining / ((num_layers - len(sparsity_list)) * kept)
        if share > 1:
            return 0
        return 1 - share


def cosine_similarity(mat1, mat2):
    """main.py:345-349 for [..., d] operands, evaluated by the pair-similarity kernel: rows are
    interleaved (a0, b0, a1, b1, ...) with patch types (0, 0, 1, 1, ...) so pair k sits at by-patch
    position 2k+1."""
    _lib.require_gpu(mat1, "cosine_similarity")
    lead = mat1.shape[:-1]
    d = mat1.shape[-1]
    a = mat1.reshape(-1, d)
    b = mat2.expand_as(mat1).reshape(-1, d)
    n = a.shape[0]
    rows = torch.stack((a, b), dim=1).reshape(1, 2 * n, d).contiguous()
    ptype = torch.arange(n, device=mat1.device).repeat_interleave(2)[None]
    sim, _ = FrameFusion.compute_similarity_and_token_index_by_patch(rows, ptype, n)
    return sim[0, 1::2].reshape(lead)


def find_contigious_latter_index(index_tensor: torch.Tensor) -> torch.Tensor:
    """main.py:351-380 (API compatibility helper; the hot path uses the plan kernel instead):
    0 1 1 1 0 0 1 1 -> 0 0 0 3 0 0 0 2.  Pure integer index arithmetic, any device."""
    ones = index_tensor == 1
    pad = torch.zeros_like(ones[:, :1])
    rises = ones & ~torch.cat((pad, ones[:, :-1]), dim=1)
    falls = ones & ~torch.cat((ones[:, 1:], pad), dim=1)
    r = torch.nonzero(rises, as_tuple=True)
    f = torch.nonzero(falls, as_tuple=True)
    out = torch.zeros_like(index_tensor)
    out[f[0], f[1]] = (f[1] - r[1] + 1).to(index_tensor.dtype)
    return out
