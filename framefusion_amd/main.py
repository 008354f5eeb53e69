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
import struct
from typing import List, Optional

import torch
from torch import nn

from . import _lib
from ._lib import FrameFusionHipError
from .utils import LastQuery

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


_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_PACK_PTR = struct.Struct("=Q")
_PACK_I64 = struct.Struct("=q")


class _Scratch:
    """Per-device scratch of one instance (order, member, dst, keep, stats, select tables) named by an
    ``ff_ctx_t``, plus the reusable call / result blocks: a forward call crosses the C ABI with one
    pointer to each (include/framefusion_hip.h, "call context")."""

    def __init__(self, device):
        self.device = device
        self.cap = 0
        self.ctx = _lib.FFCtx()
        self.ctx_ptr = C.addressof(self.ctx)
        self.call = _lib.FFMergeCall()
        self.call_ptr = C.addressof(self.call)
        self.pcall = _lib.FFPruneCall()
        self.pcall_ptr = C.addressof(self.pcall)
        self.res = _lib.FFMergeResult()
        self.res_ptr = C.addressof(self.res)
        self.lqargs = _lib.FFLqArgs()
        self.lqargs_ptr = C.addressof(self.lqargs)
        self.lq_ws = None               # workspace of the importance kernels (grow-only; prune_from_qk)
        self.order_gen = None           # patch_type generation the context's by-patch order belongs to
        self.tables_token = None
        self.last_stream = None
        self.last_stream_ptr = None

    _BUFFERS = ("order", "order_next", "inv", "inv_next", "member", "dst", "keep", "sim32", "ws", "stats")

    def __del__(self):
        ptr = getattr(self, "stats_host_ptr", None)
        if ptr:
            self.stats_host_ptr = None
            try:
                if self.last_stream is not None:
                    self.last_stream.synchronize()       # (a kernel that still publishes into the block must have finished)
                _lib.load().ff_host_free(ptr)
            except Exception:                            # noqa: BLE001 (interpreter shutdown)
                pass

    def ensure(self, L: int):
        if L > self.cap:
            cap = (max(L, 1024) + 63) & ~63          # (whole 16-byte words behind every int32 / T array)
            dev = self.device
            if self.cap and self.last_stream is not None:
                # kernels of the stream that used the old buffers last may still read them: keep the
                # allocator from handing the memory out before that stream gets there
                for name in self._BUFFERS:
                    getattr(self, name).record_stream(self.last_stream)
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
            # the result block (device -> host) and the output mail (host -> device, read by a RUNNING kernel): coherent pinned
            # memory from the library's own runtime (torch's pin_memory() is only guaranteed visible at kernel boundaries)
            if getattr(self, "stats_host_ptr", None):
                pass                                             # (one block per scratch: its size does not depend on the capacity)
            else:
                self.stats_host_ptr = _lib.load().ff_host_alloc(_lib.HOST_WORDS * 8)
                if not self.stats_host_ptr:
                    raise FrameFusionHipError("ff_host_alloc failed (pinned, coherent host memory for the result block)")
                import numpy as np
                self.stats_np = np.ctypeslib.as_array((C.c_int64 * _lib.HOST_WORDS).from_address(self.stats_host_ptr))
                self.stats_host = torch.from_numpy(self.stats_np)
            c = self.ctx
            c.cap = cap
            c.order, c.order_next = self.order.data_ptr(), self.order_next.data_ptr()
            c.inv, c.inv_next = self.inv.data_ptr(), self.inv_next.data_ptr()
            c.sim, c.member, c.dst, c.keep = (self.sim32.data_ptr(), self.member.data_ptr(), self.dst.data_ptr(),
                                              self.keep.data_ptr())
            c.stats, c.stats_host = self.stats.data_ptr(), self.stats_host_ptr
            c.ws, c.ws_bytes = self.ws.data_ptr(), self.ws_bytes
            c.order_len, c.dirty, c.in_flight, c.swaps = 0, 0, 0, 0
            self._swaps = 0
            self.cap = cap
            self.order_gen = None
        return self

    # ---- views for tests / tools ----
    @property
    def dirty(self) -> bool:
        return bool(self.ctx.dirty)

    @property
    def seq(self) -> int:
        return int(self.ctx.seq)

    @property
    def order_valid_for(self):
        """(patch_type generation, sequence length) the maintained by-patch order describes, or None."""
        n = int(self.ctx.order_len)
        return (self.order_gen, n) if (n > 0 and self.order_gen is not None) else None

    def forget_order(self):
        self.order_gen = None

    def sim(self, dtype, n):
        return self.sim32.view(dtype)[:n]

    def sync_views(self):
        """The library exchanges order <-> order_next (and the inverses) inside the context when a merge
        call shortened the sequence: mirror it on the tensors that own the memory."""
        if (int(self.ctx.swaps) - self._swaps) & 1:
            self.order, self.order_next = self.order_next, self.order
            self.inv, self.inv_next = self.inv_next, self.inv
        self._swaps = int(self.ctx.swaps)

    # ---- the call blocks ----
    def put_aux(self, block, offset, pairs, L, room=_lib.MAX_AUX):
        """Describe (src, out) tensor pairs as ff_aux_t entries; returns their number.  Sources may be `_token_dense` views
        (the leading dims one uniform stride apart): the stride travels as src_outer_bytes."""
        n = 0
        pack = _lib.AUX_ENTRY.pack_into
        size = _lib.AUX_ENTRY.size
        for s, o in pairs:
            if n >= room:
                raise FrameFusionHipError("too many auxiliary tensors")
            if type(s) is _Strided:       # (its strides were looked at a moment ago, by _token_dense, in this very call)
                s, known = s.t, s.outer_bytes
            else:
                known = None
            if s.ndim == 2:         # [B, L] ids
                row, outer = s.element_size(), s.shape[0]
                outer_bytes = 0
            else:                   # [..., L, dh]
                last = s.shape[-1]
                row = last * s.element_size()
                outer = s.numel() // (L * last)
                # (from the strides of THIS call, never from an attribute a caller's tensor may carry from an earlier life)
                outer_bytes = known if known is not None else (0 if s.is_contiguous() else _outer_stride(s) * s.element_size())
            pack(block, offset + size * n, s.data_ptr(), o.data_ptr() if o is not None else 0, row, outer, outer_bytes)
            n += 1
        return n


class _Strided:
    """A [..., L, dh] tensor the merge kernel reads in place although it is not contiguous, with the byte stride between its outer
    slices - what `_token_dense` found out, handed to `put_aux` so that the strides are looked at once per call."""
    __slots__ = ("t", "outer_bytes")

    def __init__(self, t, outer_bytes):
        self.t, self.outer_bytes = t, outer_bytes


def _token_dense(t: torch.Tensor):
    """`t` itself when it is contiguous; `_Strided(t, stride)` when the merge kernel can still read it in place (rows dense and
    the leading dims one uniform stride apart: `_outer_stride`); else a contiguous copy."""
    if t.is_contiguous():
        return t
    st = _outer_stride(t)
    if st is not None:
        return _Strided(t, st * t.element_size())
    return t.contiguous()


def _outer_stride(t: torch.Tensor):
    """Element stride between consecutive slices of the leading dims of a [..., L, dh] tensor whose rows are dense
    (stride 1 / dh on the last two axes) when those leading dims are ONE uniform stride apart - None otherwise.  True for
    contiguous tensors and for the [3, 1, L_out, dh] views a merge call returns for M-RoPE tables (three planes, L_cap rows
    apart), which therefore go into the next call without a copy."""
    shape, strides = t.shape, t.stride()
    nd = len(shape)
    if nd < 3 or strides[-1] != 1 or strides[-2] != shape[-1]:
        return None
    stride = None
    expect = None                       # stride the next-outer non-trivial dim must have
    for k in range(nd - 3, -1, -1):
        if shape[k] == 1:
            continue
        if stride is None:
            stride = strides[k]
            expect = stride * shape[k]
        elif strides[k] != expect:
            return None
        else:
            expect *= shape[k]
    if stride is None:
        return shape[-2] * shape[-1]
    # a broadcast (stride 0) or overlapping leading dim has no plane to read in place: the caller copies
    return stride if stride >= shape[-2] * shape[-1] else None


def _fail(rc: int, what: str, err_bits: int = 0):
    if rc == _lib.ERR_DEVICE:
        if err_bits:
            raise FrameFusionHipError(f"device-side check failed in the {what} call (error bits {err_bits:#x})")
        raise FrameFusionHipError(f"the device never published the result block of this {what} call")
    _lib.check(rc, what)


class FrameFusion(nn.Module):
    def __init__(self, cost=0.3, similarity_lower_bound=0.6, ratio_lower_bound=0.1, compact_outputs=True):
        super().__init__()
        self.cost = cost
        self.similarity_lower_bound = similarity_lower_bound
        self.ratio_lower_bound = ratio_lower_bound
        # compact_outputs = True (the default since round 5): exactly sized outputs, what the reference returns
        # (main.py:132-138).  They are allocated for the length the top-k branch gives (host arithmetic) while the kernels run
        # and the merge kernel goes out blind into them, guarded on the device; if the plan decides otherwise the host sizes
        # them to the l_out it reads from the result block (the one-launch kernel waits for that with the rows in registers,
        # the three launches repeat the merge kernel).  Nothing is over-allocated, nothing is copied.  False: buffers of the
        # INPUT length, narrow() views of them are returned (a view keeps its whole buffer alive: 302 MB for a 91 MB result
        # at 64 x 576 x 4096): never a repeat.  End to end the two differ by less than +-0.35 %
        # (profiles/r05_e2e_prefill.json).  INTEGRATION.md, "Output buffers".
        self.compact_outputs = compact_outputs
        self._scratch = {}
        self._ptype_gen = 0       # bumped whenever patch_type is (re)assigned: keys the cached by-patch order
        self._host_ints = {}      # id(0-d device tensor) -> (tensor, int): the prepare() scalars, read back once
        self.last_call = None     # diagnostics of the most recent reduction (tests / bench)

    _PLAIN = (bool, int, float, str, list, tuple, dict, type(None), torch.Tensor)
    supports_residual = True      # forward(..., residual=) exists: see call_b_with_residual()
    accepts_last_query = True     # forward(..., self_attn_weights=LastQuery) exists: see utils.last_query_importance(defer=True)

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
        state["_host_ints"] = {}
        state["last_call"] = None
        state["_ticket"] = None
        state["_bad_hints"] = dict(self.__dict__.get("_bad_hints", {}))
        state["_guess_held"] = dict(self.__dict__.get("_guess_held", {}))
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
        d = self.__dict__
        d["_ptype_gen"] = d.get("_ptype_gen", 0) + 1
        d["patch_type"] = patch_type
        d["patch_num"] = patch_num
        d["image_token_start_index"] = image_token_start_index
        d["image_token_end_index"] = image_token_end_index
        d["image_token_length"] = image_token_length
        d["original_length"] = original_length
        d["finish_merging"] = finish_merging
        d["finish_pruning"] = finish_pruning
        d["sparsity_list"] = [] if sparsity_list is None else sparsity_list
        d["_host_ints"] = {}
        d["_layout_hint"] = self._frame_major_hint(patch_num, image_token_start_index, image_token_length)

    def _host_int(self, x):
        """int(x) for the prepare() scalars.  The reference's packers hand them over as 0-d / 1-element
        DEVICE tensors (llava_video/modeling_llava_video.py:332: ``torch.where(...)[0]``) right after
        having synchronised on them themselves (``[TEXT_TOKEN] * image_token_start_index``, :335); the
        reference reads them back in every prune call (main.py:64-66).  Here each is read back once per
        prefill and remembered."""
        if not isinstance(x, torch.Tensor):
            return int(x)
        hit = self._host_ints.get(id(x))
        if hit is not None and hit[0] is x:
            return hit[1]
        v = int(x.item())
        self._host_ints[id(x)] = (x, v)
        return v

    def _frame_major_hint(self, patch_num, start, length):
        """(pre, frames) if the scalars describe whole frames of `patch_num` tokens starting at `start`
        (the layout every packer of the reference builds), else None.  The similarity kernel verifies
        the hint against patch_type."""
        def whole(x):
            if isinstance(x, torch.Tensor):
                if x.numel() != 1 or x.is_floating_point() or x.dtype == torch.bool:
                    return None
                return self._host_int(x)
            try:
                return int(x) if float(x) == int(x) else None
            except (TypeError, ValueError):
                return None
        P, pre, n = whole(patch_num), whole(start), whole(length)
        if P is None or pre is None or n is None or P < 1 or pre < 0 or n < P or n % P:
            return None
        bad = self.__dict__.get("_bad_hints")
        if bad and (P, pre, n // P) in bad:                   # (found wrong on the device in an earlier prefill)
            bad[(P, pre, n // P)] -= 1                         # ... but not for ever: after 8 prefills the hint gets another chance
            if bad[(P, pre, n // P)] <= 0:
                del bad[(P, pre, n // P)]
            return None
        return pre, n // P

    # ---- reference main.py:40-140 ------------------------------------------------------------
    def forward(self, hidden_states, position_embeddings, attention_mask, self_attn_weights=None, residual=None):
        """Reference signature (main.py:40-42) plus `residual`: when given, the sequence that is reduced
        is ``residual + hidden_states`` - the add the decoder performs right before call B
        (models/qwen2/modeling_qwen2.py:64-67) - formed inside the two streaming passes instead of by an
        eager add whose result would be written once and read twice."""
        dev = hidden_states.device
        if dev.type == "cuda" and dev.index != _get_device():
            # the kernels are launched through ctypes on the CURRENT device's stream: follow the
            # tensors (several replicas on several GPUs in one process, as in the reference's demo)
            with torch.cuda.device(dev):
                return self.forward(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)
        if self.__dict__.get("_ticket") is not None:
            raise FrameFusionHipError("forward(): a submitted call of this instance has not been collected")
        bsz, q_len, hidden_size = hidden_states.size()
        prune_now = q_len > 1 and self.finish_merging == True and self.finish_pruning == False
        merge_now = q_len > 1 and (not self.finish_merging)
        if (prune_now or merge_now) and (hidden_size * hidden_states.element_size()) & 15:
            # rows that are not whole 16-byte words (no model the reference supports has them; its torch path does not care):
            # zero columns up to the next word change neither a dot product, a norm nor a fold - pad, reduce, cut them off again
            pad = (-hidden_size) % (16 // hidden_states.element_size())
            out, position_embeddings, attention_mask = self.forward(
                torch.nn.functional.pad(hidden_states, (0, pad)), position_embeddings, attention_mask, self_attn_weights,
                None if residual is None else torch.nn.functional.pad(residual, (0, pad)))
            return out[..., :hidden_size].contiguous(), position_embeddings, attention_mask
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
        if self.__dict__.get("_thr_cache_key") != key:
            self._thr_cache_key, self._thr_cache = key, _round_to(self.similarity_lower_bound, dtype)
        return self._thr_cache

    def _scratch_for(self, device, L) -> _Scratch:
        """The scratch of `device` sized for L tokens, and the raw hipStream_t of PyTorch's current stream."""
        key = (device.type, device.index)
        s = self._scratch.get(key)
        if s is None:
            s = self._scratch[key] = _Scratch(device)
        # the scratch is reused call after call: if the caller switched streams, order the new
        # stream behind the one that last touched it
        ptr = _lib.stream_ptr()
        if s.last_stream_ptr != ptr:
            cur = torch.cuda.current_stream(device)
            last = s.last_stream
            if last is not None and last != cur:
                cur.wait_stream(last)
            s.last_stream, s.last_stream_ptr = cur, ptr
        if L > s.cap:
            s.ensure(L)
        return s, ptr

    @staticmethod
    def _position_outputs(position_embeddings, L_cap: int):
        """The OUTPUT half of `_aux_for_positions`: (outputs, rebuild(L_out) -> new container) for a container that has been
        looked at already (a merge call whose outputs are sized a second time, to the l_out of the result block)."""
        if type(position_embeddings) == list:
            a, b = position_embeddings
            shape = a.shape
            out_shape = shape[:-2] + (L_cap, shape[-1])
            if b.shape == shape and b.dtype == a.dtype:
                both = torch.empty((2,) + out_shape, dtype=a.dtype, device=a.device)     # cos and sin: one allocation
                outs = [both[0], both[1]]
            else:
                outs = [torch.empty(out_shape, dtype=a.dtype, device=a.device),
                        torch.empty(b.shape[:-2] + (L_cap, b.shape[-1]), dtype=b.dtype, device=b.device)]
            ax = len(shape) - 2

            def rebuild(L_out):
                for x in (0, 1):
                    position_embeddings[x] = outs[x].narrow(ax, 0, L_out)
                return position_embeddings
            return outs, rebuild
        t = position_embeddings
        out = torch.empty(t.shape[0], L_cap, dtype=t.dtype, device=t.device)
        return [out], (lambda L_out: out.narrow(1, 0, L_out))

    @staticmethod
    def _aux_for_positions(position_embeddings, L: int, L_cap: int):
        """Describe the position container for K4 (main.py:142-178): returns
        (sources, outputs, rebuild(L_out) -> new container)."""
        if type(position_embeddings) == list:
            assert len(position_embeddings) == 2
            a, b = position_embeddings
            for t in (a, b):
                if t.ndim not in (3, 4) or t.shape[-2] != L:
                    raise FrameFusionHipError(f"position embedding of shape {tuple(t.shape)} does not have "
                                              f"{L} tokens on its second-to-last axis")
            src_a, src_b = _token_dense(a), _token_dense(b)       # (tensors, or _Strided wrappers of a and b themselves)
            outs, rebuild = FrameFusion._position_outputs(position_embeddings, L_cap)
            return [src_a, src_b], outs, rebuild
        if type(position_embeddings) == torch.Tensor:
            if position_embeddings.ndim != 2:
                raise NotImplementedError("Only support 2D position embeddings")
            outs, rebuild = FrameFusion._position_outputs(position_embeddings, L_cap)
            return [position_embeddings.contiguous()], outs, rebuild
        raise NotImplementedError("Only support list or tensor for position embeddings")

    @staticmethod
    def _mask_for(attention_mask, L):
        m = attention_mask
        if m.ndim != 4 or m.shape[0] != 1 or m.shape[1] != 1 or m.shape[2] != L or m.shape[3] != L:
            raise FrameFusionHipError(f"attention mask of shape {tuple(m.shape)} is not [1, 1, {L}, {L}]")
        m = m.contiguous()
        if m.data_ptr() & 15:
            m = m.clone()
        return m

    # ---- merge call: main.py:104-138 -------------------------------------------------------------
    def _merge(self, hidden_states, position_embeddings, attention_mask, residual=None):
        """One merge call = two crossings of the C ABI: ff_ctx_merge_begin enqueues the similarity pass
        before any output tensor exists (the allocations below overlap it), ff_ctx_merge_finish enqueues
        plan + merge and waits - in C, interpreter lock released - for the 256-byte result block the plan
        kernel publishes before the second streaming pass starts."""
        lib = _lib.load()
        st = self._merge_prepare(hidden_states, position_embeddings, attention_mask, residual)
        sc = st["sc"]
        one = bool(self.one_launch and lib.ff_ctx_merge_one_launch(sc.ctx_ptr, sc.call_ptr))
        guess = st.get("L_guess")
        # Does the guessed length (the top-k branch's) usually come true for THIS call of a prefill?  The threshold branch runs
        # for every merge call of a prefill but the last, and a blind merge kernel that finds its buffers too short costs a
        # wasted launch (7.5 us at 37 k tokens) on top of the repeat: a call index whose guess failed in the previous prefill goes
        # plan -> wait -> outputs -> merge kernel instead (no wasted launch).  The one-launch kernel does not care (second mail).
        st["guess_idx"] = idx = len(self.sparsity_list)
        trust = self.__dict__.get("_guess_held", {}).get(idx, True)
        if one or (self.compact_outputs and guess is not None and 0 < guess < st["L"] and trust):
            # everything goes out in ONE crossing: the one-launch kernel when the activation fits on the chip; else the three
            # launches with the merge kernel blind into outputs of the guessed length (no plan -> host -> merge kernel bubble)
            st["one_launch"] = one
            st["flow"] = "submit"
            return self._merge_submitted(st)
        rc = lib.ff_ctx_merge_begin(sc.ctx_ptr, sc.call_ptr)
        if rc:
            _fail(rc, "merge")
        if self.compact_outputs:
            # exactly sized outputs (the default) and no length to guess: the plan goes out behind K1, the host waits for l_out,
            # sizes the outputs to it and only then enqueues the merge kernel - no input-length buffers, no copy; the GPU idles
            # for the host's reaction time between the two kernels instead
            st["flow"] = "wait"
            return self._merge_exact_tail(st)
        st["flow"] = "finish"
        self._merge_outputs(st)
        # The one device->host hand-off of the call: the plan kernel stores the result block into pinned
        # host memory (sequence word last) BEFORE the merge kernel runs, so the host learns L_out while the
        # second streaming pass is still in flight and returns without waiting for it.
        rc = lib.ff_ctx_merge_finish(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
        return self._merge_complete(st, rc)

    # The one-launch kernel (csrc/ff_resident.hip): the activation fits into the chip's registers + LDS (the LLaVA-Video-7B and
    # Qwen2-VL-7B prefills do) and is read ONCE.  False: always the three launches (A/B measurements, tests).
    one_launch = True
    # Two more forms the C ABI offers and this host does not use by default (measured slower HERE, profiles/EXPERIMENTS.md
    # 5.5 / 5.6; both are what a host without a cheap allocator or with cheap call preparation would pick): the attention mask
    # gathered by the merge call itself into a buffer the caller passes in the call block, and hook + prune as ONE crossing
    # (ff_ctx_prune_from_qk) at the end of the prune call's preparation.  Public switches: the tests run both.
    mask_through_call = False
    prune_in_one_crossing = False

    def _merge_submitted(self, st):
        """Outputs first, then ONE crossing that enqueues the whole call (ff_ctx_merge_submit: the one-launch kernel, or K1 + plan +
        merge kernel), then the wait (ff_ctx_merge_collect).  Exactly sized outputs (the default) are allocated for the length the
        top-k branch gives (main.py:122, host arithmetic) and the merge kernel / phase goes out BLIND into them, guarded on the
        device: if the plan decides otherwise - or no length can be guessed - the three launches write nothing (`applied` = 0) and
        the merge kernel follows alone, into outputs of the length the result block names; the one-launch kernel, which publishes
        that length before its plan is through, waits for them rows in hand (`applied` = 2; second mail slot) - no second launch
        (ff_ctx_merge_apply either way)."""
        lib = _lib.load()
        sc, L = st["sc"], st["L"]

        def outputs():
            if self.compact_outputs:
                guess = st.get("L_guess")
                if guess is not None and 0 < guess < L:
                    self._merge_outputs(st, guess)
                else:
                    self._no_outputs(st)                 # (only the one-launch kernel gets here: it stops behind its plan)
            else:
                self._merge_outputs(st)
        late = 1 if st.get("one_launch") else 0
        _PACK_I64.pack_into(sc.call, _lib.MERGE_CALL_LATE_OFFSET, late)
        if late:
            # the one-launch kernel needs its outputs - and the position tables / patch types that are gathered with them -
            # ~35 us after it starts: launch first, describe and allocate under it, hand the lot over in pinned memory
            # (ff_ctx_merge_mail, slot 1: plain stores, written before the wait below begins)
            self._no_outputs(st)
            rc = lib.ff_ctx_merge_submit(sc.ctx_ptr, sc.call_ptr)
            if rc:
                _fail(rc, "merge")
            outputs()
            rc = lib.ff_ctx_merge_mail(sc.ctx_ptr, sc.call_ptr)
            if rc:
                _fail(rc, "merge")
        else:
            outputs()
            rc = lib.ff_ctx_merge_submit(sc.ctx_ptr, sc.call_ptr)
            if rc:
                _fail(rc, "merge")
        rc = lib.ff_ctx_merge_collect(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
        if rc or sc.res.applied == 1:
            return self._merge_complete(st, rc)
        # applied = 2: the one-launch kernel published l_out right behind its grid barrier and now waits, rows in hand, for
        # outputs that hold it - the mailed ones if the guess came true, else the ones allocated here (second mail slot, written
        # by ff_ctx_merge_apply, which also reads the kernel's acknowledgement: no launch).  applied = 0: the three launches'
        # blind merge kernel wrote nothing (the plan decided otherwise): ff_ctx_merge_apply repeats it into the right outputs.
        L_out = int(sc.res.l_out)
        if L_out == L:
            self._no_outputs(st, hidden=st["hidden"])       # nothing folds: the launch only clears the select tables
        elif st.get("L_cap") != L and st.get("L_cap") != L_out:
            self._merge_outputs(st, L_out)
            if sc.res.applied == 2:
                # the kernel is waiting for exactly these: second mail slot, NOW - every microsecond until it is written is one
                # the whole chip spends spinning (ff_ctx_merge_apply below would mail them too, after the bookkeeping)
                rc = lib.ff_ctx_merge_mail(sc.ctx_ptr, sc.call_ptr)
                if rc:
                    _fail(rc, "merge")
        # (the host's own bookkeeping of the call comes first: the kernel's answer to the mail is two PCIe round trips away)
        return self._merge_complete(st, 0, confirm=lambda: lib.ff_ctx_merge_apply(sc.ctx_ptr, sc.call_ptr, sc.res_ptr))

    def _no_outputs(self, st, hidden=None):
        """The output half of the call block, empty: a plan-only submit (hidden = None) / an apply that writes nothing."""
        call, L = st["sc"].call, st["L"]
        _PACK_PTR.pack_into(call, 16, hidden.data_ptr() if hidden is not None else 0)
        _PACK_I64.pack_into(call, 56, L if hidden is not None else 0)             # ff_merge_call_t.L_cap
        _PACK_I64.pack_into(call, _lib.MERGE_CALL_AUX_OFFSET - 8, 0)
        _lib.MASK_TRIPLE.pack_into(call, _lib.MERGE_CALL_MASK_OFFSET, 0, 0, 0)
        st.update(mask_cap=None, L_cap=L if hidden is not None else 0)

    # ---- the same call in two halves (FrameFusionPair): everything enqueued by submit, the wait in collect ------------------------
    def submit(self, hidden_states, position_embeddings, attention_mask, self_attn_weights=None, residual=None):
        """``forward`` without its wait: a merge call is enqueued in full (similarity, plan, merge: one crossing of the C ABI,
        ``ff_ctx_merge_submit``) on PyTorch's current stream and a ticket comes back at once; ``collect(ticket)`` waits for the
        result block and returns what ``forward`` would have.  Any other call (prune, nothing to do) runs at once and the ticket
        carries its result.  Between submit and collect the instance must not be touched (no prepare(), no other call)."""
        dev = hidden_states.device
        if dev.type == "cuda" and dev.index != _get_device():
            with torch.cuda.device(dev):
                return self.submit(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)
        if self.__dict__.get("_ticket") is not None:
            raise FrameFusionHipError("submit(): the previous call of this instance has not been collected")
        q_len = hidden_states.shape[1]
        if not (q_len > 1 and not self.finish_merging) or (hidden_states.shape[2] * hidden_states.element_size()) & 15:
            return {"done": self.forward(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)}
        st = self._merge_prepare(hidden_states, position_embeddings, attention_mask, residual)
        sc = st["sc"]
        lib = _lib.load()
        if self.compact_outputs:                 # exactly sized outputs: K1 now; plan, outputs and merge kernel at collect()
            rc = lib.ff_ctx_merge_begin(sc.ctx_ptr, sc.call_ptr)        # (the plan follows with the wait, at collect())
            st["exact"] = True
            guess = st.get("L_guess")
            if not rc and guess is not None and 0 < guess < st["L"]:     # the top-k branch's outputs, allocated under K1
                self._merge_outputs(st, guess)
                st["guessed"] = True
        else:
            self._merge_outputs(st)
            # (two samples in flight: never the one-launch kernel - it needs every CU until its grid barrier, two of them
            # side by side wait for each other until one gives up)
            sc.ctx.res_off = max(int(sc.ctx.res_off), 2)
            rc = lib.ff_ctx_merge_submit(sc.ctx_ptr, sc.call_ptr)
        if rc:
            _fail(rc, "merge")
        self._ticket = st
        return st

    def collect(self, ticket):
        if "done" in ticket:
            return ticket["done"]
        if self.__dict__.get("_ticket") is not ticket:
            raise FrameFusionHipError("collect(): not the ticket of this instance's call in flight")
        self._ticket = None
        sc = ticket["sc"]
        with torch.cuda.device(ticket["device"]):
            if ticket.get("exact"):
                return self._merge_exact_tail(ticket)
            rc = _lib.load().ff_ctx_merge_collect(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
            return self._merge_complete(ticket, rc)

    def _merge_exact_tail(self, st):
        """wait for the result block -> outputs of exactly l_out rows -> merge kernel (ff_ctx_merge_wait / _apply)."""
        lib = _lib.load()
        sc = st["sc"]
        # While K1 runs: outputs for the length the top-k branch would give (L - int(sub * ftn), main.py:122 - host arithmetic
        # when the number of non-text tokens is known: from prepare()'s layout scalars, or from the previous call of the
        # prefill).  If the plan decides that way the merge kernel goes out the moment the result is seen; if not (threshold
        # branch) the guess costs one allocation that was made under the similarity pass.
        guess = st.get("L_guess")
        if guess is not None and 0 < guess < st["L"]:
            if not st.get("guessed"):                                # (submit() has done it already)
                self._merge_outputs(st, guess)
        else:
            guess = None
        rc = lib.ff_ctx_merge_wait(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
        if rc:
            return self._merge_complete(st, rc)                      # (raises)
        L, L_out = st["L"], int(sc.res.l_out)
        if L_out == guess:
            pass                                                     # (the call block already describes the right outputs)
        elif L_out == L:
            # nothing folds: the launch only clears the select tables; no output is written (any valid pointer will do)
            self._no_outputs(st, hidden=st["hidden"])
        else:
            self._merge_outputs(st, L_out)
        rc = lib.ff_ctx_merge_apply(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
        return self._merge_complete(st, rc)

    def _merge_prepare(self, hidden_states, position_embeddings, attention_mask, residual=None):
        """Validation + the input half of the call block (everything ff_ctx_merge_begin reads), on PyTorch's current stream."""
        _lib.require_gpu(hidden_states, "FrameFusion.forward")
        bsz, L, d = hidden_states.shape
        assert bsz == 1, "Only support batch size 1"                                # main.py:203
        device = hidden_states.device
        dtype = hidden_states.dtype
        code = _dtype_code(hidden_states)
        hidden = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        addend_ptr, addend = 0, None
        if residual is not None:
            addend = residual if residual.is_contiguous() else residual.contiguous()
            addend_ptr = addend.data_ptr()

        ptype = self.patch_type
        if self.__dict__.get("_ptype_checked") != (self._ptype_gen, L, device):
            if ptype.device != device or ptype.dtype != torch.int64 or not ptype.is_contiguous():
                ptype = ptype.to(device=device, dtype=torch.int64).contiguous()
                self.patch_type = ptype
            if ptype.numel() != L:
                raise FrameFusionHipError(f"patch_type has {ptype.numel()} entries for a sequence of {L}")
            self._ptype_checked = (self._ptype_gen, L, device)

        sub = self._compute_pruning_ratio(self.sparsity_list, self.cost)           # main.py:109
        mask_in = self._mask_for(attention_mask, L) if attention_mask is not None else None     # (shape errors before anything runs)
        sc, stream = self._scratch_for(device, L)
        if sc.ctx.in_flight:
            # an earlier call of this instance died between two crossings (an exception in the host code above the library):
            # forget it - the context's reset also restores the workspace protocol
            _lib.check(_lib.load().ff_ctx_reset(sc.ctx_ptr, stream or 0), "ff_ctx_reset")
            sc.order_gen = None
        order_valid = 1 if sc.order_gen == self._ptype_gen else 0
        # first call of a prefill: hand the frame-major layout the prepare() scalars describe to the
        # similarity kernel, which derives and verifies the by-patch order itself (no K0 launch)
        hint_pre = hint_frames = 0
        P = int(self.patch_num)
        if not order_valid:
            hint = self.__dict__.get("_layout_hint")
            if hint is not None and hint[0] + hint[1] * P <= L:
                hint_pre, hint_frames = hint
        _lib.MERGE_CALL_HEAD.pack_into(sc.call, 0, hidden.data_ptr(), addend_ptr, 0, ptype.data_ptr(), code, L, d, L, P,
                                       order_valid, self._threshold_for(dtype), sub, self.ratio_lower_bound, -1,
                                       _lib.FOLD_SEQUENTIAL, hint_pre, hint_frames, stream or 0, 0)
        _PACK_I64.pack_into(sc.call, _lib.MERGE_CALL_LATE_OFFSET, 0)
        sc.order_gen = None                  # until the call has come back
        # the number of non-text tokens, where the host can know it without asking the device (see _merge_exact_tail)
        ftn = None
        if hint_frames:
            ftn = hint_frames * P
        elif order_valid and self.last_call is not None and self.last_call.get("kind") == "merge" and self.last_call["L_out"] == L:
            ftn = self.last_call["ftn"] - (self.last_call["L_in"] - L)
        L_guess = None if ftn is None or ftn <= 0 else L - min(int(sub * ftn), ftn)
        return dict(sc=sc, stream=stream, L=L, d=d, dtype=dtype, device=device, hidden=hidden, addend=addend, ptype=ptype, L_guess=L_guess,
                    hidden_states=hidden_states, position_embeddings=position_embeddings, attention_mask=attention_mask,
                    mask_in=mask_in, residual=residual)

    def _merge_outputs(self, st, L_cap=None):
        """The output half of the call block (read by ff_ctx_merge_finish / _apply only).  L_cap = None: buffers of the INPUT
        length, allocated while the similarity pass runs (the output length is not known yet); else exactly L_cap = l_out rows."""
        sc, L, d, dtype, device, ptype = st["sc"], st["L"], st["d"], st["dtype"], st["device"], st["ptype"]
        call = sc.call
        exact = L_cap is not None
        if not exact:
            L_cap = L
        _PACK_I64.pack_into(call, 56, L_cap)                           # ff_merge_call_t.L_cap
        out = torch.empty((1, L_cap, d), dtype=dtype, device=device)
        ptype_out = torch.empty((1, L_cap), dtype=torch.int64, device=device)
        if st.get("n_aux") and st.get("mask_cap") is None:
            # sized a second time (the guessed length did not come true, and somebody - the GPU, idle, or a kernel with the rows
            # in its registers - is waiting): the sources are described in the call block already; new destinations only
            outs, rebuild = self._position_outputs(st["position_embeddings"], L_cap)
            size, at = _lib.AUX_ENTRY.size, _lib.MERGE_CALL_AUX_OFFSET + 8
            _PACK_PTR.pack_into(call, at, ptype_out.data_ptr())
            for x, o in enumerate(outs):
                _PACK_PTR.pack_into(call, at + size * (x + 1), o.data_ptr())
            _PACK_PTR.pack_into(call, 16, out.data_ptr())
            _PACK_I64.pack_into(call, _lib.MERGE_CALL_AUX_OFFSET - 8, st["n_aux"])
            st.update(out=out, ptype_out=ptype_out, rebuild=rebuild, L_cap=L_cap)
            return
        srcs, outs, rebuild = self._aux_for_positions(st["position_embeddings"], L, L_cap)
        _lib.AUX_ENTRY.pack_into(call, _lib.MERGE_CALL_AUX_OFFSET, ptype.data_ptr(), ptype_out.data_ptr(), 8, 1, 0)    # patch types
        n_aux = 1 + sc.put_aux(call, _lib.MERGE_CALL_AUX_OFFSET + _lib.AUX_ENTRY.size, zip(srcs, outs), L, room=_lib.MAX_AUX - 1)
        _PACK_PTR.pack_into(call, 16, out.data_ptr())
        _PACK_I64.pack_into(call, _lib.MERGE_CALL_AUX_OFFSET - 8, n_aux)
        # (the attention mask is gathered behind the call, into a buffer of the OUTPUT length - never a speculative one: see
        # _merge_complete.  The C ABI's other form - mask + an [L_cap, L_cap] buffer in the call block, gathered by
        # ff_ctx_merge_finish itself - is what a host without a cheap allocator uses; `mask_through_call` routes through it
        # for the tests)
        mask_cap, mask_in = None, st["mask_in"]
        if mask_in is not None and self.mask_through_call:
            mask_cap = torch.empty(1, 1, L_cap, L_cap, dtype=mask_in.dtype, device=mask_in.device)
            _lib.MASK_TRIPLE.pack_into(call, _lib.MERGE_CALL_MASK_OFFSET, mask_in.data_ptr(), mask_cap.data_ptr(), mask_in.element_size())
        else:
            _lib.MASK_TRIPLE.pack_into(call, _lib.MERGE_CALL_MASK_OFFSET, 0, 0, 0)
        st.update(out=out, ptype_out=ptype_out, srcs=srcs, rebuild=rebuild, mask_cap=mask_cap, L_cap=L_cap, n_aux=n_aux)

    def _merge_complete(self, st, rc, confirm=None):
        """The result block -> the state machine of main.py:112-138 and the returned views.  `confirm`: the crossing that finishes
        the call on the device side (ff_ctx_merge_apply), made once the host's bookkeeping is done."""
        lib = _lib.load()
        sc, L, dtype, stream = st["sc"], st["L"], st["dtype"], st["stream"]
        hidden_states, position_embeddings, attention_mask, residual = (st["hidden_states"], st["position_embeddings"],
                                                                        st["attention_mask"], st["residual"])
        nv, ftn, count, branch, k, L_out, err, unhinted, wait_ns, _applied = _lib.MERGE_RESULT.unpack_from(sc.res)
        if unhinted:
            # patch_type is not the frame-major layout the prepare() scalars suggested (e.g. text between
            # the frames): the library repeated the call through K0; stop hinting for this prefill - and for
            # every later prefill whose scalars describe the same (wrong) layout: a packer that puts separators between its
            # frames does so for every prompt, and each wrong hint costs a whole wasted call
            bad = self.__dict__.get("_layout_hint")
            if bad is not None:
                hints = self.__dict__.setdefault("_bad_hints", {})
                if len(hints) >= 16:                           # (bounded: the oldest entry goes)
                    hints.pop(next(iter(hints)))
                hints[(int(self.patch_num),) + tuple(bad)] = 8
            self._layout_hint = None
        if rc:
            _fail(rc, "merge", err)
        assert nv > 0, "no visual tokens"                                          # main.py:240
        if st.get("L_guess") is not None and "guess_idx" in st:
            self.__dict__.setdefault("_guess_held", {})[st["guess_idx"]] = (L_out == st["L_guess"])

        def finish():
            # (with `confirm` the library's bookkeeping of the call - the order swap - happens in that crossing)
            if confirm is not None:
                rc = confirm()
                if rc:
                    _fail(rc, "merge")
            sc.sync_views()
            # which mail the waiting one-launch kernel took (1: sent before the result was known, 2: sized to it; 3: none came
            # in time and the merge kernel followed as a launch of its own); 0: no kernel waited
            return int(sc.stats_host[_lib.STAT_ACK]) & 3 if _applied == 2 else 0

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
            mail_slot = finish()
            self.last_call = dict(kind="merge", L_in=L, L_out=L, nv=nv, ftn=ftn, count=count, branch=branch,
                                  k=k, scratch=sc, dtype=dtype, order=sc.order, wait_ns=wait_ns, unhinted=bool(unhinted),
                                  one_launch=bool(st.get("one_launch")), applied=int(_applied), mail_slot=mail_slot, flow=st.get("flow"))
            sc.order_gen = self._ptype_gen
            if residual is not None:          # nothing folded, but the caller is owed the sum
                return residual + hidden_states, position_embeddings, attention_mask
            return hidden_states, position_embeddings, attention_mask

        hidden_out, ptype_new, pos_new = st["out"].narrow(1, 0, L_out), st["ptype_out"].narrow(1, 0, L_out), st["rebuild"](L_out)
        mail_slot = finish()
        # order maintenance: the merge kernel also wrote the by-patch order of the compacted sequence
        # (now the context's current one), so the next merge call of this prefill skips K0
        self.last_call = dict(kind="merge", L_in=L, L_out=L_out, nv=nv, ftn=ftn, count=count, branch=branch,
                              k=k, scratch=sc, dtype=dtype, order=sc.order_next, wait_ns=wait_ns, unhinted=bool(unhinted),
                              one_launch=bool(st.get("one_launch")), applied=int(_applied), mail_slot=mail_slot, flow=st.get("flow"))
        mask_cap = st["mask_cap"]
        if mask_cap is not None:
            attention_mask = mask_cap[:, :, :L_out, :L_out]
        elif attention_mask is not None:                                            # main.py:137-138
            # gathered now that L_out is known: [1, 1, L_out, L_out] exactly (an input-sized capacity buffer would be
            # 2.7 GB for a bf16 mask at 37 k tokens), on the keep set still sitting in the context
            m = st["mask_in"]
            mask_out = torch.empty(1, 1, L_out, L_out, dtype=m.dtype, device=m.device)
            rc = lib.ff_ctx_gather_mask(sc.ctx_ptr, m.data_ptr(), mask_out.data_ptr(), m.element_size(), L, L_out, stream or 0)
            if rc:
                _fail(rc, "merge (attention mask)")
            attention_mask = mask_out
        self.patch_type = ptype_new                                                 # main.py:132
        sc.order_gen = self._ptype_gen
        return hidden_out, pos_new, attention_mask

    def last_plan(self):
        """Diagnostics of the most recent merge / prune call (views into the reusable scratch: valid
        until the next call): keep mask by sequence position, similarities, by-patch order, member flags."""
        c = self.last_call
        sc, L, nv = c["scratch"], c["L_in"], c["nv"]
        out = dict(keep=sc.keep[:L], member=sc.member[:L])
        if c["kind"] == "merge":
            out.update(sim=sc.sim(c["dtype"], nv), order=c["order"][:nv])
        return out

    def _prune_range(self, q_len):
        """(start, n_img) of main.py:64-66 as host ints."""
        start = self._host_int(self.image_token_start_index)
        n_img = self._host_int(self.image_token_length) - (self._host_int(self.original_length) - q_len)
        return start, n_img

    def _expect_importance(self, S: int, dtype, device):
        """The attention hook is about to compute the importance of a prune call over S tokens: returns
        (plan, token) - `plan` = (ctx pointer, start, n_img, k, stream) for ff_ctx_last_query_importance, which accumulates
        the select tables of the prune in this instance's workspace;
        `token` is what _prune recognises the tensor by.  (None, None) when the prune range does not fit S."""
        with torch.cuda.device(device):
            sc, stream = self._scratch_for(device, S)
            start, n_img = self._prune_range(S)
            if n_img < 0 or start < 0 or start + n_img > S:
                return None, None
            k = round(n_img * (1 - self._compute_pruning_ratio(self.sparsity_list, self.cost)))      # main.py:73-76
            if k < 0 or k > n_img:
                return None, None
            lib = _lib.load()
            _lib.check(lib.ff_ctx_reset(sc.ctx_ptr, stream), "ff_ctx_reset")    # (zeroes the tables if a call died)
            sc.order_gen = None
            token = (id(sc), sc.seq + 1, S, start, n_img, dtype, k)             # (the call below advances the sequence number)
            sc.tables_token = token
            return (sc.ctx_ptr, start, n_img, k, stream), token

    # ---- prune call: main.py:61-101 ----------------------------------------------------------------
    def _prune(self, hidden_states, position_embeddings, attention_mask, self_attn_weights, residual=None):
        _lib.require_gpu(hidden_states, "FrameFusion.forward")
        lib = _lib.load()
        bsz, q_len, d = hidden_states.shape
        assert bsz == 1, "Only support batch size 1"
        device, dtype = hidden_states.device, hidden_states.dtype
        code = _dtype_code(hidden_states)
        hidden = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        start, n_img = self._prune_range(q_len)                                     # main.py:64-66
        sc, stream = self._scratch_for(device, q_len)

        w = self_attn_weights                                                       # main.py:69-70
        lq = w if type(w) is LastQuery else None
        if lq is not None:
            # the hook handed over q_last / K instead of weights (last_query_importance(..., defer=True)): importance, plan
            # and gather go out in ONE host call below
            if lq.S != q_len or lq.device != device:
                raise FrameFusionHipError(f"the deferred importance covers {lq.S} keys on {lq.device}, the sequence has {q_len} tokens on {device}")
            w_code, token, w_ptr, w_dtype = lq.code, None, 0, lq.dtype
        else:
            _lib.require_gpu(w, "FrameFusion.forward(self_attn_weights)")
            if w.ndim != 4 or w.shape[0] != 1 or w.shape[-1] != q_len:
                raise FrameFusionHipError(f"self_attn_weights of shape {tuple(w.shape)} is not [1, H, num, {q_len}]")
            # the head mean and the top-k run in the WEIGHTS' dtype (main.py:69-76: torch.mean / topk of the
            # tensor the attention hook handed over), whatever the activation dtype is
            w_code = _lib.DTYPE_CODE.get(w.dtype)
            if w_code is None:
                raise FrameFusionHipError(f"unsupported attention-weight dtype {w.dtype} (fp32 / bf16 / fp16 only)")
            token = getattr(w, "_ff_tables", None)
            if not w.is_contiguous():
                w = w.contiguous()
            if w.data_ptr() & 15:
                w = w.clone()
            w_ptr, w_dtype = w.data_ptr(), w.dtype
        pruning_ratio = self._compute_pruning_ratio(self.sparsity_list, self.cost)  # main.py:73
        k = round(n_img * (1 - pruning_ratio))                                      # main.py:76
        if k < 0 or k > n_img:
            raise RuntimeError("selected index k out of range")                     # torch.topk's error
        L_out = q_len - n_img + k
        # importance from last_query_importance(..., framefusion=self): its kernel has already filled the
        # select tables of exactly this call in the workspace
        H, num = w.shape[1], w.shape[2]
        tables_ready = int(token is not None and token == sc.tables_token and
                           token == (id(sc), sc.seq, q_len, start, n_img, w_dtype, k) and H * num == 1)
        sc.tables_token = None
        one_crossing = False
        if lq is not None:
            need = int(lib.ff_last_query_workspace_bytes(lq.code, lq.H, lq.num, q_len, lq.dh))
            if sc.lq_ws is None or sc.lq_ws.numel() < need:
                if sc.lq_ws is not None and sc.last_stream is not None:
                    sc.lq_ws.record_stream(sc.last_stream)
                sc.lq_ws = torch.empty(need, dtype=torch.uint8, device=device)
            one_crossing = bool(self.prune_in_one_crossing)
            if not one_crossing:
                # the importance kernels go out NOW, before any output tensor exists: the allocations and descriptors below
                # (~30 us of host work) run under them.  (ff_ctx_prune_from_qk - importance, plan and gather enqueued by one
                # crossing at the END of this function - is the form for a host whose preparation is cheap; here it left the
                # GPU idle while Python prepared: 166-169 vs 156 us per Qwen2-VL cascade, profiles/EXPERIMENTS.md 5.4.)
                if sc.ctx.dirty:
                    _lib.check(lib.ff_ctx_reset(sc.ctx_ptr, stream or 0), "ff_ctx_reset")
                rc = lib.ff_ctx_last_query_importance(sc.ctx_ptr, lq.q_last.data_ptr(), lq.k.data_ptr(), lq.code, lq.H, lq.H_kv, lq.num,
                                                      q_len, lq.dh, lq.sh, lq.ss, lq.factor, 1 if lq.is_causal else 0,
                                                      lq.bias.data_ptr() if lq.bias is not None else None, sc.sim32.data_ptr(),
                                                      start, n_img, k, sc.lq_ws.data_ptr(), sc.lq_ws.numel(), stream or 0)
                if rc:
                    _fail(rc, "prune (importance)")
                w_ptr, H, num, tables_ready = sc.sim32.data_ptr(), 1, 1, 1
        # nothing is read back (L_out is known): head mean + select tables, plan, gather - one host call
        out = torch.empty((1, L_out, d), dtype=dtype, device=device)
        srcs, outs, rebuild = self._aux_for_positions(position_embeddings, q_len, L_out)
        addend_ptr = 0
        if residual is not None:
            addend = residual if residual.is_contiguous() else residual.contiguous()
            addend_ptr = addend.data_ptr()
        call = sc.pcall
        n_aux = sc.put_aux(call, _lib.PRUNE_CALL_AUX_OFFSET, zip(srcs, outs), q_len)
        _lib.PRUNE_CALL_HEAD.pack_into(call, 0, hidden.data_ptr(), addend_ptr, out.data_ptr(), w_ptr, code, q_len, d,
                                       L_out, w_code, H, num, tables_ready, start, n_img, k, stream or 0, n_aux)
        mask_out = None
        if attention_mask is not None:
            m = self._mask_for(attention_mask, q_len)
            mask_out = torch.empty(1, 1, L_out, L_out, dtype=m.dtype, device=m.device)
            _lib.MASK_TRIPLE.pack_into(call, _lib.PRUNE_CALL_MASK_OFFSET, m.data_ptr(), mask_out.data_ptr(), m.element_size())
        else:
            _lib.MASK_TRIPLE.pack_into(call, _lib.PRUNE_CALL_MASK_OFFSET, 0, 0, 0)
        sc.order_gen = None
        if one_crossing:
            _lib.LQ_ARGS.pack_into(sc.lqargs, 0, lq.q_last.data_ptr(), lq.k.data_ptr(), lq.code, lq.H, lq.H_kv, lq.num, lq.dh, lq.sh, lq.ss,
                                   lq.factor, 1 if lq.is_causal else 0, lq.bias.data_ptr() if lq.bias is not None else 0,
                                   sc.lq_ws.data_ptr(), sc.lq_ws.numel())
            rc = lib.ff_ctx_prune_from_qk(sc.ctx_ptr, sc.pcall_ptr, sc.lqargs_ptr)
        else:
            rc = lib.ff_ctx_prune(sc.ctx_ptr, sc.pcall_ptr)
        if rc:
            _fail(rc, "prune")
        self.finish_pruning = True                                                  # main.py:101
        self.last_call = dict(kind="prune", L_in=q_len, L_out=L_out, k=k, nv=q_len, scratch=sc, dtype=w_dtype)
        return out, rebuild(L_out), mask_out

    def prune_from_qk(self, hidden_states, position_embeddings, attention_mask, query, key, num=1, is_causal=True, scale=None,
                      residual=None):
        """The prune call (main.py:61-101) fed by the attention hook's INPUTS instead of its output: ``query`` [1, H, L, dh] /
        ``key`` [1, H_kv, S, dh] of the layer (un-repeated GQA keys, utils.py:27-57 semantics).  Importance, select, gather =
        one crossing of the C ABI (``ff_ctx_prune_from_qk``), nothing read back.  Equal to
        ``forward(hidden, pos, mask, last_query_importance(query, key, num, is_causal, scale, framefusion=self))``."""
        bsz, q_len, _ = hidden_states.size()
        if not (q_len > 1 and self.finish_merging == True and self.finish_pruning == False):
            raise FrameFusionHipError("prune_from_qk: no prune is due (finish_merging must be set, finish_pruning not)")
        return self.forward(hidden_states, position_embeddings, attention_mask,
                            LastQuery(query, key, num, is_causal, scale), residual=residual)

    # ---- the reference's public position handlers: main.py:142-178 ----------------------------------------------
    def _handler_tensors(self, position_embeddings):
        if type(position_embeddings) == list:
            assert len(position_embeddings) == 2                                     # main.py:144,163
            return [t if t.is_contiguous() else t.contiguous() for t in position_embeddings]
        if type(position_embeddings) == torch.Tensor:
            if position_embeddings.ndim != 2:
                raise NotImplementedError("Only support 2D position embeddings")     # main.py:155,175
            return [position_embeddings.contiguous()]
        raise NotImplementedError("Only support list or tensor for position embeddings")   # main.py:157,177

    @staticmethod
    def _handler_aux(srcs, outs):
        aux = (_lib.FFAux * _lib.MAX_AUX)()
        for n, (s, o) in enumerate(zip(srcs, outs)):
            if s.ndim == 2:
                L, row, outer = s.shape[1], s.element_size(), s.shape[0]
            else:
                L, last = s.shape[-2], s.shape[-1]
                row, outer = last * s.element_size(), s.numel() // (L * last)
            aux[n] = _lib.FFAux(s.data_ptr(), o.data_ptr(), row, outer, 0)
        return aux

    def position_embedding_handler_at_pruning(self, position_embeddings, keep_indexs):
        """main.py:142-158: ``pe[..., keep_indexs, :]`` for the [cos, sin] list (3-D or 4-D, mutated in place like the
        reference) or ``pe[:, keep_indexs]`` for a 2-D position tensor - one gather launch for the whole container."""
        srcs = self._handler_tensors(position_embeddings)
        _lib.require_gpu(srcs[0], "position_embedding_handler_at_pruning")
        with torch.cuda.device(srcs[0].device):
            idx = keep_indexs.to(device=srcs[0].device, dtype=torch.int64).contiguous()
            n = idx.numel()
            axis = -1 if srcs[0].ndim == 2 else -2
            L = srcs[0].shape[axis]
            if n:
                # torch's advanced indexing raises on an index outside [-L, L): so does this (one read-back; the handler is
                # a compatibility entry point, the hot path gathers its position tensors inside the merge kernel)
                lo, hi = int(idx.min()), int(idx.max())
                if lo < -L or hi >= L:
                    bad = lo if lo < -L else hi
                    raise IndexError(f"index {bad} is out of bounds for dimension {srcs[0].ndim + axis} with size {L}")
            outs = []
            for t in srcs:
                shape = list(t.shape)
                shape[axis] = n
                outs.append(torch.empty(shape, dtype=t.dtype, device=t.device))
            _lib.check(_lib.load().ff_gather_tokens_by_index(idx.data_ptr(), n, L, self._handler_aux(srcs, outs), len(srcs),
                                                             _lib.stream_ptr()), "ff_gather_tokens_by_index")
        if type(position_embeddings) == list:
            position_embeddings[0], position_embeddings[1] = outs
            return position_embeddings
        return outs[0]

    def position_embedding_handler_at_merging(self, position_embeddings, token_mask):
        """main.py:161-178: ``pe[..., token_mask[0], :]`` / ``pe[:, token_mask[0]]``.  The number of kept tokens is read
        back (the reference's boolean indexing synchronises as well)."""
        srcs = self._handler_tensors(position_embeddings)
        _lib.require_gpu(srcs[0], "position_embedding_handler_at_merging")
        dev = srcs[0].device
        with torch.cuda.device(dev):
            keep = token_mask[0].to(device=dev, dtype=torch.bool).contiguous()
            axis = -1 if srcs[0].ndim == 2 else -2
            L = srcs[0].shape[axis]
            if keep.numel() != L:
                raise IndexError(f"The shape of the mask [{keep.numel()}] does not match the {L} tokens of the position embeddings")
            if keep.data_ptr() & 15:
                keep = keep.clone()
            dst = torch.empty(L, dtype=torch.int32, device=dev)
            stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=dev)
            outs = []
            for t in srcs:
                outs.append(torch.empty_like(t))                      # capacity L rows: narrowed once the count is known
            _lib.check(_lib.load().ff_gather_tokens_by_mask(keep.data_ptr(), L, L, dst.data_ptr(), stats.data_ptr(),
                                                            self._handler_aux(srcs, outs), len(srcs), _lib.stream_ptr()),
                       "ff_gather_tokens_by_mask")
            n = int(stats[_lib.STAT_LOUT])
        outs = [o.narrow(o.ndim + axis, 0, n) for o in outs]
        if type(position_embeddings) == list:
            position_embeddings[0], position_embeddings[1] = outs
            return position_embeddings
        return outs[0]

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
        if (d * hidden.element_size()) & 15:              # (zero columns up to whole 16-byte words: the similarities do not change)
            hidden = torch.nn.functional.pad(hidden, (0, (-d) % (16 // hidden.element_size())))
            d = hidden.shape[2]
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
        d_pad = d + (-d) % (16 // hidden.element_size())     # (rows as whole 16-byte words: zero columns fold to zero)
        if d_pad != d:
            hidden = torch.nn.functional.pad(hidden, (0, d_pad - d))
        compact = torch.empty(L, d_pad, dtype=dtype, device=device)
        _lib.check(lib.ff_merge_compact(hidden.data_ptr(), compact.data_ptr(), code, L, d_pad, L, order.data_ptr(),
                                        member.data_ptr(), 1, dst.data_ptr(), None, None, 0, stream),
                   "ff_merge_compact")
        keep_b = keep.bool()
        kept = torch.nonzero(keep_b).reshape(-1)
        hidden_states[0, kept] = compact[: kept.numel(), :d]      # members keep their old rows, like the reference
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


def call_b_with_residual(framefusion, residual, hidden_states, position_embeddings, attention_mask, self_attn_weights=None):
    """Call B of a decoder layer (models/qwen2/modeling_qwen2.py:64-67: ``hidden = residual + attn_out`` and
    then ``self.framefusion(hidden, ...)``) for whatever object sits on ``.framefusion``: through
    ``nn.Module.__call__`` (forward hooks run) with the add fused into the reduction when the object takes
    ``residual=``, else the reference contract - the eager add and the 4-argument call."""
    if getattr(framefusion, "supports_residual", False):
        return framefusion(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual=residual)
    return framefusion(residual + hidden_states, position_embeddings, attention_mask, self_attn_weights)


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
