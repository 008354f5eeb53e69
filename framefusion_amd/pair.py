"""Two samples in flight on ONE GPU from one host thread.

The reference uses one GPU for two samples by running two model replicas on two Python threads, never sharing a
``FrameFusion`` instance (script/demo/llava_video_compare.py:217-223, 310-313).  A merge call of this build leaves the chip
idle for the ~9 us of its plan kernel and for the ramp / drain of its two streaming passes; with a second, independent sample
on a second HIP stream those bubbles sit under the other sample's streaming pass.  ``FrameFusionPair`` does that without
threads: one crossing of the C ABI (``ff_ctx_merge_pair``) enqueues K1(a), K1(b), plan + K4 (a), plan + K4 (b) - each sample
on its own stream - and only then waits for the two result blocks.  Results are those of two independent instances, bit for
bit (tests/test_gpu_pair.py); each instance keeps its own scratch, state machine and by-patch order.
"""
from __future__ import annotations

import torch

from . import _lib
from .main import FrameFusion, _fail


class FrameFusionPair:
    """``pair = FrameFusionPair(ff_a, ff_b)``; ``out_a, out_b = pair(args_a, args_b)`` with ``args_x`` the positional arguments
    of ``FrameFusion.forward`` for that sample ``(hidden_states, position_embeddings, attention_mask[, self_attn_weights])``.

    When both calls are merge calls on the same device they go out as one ``ff_ctx_merge_pair``; anything else (a prune, a call
    after the reductions have finished, a decode step, different devices) runs as two ordinary calls, each still on its own
    stream.  ``sync_with_current`` (default): the pair's streams first wait for PyTorch's current stream (the inputs were
    produced there) and the current stream then waits for both samples (the outputs are consumed there) - no host
    synchronisation either way.  A caller that keeps each sample on its pair stream itself (``pair.streams``) passes False and
    gets full pipelining across consecutive pairs as well."""

    def __init__(self, ff_a: FrameFusion, ff_b: FrameFusion, device=None, sync_with_current: bool = True):
        if ff_a is ff_b:
            raise _lib.FrameFusionHipError("FrameFusionPair needs two different FrameFusion instances (one per sample)")
        self.a, self.b = ff_a, ff_b
        self.sync_with_current = sync_with_current
        self._streams = {}
        if device is not None:
            self.streams_for(torch.device(device))

    def streams_for(self, device):
        key = (device.type, device.index)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return st

    @property
    def streams(self):
        """(stream of sample a, stream of sample b) on the current device."""
        return self.streams_for(torch.device("cuda", torch.cuda.current_device()))

    @staticmethod
    def _is_merge(ff, args):
        h = args[0]
        return h.is_cuda and h.shape[1] > 1 and not ff.finish_merging

    def __call__(self, args_a, args_b):
        return self.forward(args_a, args_b)

    def forward(self, args_a, args_b):
        ha, hb = args_a[0], args_b[0]
        _lib.require_gpu(ha, "FrameFusionPair.forward")
        _lib.require_gpu(hb, "FrameFusionPair.forward")
        dev = ha.device
        if hb.device != dev:
            return self.a(*args_a), self.b(*args_b)
        with torch.cuda.device(dev):
            sa, sb = self.streams_for(dev)
            cur = torch.cuda.current_stream(dev)
            if self.sync_with_current:
                sa.wait_stream(cur)
                sb.wait_stream(cur)
            both_merge = self._is_merge(self.a, args_a) and self._is_merge(self.b, args_b)
            if not both_merge:
                with torch.cuda.stream(sa):
                    out_a = self.a(*args_a)
                with torch.cuda.stream(sb):
                    out_b = self.b(*args_b)
            else:
                out_a, out_b = self._merge_pair(args_a, args_b, sa, sb)
            if self.sync_with_current:
                cur.wait_stream(sa)
                cur.wait_stream(sb)
                for out in (out_a, out_b):          # allocated on a pair stream, consumed on the caller's
                    for t in (out[0], out[2], *(out[1] if type(out[1]) == list else [out[1]])):
                        if isinstance(t, torch.Tensor):
                            t.record_stream(cur)
            return out_a, out_b

    def _merge_pair(self, args_a, args_b, sa, sb):
        lib = _lib.load()
        a, b = self.a, self.b
        with torch.cuda.stream(sa):
            st_a = a._merge_prepare(*args_a[:3], residual=None)
            a._merge_outputs(st_a)
        try:
            with torch.cuda.stream(sb):
                st_b = b._merge_prepare(*args_b[:3], residual=None)
                b._merge_outputs(st_b)
        except Exception:
            # sample b never started: a goes alone
            with torch.cuda.stream(sa):
                sc = st_a["sc"]
                rc = lib.ff_ctx_merge(sc.ctx_ptr, sc.call_ptr, sc.res_ptr)
                a._merge_complete(st_a, rc)
            raise
        ca, cb = st_a["sc"], st_b["sc"]
        if ca is cb or st_a["stream"] == st_b["stream"]:
            raise _lib.FrameFusionHipError("the two samples of a pair need their own scratch and their own stream")
        rc = lib.ff_ctx_merge_pair(ca.ctx_ptr, ca.call_ptr, ca.res_ptr, cb.ctx_ptr, cb.call_ptr, cb.res_ptr)
        # each sample's own verdict: the error word of ITS result block, else the pair's return code where the block says nothing
        err_a = int(ca.res.error)
        first = None
        outs = []
        for ff, st, stream in ((a, st_a, sa), (b, st_b, sb)):
            sc = st["sc"]
            with torch.cuda.stream(stream):
                try:
                    bad = rc if (rc and (int(sc.res.error) or int(sc.ctx.dirty))) else 0
                    outs.append(ff._merge_complete(st, bad))
                except Exception as e:                  # noqa: BLE001 (finish the other sample's bookkeeping first)
                    first = first or e
                    outs.append(None)
        if first is not None:
            raise first
        if rc:
            _fail(rc, "merge pair", err_a)
        return outs[0], outs[1]
