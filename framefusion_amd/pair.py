"""Two samples in flight on ONE GPU from one host thread.

The reference uses one GPU for two samples by running two model replicas on two Python threads, never sharing a
``FrameFusion`` instance (script/demo/llava_video_compare.py:217-223, 310-313).  A merge call of this build leaves the chip
idle for the ~9 us of its plan kernel and for the ramp / drain of its two streaming passes; with a second, independent sample
on a second HIP stream, half a call out of phase, those bubbles sit under the other sample's streaming pass.

``FrameFusionPair`` does that without threads, on top of ``FrameFusion.submit`` / ``collect`` (``ff_ctx_merge_submit`` /
``ff_ctx_merge_collect``: a merge call enqueued in full, the wait for its result block split off): the host submits sample B
on stream 1 BEFORE it collects sample A on stream 0, then submits the next A before it collects B, and so on - always one
call enqueued ahead of the one being waited for.  Submitting both samples together and collecting both (lock step) was built
first and measured SLOWER than two calls one after the other (332 vs 270 us per two calls at 64 x 576 x 4096: the two
similarity passes compete and both plan bubbles coincide); what pays is the phase shift, which the alternation below
creates and keeps.  Results are those of two independent instances, bit for bit (tests/test_gpu_pair.py); each instance
keeps its own scratch, state machine and by-patch order.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .main import FrameFusion


_STREAM_PAIRS = {}


STREAM_SOURCE = {}        # device key -> "hip runtime (<path>)" | "torch pool": which streams the pair got (bench.py reports it)


def _loaded_hip_runtime() -> str:
    """Path of the HIP runtime THIS process has mapped (the one torch loaded and libframefusion_hip.so resolved its SONAME
    against): a stream handed to torch.cuda.ExternalStream must come from that very runtime, not from another copy a bare
    SONAME might find.  OSError if none is mapped."""
    with open("/proc/self/maps") as maps:
        for line in maps:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in os.path.basename(path):
                return path
    raise OSError("no libamdhip64 mapped into this process")


def _new_hip_stream(device):
    """A stream created through the HIP runtime itself (hipStreamCreateWithFlags, non-blocking), wrapped for PyTorch.  Streams
    from torch's pool were all created when the pool was: which hardware queue each sits on is fixed, and two of them can share
    one - their kernels then run in submission order (1 pool pair in 5 did, profiles/r05_pair_probe.txt: 270 us per two calls
    where independent streams give 231-242).  A stream created NOW gets the least used queue: 4 pairs of 4 ran at 233-239 us."""
    import ctypes
    hip = ctypes.CDLL(_loaded_hip_runtime())
    s = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipStreamCreateWithFlags(ctypes.byref(s), ctypes.c_uint(1))      # hipStreamNonBlocking
    if rc != 0 or not s.value:
        raise OSError(f"hipStreamCreateWithFlags failed ({rc})")
    return torch.cuda.ExternalStream(s.value, device=device)


def concurrent_streams(device):
    """The two sample streams of `device`, created once per process (never destroyed) and shared by every pair."""
    key = (device.type, device.index)
    got = _STREAM_PAIRS.get(key)
    if got is None:
        try:
            got = (_new_hip_stream(device), _new_hip_stream(device))
            STREAM_SOURCE[key] = f"hip runtime ({_loaded_hip_runtime()})"
        except (OSError, AttributeError):                       # no direct access to the runtime: torch's pool will do
            with torch.cuda.device(device):
                got = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
            STREAM_SOURCE[key] = "torch pool"
        _STREAM_PAIRS[key] = got
    return got


class FrameFusionPair:
    """``pair = FrameFusionPair(ff_a, ff_b)``.

    * ``pair.submit(x, hidden, position_embeddings, attention_mask[, self_attn_weights])`` enqueues the call of sample
      ``x`` (0 or 1) on that sample's stream and returns at once; ``pair.collect(x)`` waits for it and returns what
      ``FrameFusion.forward`` returns.  Keep the other sample submitted while you collect one.
    * ``pair.run(calls)``: ``calls`` yields ``(x, hidden, position_embeddings, attention_mask[, self_attn_weights])`` with
      x alternating 0, 1, 0, 1, ...; yields the results in the same order, one call late (the call behind a result is
      already enqueued when the result is waited for).

    The sample streams (``pair.streams``) first wait for PyTorch's current stream at every submit (the inputs were produced
    there); ``collect`` makes the current stream wait for the sample's stream (the outputs are consumed there) - device-side
    dependencies only, never a host synchronisation.  ``sync_with_current=False`` leaves both out for a caller that keeps each
    sample's tensors on its pair stream itself."""

    def __init__(self, ff_a: FrameFusion, ff_b: FrameFusion, device=None, sync_with_current: bool = True):
        if ff_a is ff_b:
            raise _lib.FrameFusionHipError("FrameFusionPair needs two different FrameFusion instances (one per sample)")
        self.ffs = (ff_a, ff_b)
        self.sync_with_current = sync_with_current
        self._streams = {}
        self._pending = [None, None]          # (ticket, device) of each sample's call in flight
        if device is not None:
            self.streams_for(torch.device(device))

    @property
    def a(self):
        return self.ffs[0]

    @property
    def b(self):
        return self.ffs[1]

    def streams_for(self, device):
        key = (device.type, device.index)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = concurrent_streams(device)
        return st

    @property
    def streams(self):
        """(stream of sample 0, stream of sample 1) on the current device."""
        return self.streams_for(torch.device("cuda", torch.cuda.current_device()))

    def submit(self, x, hidden_states, position_embeddings, attention_mask, self_attn_weights=None, residual=None):
        _lib.require_gpu(hidden_states, "FrameFusionPair.submit")
        if self._pending[x] is not None:
            raise _lib.FrameFusionHipError(f"sample {x} already has a call in flight: collect({x}) first")
        dev = hidden_states.device
        with torch.cuda.device(dev):
            s = self.streams_for(dev)[x]
            if self.sync_with_current:
                s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                ticket = self.ffs[x].submit(hidden_states, position_embeddings, attention_mask, self_attn_weights, residual)
        self._pending[x] = (ticket, dev)

    def collect(self, x):
        if self._pending[x] is None:
            raise _lib.FrameFusionHipError(f"sample {x} has no call in flight")
        ticket, dev = self._pending[x]
        self._pending[x] = None
        with torch.cuda.device(dev):
            s = self.streams_for(dev)[x]
            with torch.cuda.stream(s):
                out = self.ffs[x].collect(ticket)
            if self.sync_with_current:
                cur = torch.cuda.current_stream(dev)
                cur.wait_stream(s)
                for t in (out[0], out[2], *(out[1] if type(out[1]) == list else [out[1]])):      # allocated on the sample's stream
                    if isinstance(t, torch.Tensor):
                        t.record_stream(cur)
        return out

    def run(self, calls):
        """Generator: results of `calls` in order, each collected after the NEXT call has been submitted."""
        waiting = None
        for call in calls:
            x = call[0]
            if waiting is not None and waiting == x:
                yield self.collect(waiting)           # (two calls of the same sample in a row: nothing to overlap with)
                waiting = None
            self.submit(x, *call[1:])
            if waiting is not None:
                yield self.collect(waiting)
            waiting = x
        if waiting is not None:
            yield self.collect(waiting)
