"""Several independent video samples on ONE GPU.

A FrameFusion instance reduces one sample (`bsz == 1`, reference main.py:203).  A merge call is two
HBM-bound streaming passes separated by a handful of latency-bound index kernels; run alone, those
small kernels leave the memory system idle for ~20 % of the call.  `forward_many` gives every sample
its own HIP stream and enqueues ALL samples before waiting for any result block, so one sample's
index kernels run underneath another sample's streaming pass.  Results are identical to calling the
instances one after another (each instance has its own scratch and state).
"""
from __future__ import annotations

from typing import List, Sequence

import torch

_streams = {}


def _stream_pool(device, n):
    pool = _streams.setdefault((device.type, device.index), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def forward_many(instances: Sequence, hidden_states: Sequence[torch.Tensor], position_embeddings: Sequence,
                 attention_masks: Sequence = None, self_attn_weights: Sequence = None) -> List[tuple]:
    """`instances[i].forward(hidden_states[i], position_embeddings[i], attention_masks[i], self_attn_weights[i])`
    for every i, overlapped on per-sample streams.  Returns the list of result triples."""
    n = len(instances)
    masks = attention_masks if attention_masks is not None else [None] * n
    weights = self_attn_weights if self_attn_weights is not None else [None] * n
    if n == 0:
        return []
    device = hidden_states[0].device
    main = torch.cuda.current_stream(device)
    streams = _stream_pool(device, n)
    results: List = [None] * n
    pending: List = [None] * n
    for i, ff in enumerate(instances):
        h = hidden_states[i]
        q_len = h.shape[1]
        s = streams[i]
        s.wait_stream(main)                      # inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            if q_len > 1 and not ff.finish_merging:
                pending[i] = ff._merge_launch(h, position_embeddings[i], masks[i])
            else:                                # prune / no-op: no readback to overlap
                results[i] = ff.forward(h, position_embeddings[i], masks[i], weights[i])
    for i, ff in enumerate(instances):
        if pending[i] is not None:
            with torch.cuda.stream(streams[i]):  # a rejected layout hint repeats the call: same stream
                results[i] = ff._merge_complete(pending[i])
    for s in streams:
        main.wait_stream(s)                      # consumers on the caller's stream see finished outputs
    return results
