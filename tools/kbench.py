#!/usr/bin/env python
"""Per-kernel timing of the merge call on one GPU (development tool; bench.py is the contract).

    python tools/kbench.py [--frames 64 --patches 576 --dim 4096 --p-change 0.2 --layout frame|patch]

Prints one line per stage: average microseconds (hipEvents on the launch stream), algorithmic
bytes and GB/s.  --layout patch stores the tokens patch-major, so the by-patch order is the
memory order: the sequential-access upper bound for K1/K4.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa                      # noqa: E402
from framefusion_amd import _lib                   # noqa: E402
from framefusion_amd.synth import video_tokens, rotary_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--patches", type=int, default=576)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--p-change", type=float, default=0.2)
    ap.add_argument("--layout", default="frame")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    F, P, d = args.frames, args.patches, args.dim
    hidden, ptype = video_tokens(F, P, d, p_change=args.p_change, seed=1234, dtype=dtype, device="cuda:0")
    if args.layout == "patch":
        hidden = hidden.view(F, P, d).transpose(0, 1).reshape(1, F * P, d).contiguous()
        ptype = ptype.view(F, P).t().reshape(1, F * P).contiguous()
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, 128, dtype, device="cuda:0")
    ff = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)
    ff.prepare(ptype, P, 0, L, L, L)
    out, _, _ = ff(hidden, [cos, sin], None)
    info = ff.last_call
    L_out, nv = info["L_out"], info["nv"]
    print(f"L={L} -> {L_out}  branch={'topk' if info['branch'] else 'thr'} count={info['count']} k={info['k']}")

    lib = _lib.load()
    sc = ff._scratch[(dev.type, dev.index)]
    order_buf = info["order"]
    stream = _lib.stream_ptr()
    code = _lib.DTYPE_CODE[dtype]
    elt = hidden.element_size()
    sim = sc.sim(dtype, L)
    out_buf = torch.empty(1, L, d, dtype=dtype, device=dev)
    ptype_out = torch.empty(1, L, dtype=torch.int64, device=dev)
    cos_o, sin_o = torch.empty_like(cos), torch.empty_like(sin)
    aux = (_lib.FFAux * _lib.MAX_AUX)()
    aux[0] = _lib.FFAux(ptype.data_ptr(), ptype_out.data_ptr(), 8, 1)
    aux[1] = _lib.FFAux(cos.data_ptr(), cos_o.data_ptr(), 128 * elt, 1)
    aux[2] = _lib.FFAux(sin.data_ptr(), sin_o.data_ptr(), 128 * elt, 1)
    thr = float(torch.tensor(0.6, dtype=dtype))
    sub = float(ff._compute_pruning_ratio([], 0.3))
    stages = {
        "order": (lambda: lib.ff_build_order(ptype.data_ptr(), L, P, order_buf.data_ptr(), None, sc.stats.data_ptr(), sc.ws.data_ptr(), sc.ws_bytes, stream),
                  8 * L + 4 * L),
        "similarity": (lambda: lib.ff_pair_similarity(hidden.data_ptr(), code, L, d, ptype.data_ptr(), order_buf.data_ptr(),
                                                      sc.stats.data_ptr(), sim.data_ptr(), stream), nv * d * elt),
        "plan": (lambda: lib.ff_plan_merge(sim.data_ptr(), code, order_buf.data_ptr(), L, thr, sub, 0.1, sc.member.data_ptr(),
                                           sc.dst.data_ptr(), sc.keep.data_ptr(), sc.stats.data_ptr(), sc.ws.data_ptr(),
                                           sc.ws_bytes, stream), nv * elt + 4 * L * 3),
        "merge_compact": (lambda: lib.ff_merge_compact(hidden.data_ptr(), out_buf.data_ptr(), code, L, d, L, order_buf.data_ptr(),
                                                       sc.member.data_ptr(), 1, sc.dst.data_ptr(), sc.keep.data_ptr(), aux, 3, stream),
                          (L + L_out) * d * elt + 2 * (L + L_out) * 128 * elt),
    }
    total = 0.0
    for name, (fn, nbytes) in stages.items():
        for _ in range(3):
            _lib.check(fn(), name)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            _lib.check(fn(), name)
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        us = sum(ts) / len(ts)
        total += us
        print(f"{name:14s} avg {us:8.1f} us  min {ts[0]:8.1f}  med {ts[len(ts)//2]:8.1f}   {nbytes/1e6:8.1f} MB  {nbytes/us/1e3:8.1f} GB/s")
    torch.cuda.synchronize()
    st = sc.stats.cpu().tolist()
    print("K0 phase cycles:", st[_lib.STAT_T_ORDER:_lib.STAT_T_ORDER + 2], " plan pass cycles [A, select, D, E]:",
          st[_lib.STAT_T_PLAN:_lib.STAT_T_PLAN + 4])
    # whole call
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        ff.prepare(ptype, P, 0, L, L, L)
        ff(hidden, [cos, sin], None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e6
    print(f"sum of stages {total:.1f} us; whole forward call (host wall) {dt:.1f} us")


if __name__ == "__main__":
    main()
