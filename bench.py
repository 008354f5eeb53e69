#!/usr/bin/env python
"""Headline benchmark: vision tokens reduced per second by one FrameFusion.forward merge call on
synthetic [1, 64 frames x 576 tokens, 4096] bf16 activations (BASELINE.json configs[1], "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = prepare() + one FrameFusion.forward call (by-patch order - derived inside the similarity
kernel from prepare()'s layout scalars and verified there -, K1 similarity, K2/K3 plan, K4
merge+compaction, one 256-byte result block) on one video sample resident in HBM.  With N ranks each
rank reduces its own independent sample (seed + rank): weak scaling, no data-path collective.
Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (live hipEvent timing on the
launch stream, algorithmic bytes from DESIGN.md); `cpu_baseline` times the CPU oracle
(oracle/ff_oracle.py, a torch-CPU port of the reference path) on the same input.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAMES, PATCHES, DIM, HEAD_DIM = 64, 576, 4096, 128
COST, THRESHOLD, RATIO_LB = 0.3, 0.6, 0.1          # reference operating point (README.md:123)
P_CHANGE, SIGMA = 0.2, 0.3                         # top-k regime 36864 -> 11060
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
BASELINE_METRIC = "vision tokens reduced/sec (64 frames\u00d7576 tok, d=4096 bf16), 1\u21928 MI355X"   # BASELINE.json


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--patches", type=int, default=PATCHES)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--p-change", type=float, default=P_CHANGE)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend for the scalars exchanged off the timed path (nccl = RCCL over xGMI)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs (rank r -> GPU r %% device_count; gloo only: RCCL refuses two ranks "
                         "on one device) - exercises the N > 1 path on a 1-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-calls", type=int, default=5)
    return ap.parse_args()


def algorithmic_bytes(L_in, L_out, nv, d, elt, head_dim, pe_outer=1):
    """DESIGN.md §Roofline / SURVEY.md §8d: compulsory HBM traffic of one merge call."""
    hidden_in = L_in * d * elt
    hidden_out = L_out * d * elt
    pos = 2 * (L_in + L_out) * head_dim * elt * pe_outer
    ints = 8 * (L_in + L_out)
    return dict(step=hidden_in + hidden_out + pos + ints,
                similarity=nv * d * elt + nv * elt,
                merge_compact=hidden_in + hidden_out + pos + ints)


def main():
    args = parse()
    from framefusion_amd import dp
    # `python bench.py --gpus N` starts its own N ranks (one per GPU) unless a launcher already did
    dp.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    world, rank, local = dp.env_world()
    if world != max(1, args.gpus):
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        print("bench.py needs an MI355X: framefusion_amd has no CPU path", file=sys.stderr)
        sys.exit(2)
    if world > n_dev and not (args.oversubscribe and args.backend == "gloo"):
        print(f"bench.py: {world} ranks but {n_dev} GPU(s) visible (one rank per GPU; "
              f"--oversubscribe --backend gloo shares GPUs for a functional check)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(dev)
    dist = dp.init(args.backend, dev)
    # rank 0's workload description is THE workload: broadcast over RCCL before anything is generated
    cfg = dp.broadcast_config(dist, dict(seed=args.seed, frames=args.frames, patches=args.patches, dim=args.dim,
                                         p_change=args.p_change, steps=args.steps, warmup=args.warmup), dev)
    args.seed, args.frames, args.patches, args.dim = cfg["seed"], cfg["frames"], cfg["patches"], cfg["dim"]
    args.p_change, args.steps, args.warmup = cfg["p_change"], cfg["steps"], cfg["warmup"]

    import framefusion_amd as ffa
    from framefusion_amd import _lib
    from framefusion_amd.synth import video_tokens, rotary_tables

    F, P, d = args.frames, args.patches, args.dim
    hidden, ptype = video_tokens(F, P, d, p_change=args.p_change, sigma=SIGMA, seed=dp.sample_seed(args.seed, rank),
                                 dtype=torch.bfloat16, device=str(dev))
    L = hidden.shape[1]
    cos, sin = rotary_tables(L, HEAD_DIM, torch.bfloat16, device=str(dev))
    ff = ffa.FrameFusion(COST, THRESHOLD, RATIO_LB)
    # the timed loop alternates between two copies of the sample so that no step finds its input
    # in the 256 MiB Infinity Cache just because the previous step read the very same buffer
    hidden_alt = hidden.clone()
    flip = [0]

    def step():
        flip[0] ^= 1
        ff.prepare(ptype, P, 0, L, L, L)
        out, pos, _ = ff(hidden_alt if flip[0] else hidden, [cos, sin], None)
        return out

    # W untimed steps, then EXACTLY K steps between barrier + synchronize; max over ranks
    t_max, elapsed, out = dp.timed_steps(dist, step, args.steps, args.warmup, dev)
    reduced = L - out.shape[1]
    L_out = out.shape[1]
    info = ff.last_call
    # whole-job numbers: tokens summed over ranks; one record per rank all_gathered for the report
    _, tok_all = dp.aggregate(dist, elapsed, float(reduced * args.steps), dev)
    per_rank = dp.gather_records(dist, (rank, dev.index, L, L_out, elapsed / args.steps * 1e3), dev)
    B = 1

    result = None
    if rank == 0:
        # ---- per-kernel timing with events on the launch stream (stage entry points) -------------
        lib = _lib.load()
        sc = ff._scratch[(dev.type, dev.index)]
        order_buf = info["order"]
        stream = _lib.stream_ptr()
        elt = hidden.element_size()
        nv = info["nv"]
        sim = sc.sim(hidden.dtype, L)
        out_buf = torch.empty(1, L, d, dtype=hidden.dtype, device=dev)
        ptype_out = torch.empty(1, L, dtype=torch.int64, device=dev)
        cos_o, sin_o = torch.empty_like(cos), torch.empty_like(sin)
        aux = (_lib.FFAux * _lib.MAX_AUX)()
        aux[0] = _lib.FFAux(ptype.data_ptr(), ptype_out.data_ptr(), 8, 1)
        aux[1] = _lib.FFAux(cos.data_ptr(), cos_o.data_ptr(), HEAD_DIM * elt, 1)
        aux[2] = _lib.FFAux(sin.data_ptr(), sin_o.data_ptr(), HEAD_DIM * elt, 1)
        thr = float(torch.tensor(THRESHOLD, dtype=hidden.dtype))
        sub = float(ff._compute_pruning_ratio([], COST))
        rep_no = [0]

        def cur_hidden():           # same alternation as the timed loop (set per repetition below)
            return hidden_alt if rep_no[0] & 1 else hidden
        stages = {
            "order": lambda: lib.ff_build_order(ptype.data_ptr(), L, P, order_buf.data_ptr(), None, sc.stats.data_ptr(),
                                                sc.ws.data_ptr(), sc.ws_bytes, stream),
            "similarity": lambda: lib.ff_pair_similarity(cur_hidden().data_ptr(), _lib.FF_BF16, L, d, ptype.data_ptr(),
                                                         order_buf.data_ptr(), sc.stats.data_ptr(), sim.data_ptr(), stream),
            "plan": lambda: lib.ff_plan_merge(sim.data_ptr(), _lib.FF_BF16, order_buf.data_ptr(), L, thr, sub, RATIO_LB,
                                              sc.member.data_ptr(), sc.dst.data_ptr(), sc.keep.data_ptr(),
                                              sc.stats.data_ptr(), sc.ws.data_ptr(), sc.ws_bytes, stream),
            "merge_compact": lambda: lib.ff_merge_compact(cur_hidden().data_ptr(), out_buf.data_ptr(), _lib.FF_BF16, L, d, L,
                                                          order_buf.data_ptr(), sc.member.data_ptr(), 1, sc.dst.data_ptr(),
                                                          sc.keep.data_ptr(), aux, 3, stream),
        }
        # per-kernel durations: the four stages in the order of a real step (each kernel sees the
        # cache state its predecessor leaves behind), hipEvents between the stages on the launch
        # stream, launches queued ahead so host latency is not billed to a kernel
        reps = max(10, min(args.steps, 30))
        names = list(stages)
        for name in names:
            _lib.check(stages[name](), name)
        torch.cuda.synchronize()
        marks = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(reps)]
        for r in range(reps):
            rep_no[0] = r
            marks[r][0].record()
            for q, name in enumerate(names):
                _lib.check(stages[name](), name)
                marks[r][q + 1].record()
        torch.cuda.synchronize()
        kernel_us = {name: sum(marks[r][q].elapsed_time(marks[r][q + 1]) for r in range(reps)) / reps * 1e3
                     for q, name in enumerate(names)}
        alg = algorithmic_bytes(L, L_out, nv, d, elt, HEAD_DIM)
        dominant = max(("similarity", "merge_compact"), key=lambda k: kernel_us[k])
        # An event pair around ONE launch also times two command-processor round trips (~7 us here;
        # the rocprofv3 kernel trace does not see them).  For the roofline figure the dominant
        # kernel is therefore timed differentially, still in pipeline order and with the same cache
        # state: second pass with no event between it and its predecessor,
        #   t(kernel) = t(predecessor + kernel, one event pair) - t(predecessor, one event pair),
        # so the event overhead cancels and what remains is the launch duration plus the ~0.2 us
        # dependency gap between the two kernels.
        q_dom = names.index(dominant)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for r in range(reps):
            rep_no[0] = r
            for q, name in enumerate(names):
                if q == q_dom - 1:
                    pairs[r][0].record()
                _lib.check(stages[name](), name)
                if q == q_dom:
                    pairs[r][1].record()
        torch.cuda.synchronize()
        both_us = sum(a.elapsed_time(b) for a, b in pairs) / reps * 1e3
        dense_us = both_us - kernel_us[names[q_dom - 1]]
        achieved = alg[dominant] / (dense_us * 1e-6) / 1e9
        ms_per_step = t_max / args.steps * 1e3
        result = {
            "metric": BASELINE_METRIC,
            "value": tok_all / t_max,
            "unit": "tokens/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "per_rank": [{"rank": int(r[0]), "gpu": int(r[1]), "tokens_in": int(r[2]), "tokens_out": int(r[3]),
                          "ms_per_step": r[4]} for r in per_rank],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"C2: one FrameFusion.forward merge call on [1, {F}x{P}, {d}] bf16, "
                                   f"cost={COST} thr={THRESHOLD} lb={RATIO_LB}, p_change={args.p_change} "
                                   f"({'top-k' if info['branch'] else 'threshold'} branch), one sample per GPU",
                       "tokens_in": L, "tokens_out": L_out, "samples_per_gpu_per_step": B,
                       "tokens_processed_per_s": world * B * L * args.steps / t_max,
                       "parallelism": f"dp{world} (independent samples)"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": profiled_traffic(dominant) if (F, P, d) == (FRAMES, PATCHES, DIM) else None,
                         "algorithmic_bytes": alg[dominant], "kernel_us": dense_us,
                         "kernel_us_own_event_pair": kernel_us[dominant]},
            "kernels_us": kernel_us,
            "step_roofline": {"algorithmic_bytes": alg["step"], "achieved": alg["step"] / (ms_per_step * 1e-3) / 1e9,
                              "frac": alg["step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and (F, P, d) == (FRAMES, PATCHES, DIM):
            result["cascade"] = prefill_cascade(dev)
        if not args.no_cpu_baseline and world == 1:         # reported at N = 1 only (other ranks would wait)
            result["cpu_baseline"] = cpu_baseline(hidden, ptype, cos, sin, P, L, args.cpu_calls)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_*):
    2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request for 16 B/lane streams, MI355X_MICROARCH.md
    section HBM) + WRITE_SIZE, both reported in KiB.  None when no profile is committed."""
    import glob
    pat = {"merge_compact": "k_merge_compact", "similarity": "k_pair_similarity"}[kernel]
    total = 0.0
    for kind, scale in (("fetch", 2.0), ("write", 1.0)):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kind}_summary.csv")))
        if not files:
            return None
        val = None
        for line in open(files[-1]):
            if pat in line:
                val = float(line.strip().split(",")[-1])
        if val is None:
            return None
        total += scale * val * 1024.0
    return total


def prefill_cascade(dev, reps=8):
    """Extra (SURVEY.md §8d "the full cascade per sample"): every FrameFusion.forward call of ONE
    prefill in the threshold regime (p_change 0.5, 14 + 20 text tokens): call A merges, call B of
    layer 0 finds nothing left (identity), call B of layer 1 prunes with last-query importance.
    Host wall per call after a device synchronise, mean over `reps` prefills."""
    import framefusion_amd as ffa
    from framefusion_amd.synth import video_tokens, rotary_tables
    h0, pt = video_tokens(FRAMES, PATCHES, DIM, p_change=0.5, sigma=SIGMA, sigma_hi=1.6, seed=1234, pre=14, post=20,
                          dtype=torch.bfloat16, device=str(dev))
    L = h0.shape[1]
    cos, sin = rotary_tables(L, HEAD_DIM, torch.bfloat16, device=str(dev))
    ff = ffa.FrameFusion(COST, THRESHOLD, RATIO_LB)
    gen = torch.Generator(device=dev).manual_seed(7)
    acc = []
    for rep in range(reps + 2):
        ff.prepare(pt, PATCHES, 14, 14 + FRAMES * PATCHES - 1, FRAMES * PATCHES, L)
        h, pe, calls = h0, [cos, sin], []
        while not (ff.finish_merging and ff.finish_pruning) and len(calls) < 30:
            w = None
            if ff.finish_merging and not ff.finish_pruning:            # what the attention hook hands over
                w = torch.rand(1, 1, 1, h.shape[1], generator=gen, device=dev).to(torch.bfloat16)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_in = h.shape[1]
            h, pe, _ = ff(h, pe, None, w)
            torch.cuda.synchronize()
            calls.append((ff.last_call["kind"], n_in, h.shape[1], (time.perf_counter() - t0) * 1e6))
        if rep >= 2:
            acc.append(calls)
    n_calls = len(acc[0])
    per_call = [{"kind": acc[0][i][0], "tokens_in": acc[0][i][1], "tokens_out": acc[0][i][2],
                 "us": sum(c[i][3] for c in acc) / len(acc)} for i in range(n_calls)]
    total = sum(c["us"] for c in per_call)
    return {"workload": f"one prefill, [1, {L}, {DIM}] bf16, p_change=0.5 (threshold branch), importance = random [1,1,1,S]",
            "calls": per_call, "total_us": total, "tokens_reduced_per_s": (L - per_call[-1]["tokens_out"]) / (total * 1e-6)}


def cpu_baseline(hidden, ptype, cos, sin, P, L, calls):
    """The reference's torch-CPU path, as restated in oracle/ff_oracle.py ("port"), on this box's
    host cores: `calls` merge calls on the same sample (bounded: ~0.5-1 s each)."""
    from oracle import ff_oracle as orc
    h, pt, c, s = hidden.cpu(), ptype.cpu(), cos.cpu(), sin.cpu()
    threads = torch.get_num_threads()

    def one():
        f = orc.OracleFrameFusion(COST, THRESHOLD, RATIO_LB)
        f.prepare(pt, P, 0, L, L, L)
        o, _, _ = f.forward(h, [c, s], None)
        return o.shape[1]

    one()
    t0 = time.perf_counter()
    for _ in range(calls):
        lo = one()
    dt = time.perf_counter() - t0
    return {"value": (L - lo) * calls / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{calls} merge calls of the CPU oracle on the same [1, {L}, {hidden.shape[2]}] bf16 sample "
                      f"({dt / calls * 1e3:.0f} ms per call, torch {torch.__version__} CPU, {os.cpu_count()} logical cores)"}


if __name__ == "__main__":
    main()
