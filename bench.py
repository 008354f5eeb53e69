#!/usr/bin/env python
"""Headline benchmark: vision tokens reduced per second by one FrameFusion.forward merge call on
synthetic [1, 64 frames x 576 tokens, 4096] bf16 activations (BASELINE.json configs[1], "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` (N > 1) starts its own N ranks, one per GPU (framefusion_amd/dp.py: torch.distributed.run on
127.0.0.1, RCCL); it also runs unchanged under an external `python -m torch.distributed.run`.

A step = prepare() + one FrameFusion.forward call - three launches: K1 similarity (derives and verifies
the by-patch order from prepare()'s layout scalars, accumulates the select tables), the plan kernel, K4
merge + compaction - and one 256-byte result block, on one video sample resident in HBM.  With N ranks
each rank reduces its own independent sample (seed + rank): weak scaling, no data-path collective; rank
0's workload description is broadcast and the per-rank records are all_gathered.  Rank 0 prints ONE JSON
line.  `roofline` prices the dominant kernel (live hipEvent timing on the launch stream, algorithmic bytes
from DESIGN.md, `traffic` from two rocprofv3 --pmc passes of a short child run); `cpu_baseline` times the CPU oracle (oracle/ff_oracle.py, a torch-CPU port of the reference
path) on the same input; `eager_gpu_baseline` runs the same torch port on the MI355X (what the reference costs
through PyTorch-ROCm eager: the denominator of the >= 5x target); `extra.configs` times BASELINE.json's
other single-GPU configurations (C3, C5) and the real LLaVA-Video-7B shape.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAMES, PATCHES, DIM, HEAD_DIM = 64, 576, 4096, 128
COST, THRESHOLD, RATIO_LB = 0.3, 0.6, 0.1          # reference operating point (README.md:123)
P_CHANGE, SIGMA = 0.2, 0.3                         # top-k regime 36864 -> 11060
# The headline instance is what replace_framefusion_forward() builds (framefusion/interface.py:169-214): FrameFusion(cost,
# similarity_lower_bound, ratio_lower_bound) - exactly sized outputs like the reference's hidden_states[token_mask]
# (main.py:132-138).  VIEWS = the opt-in form that returns views of input-length buffers (what rounds 1-5 timed as the headline):
# extra.view_outputs_step_us, and the two-samples-in-flight mode (FrameFusionPair).
VIEWS = dict(compact_outputs=False)
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
BASELINE_METRIC = "vision tokens reduced/sec (64 frames×576 tok, d=4096 bf16), 1→8 MI355X"   # BASELINE.json


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
    ap.add_argument("--samples", type=int, default=0,
                    help="independent video samples per step over the whole job (default: one per GPU = the headline definition); "
                         "sample i goes to rank i %% N, a rank that owns two or more keeps two in flight (FrameFusionPair)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend for the scalars exchanged off the timed path (nccl = RCCL over xGMI)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs (rank r -> GPU r %% device_count; gloo only: RCCL refuses two ranks "
                         "on one device) - exercises the N > 1 path on a 1-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="form the process group even for one rank: the broadcast / all_gather / all_reduce of the report then "
                         "run through the backend (RCCL) on a 1-GPU box")
    ap.add_argument("--dp-hooks", default=None, metavar="MODULE",
                    help="tests only: import MODULE and let it install fault / delay hooks into framefusion_amd.dp")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline and eager_gpu_baseline")
    ap.add_argument("--no-extra", action="store_true", help="skip extra.configs (C3 / C5 / 7B shape)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect roofline.traffic live (two rocprofv3 --pmc passes of a short child run, ~25 s): take it from "
                         "the committed profiles/rNN_pmc_* summaries instead (withheld when they are of another build)")
    ap.add_argument("--e2e", nargs="?", const="7b", default=None, choices=sorted(E2E_SHAPES) + ["all"],
                    help="also run BASELINE.json configs[1] (7b) / configs[4] (72b) end to end: a random-weight Qwen2 LLM of that "
                         "LLaVA-Video model's shape, 64 frames prefilled dense / with this build / with the torch port of the reference "
                         "(7b adds ~1 min, 72b ~3 min and 150 GB of HBM)")
    ap.add_argument("--cpu-calls", type=int, default=12,
                    help="at most this many merge calls of the CPU oracle are timed for cpu_baseline, at the fastest thread count of "
                         "a short sweep (capped to ~6 s of CPU work)")
    return ap.parse_args()


def merge_bytes(L_in, L_out, d, elt, head_dim, pe_outer=1, pe_tables=2):
    """DESIGN.md §3 / SURVEY.md §8d: compulsory HBM traffic of one merge call that folds something."""
    return L_in * d * elt + L_out * d * elt + pe_tables * (L_in + L_out) * head_dim * elt * pe_outer + 8 * (L_in + L_out)


def call_bytes(kind, L_in, L_out, nv, d, elt, head_dim, pe_outer=1, pe_tables=2, kv_heads=0):
    """Compulsory HBM bytes of one FrameFusion.forward call, by what the call has to touch (reference main.py):
      merge that folds  : every input row read once, every output row written once, cos/sin in + out, 8-byte ints
                          (main.py:104-138; SURVEY §8d's formula);
      identity merge    : the similarity pass reads the Nv visual rows; nothing is folded, so nothing is written and the
                          caller keeps its tensors (main.py:264-266: an empty merge set returns the input);
      prune             : the K of the last-query importance (H_kv * S * dh), then ONLY the kept rows move - read and
                          written once each, with their cos/sin rows - plus ~5 bytes per position of importance / keep / dst
                          (main.py:61-101 gathers keep_indexs; dropped rows are never read: csrc/ff_merge_body.h `fold == DROP`).
    Rounds 1-4 charged every call (L_in + L_out) rows, which put the C5 prune gather above the 8 TB/s peak."""
    row = d * elt
    pe_row = pe_tables * head_dim * elt * pe_outer
    if kind == "merge" and L_out == L_in:
        return nv * row
    if kind == "merge":
        return merge_bytes(L_in, L_out, d, elt, head_dim, pe_outer, pe_tables)
    return kv_heads * L_in * head_dim * elt + 2 * L_out * (row + pe_row) + 5 * L_in


def algorithmic_bytes(L_in, L_out, nv, d, elt, head_dim, pe_outer=1):
    step = merge_bytes(L_in, L_out, d, elt, head_dim, pe_outer)
    return dict(step=step, similarity=nv * d * elt + nv * elt, merge_compact=step)


def main():
    args = parse()
    from framefusion_amd import dp
    if args.dp_hooks:
        dp.load_hooks(args.dp_hooks)
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
    # this rank's threads (interpreter + the C poll loop) onto the cores of its GPU's NUMA node, before anything pinned exists
    if world > 1 or os.environ.get("FF_DP_BIND") == "1":
        dp.bind_to_gpu_numa(dev, local)
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", str(dp.free_port()))
    dist = dp.init(args.backend, dev, force=args.force_dist)
    # rank 0's workload description is THE workload: broadcast over RCCL before anything is generated
    cfg = dp.broadcast_config(dist, dict(seed=args.seed, frames=args.frames, patches=args.patches, dim=args.dim,
                                         p_change=args.p_change, steps=args.steps, warmup=args.warmup, samples=args.samples), dev)
    args.seed, args.frames, args.patches, args.dim = cfg["seed"], cfg["frames"], cfg["patches"], cfg["dim"]
    args.p_change, args.steps, args.warmup, args.samples = cfg["p_change"], cfg["steps"], cfg["warmup"], cfg["samples"]
    mine = dp.shard(args.samples if args.samples > 0 else world, world, rank)        # the samples this rank owns (round-robin)
    if not mine:
        print(f"bench.py: --samples {args.samples} leaves rank {rank} without work", file=sys.stderr)
        sys.exit(2)

    import framefusion_amd as ffa
    from framefusion_amd import _lib
    from framefusion_amd.synth import video_tokens, rotary_tables

    F, P, d = args.frames, args.patches, args.dim
    hidden, ptype = video_tokens(F, P, d, p_change=args.p_change, sigma=SIGMA, seed=dp.sample_seed(args.seed, mine[0]),
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

    # Measurement order: ONE priming step (builds the scratch and the by-patch order the stage entry points need), the
    # per-kernel stage timing for the roofline figure, and only then the contract's loop, with no idle time in between:
    # W untimed steps, then EXACTLY K steps between barrier + synchronize; max over ranks.  The stage timing used to
    # run after the loop; an MI355X that has just left idle runs its steps 14..60 (2 - 8 ms after the first launch)
    # 5 - 10 % slower than the ones before and after them (tools/coldstart.py, profiles/r03_coldstart.txt), which is
    # exactly where a short run (--steps 20 --warmup 5) sits.  Every rank does the same so that all GPUs enter the
    # loop in the same state.
    step()
    info = dict(ff.last_call)
    kernel_us, dominant, dense_us = stage_times(ff, _lib, hidden, hidden_alt, ptype, cos, sin, P, L, d, info, dev, args.steps)
    timed_step, reduced_of = step, (lambda o: L - o.shape[1])
    if len(mine) > 1:
        # this rank owns several samples per step (C4 with more samples than GPUs): two in flight at a time, one host thread
        timed_step, reduced_of = multi_sample_step(ffa, dev, mine, args, F, P, d, L, cos, sin, hidden, hidden_alt, ptype, ff)
        timed_step()
        timed_step.drain()
    timed = dp.timed_steps(dist, timed_step, args.steps, args.warmup, dev)
    t_max, elapsed, out = timed
    reduced = reduced_of(out)
    out = out[0] if isinstance(out, list) else out
    L_out = out.shape[1]
    info = ff.last_call
    # whole-job numbers: tokens summed over ranks; one record per rank all_gathered for the report
    _, tok_all = dp.aggregate(dist, elapsed, float(reduced * args.steps), dev)
    su = timed.step_us
    per_rank = dp.gather_records(dist, (rank, dev.index, L, L_out, elapsed / args.steps * 1e3, dp.sample_seed(args.seed, rank),
                                        info["count"], su.get("min", 0.0), su.get("median", 0.0), su.get("p90", 0.0), su.get("max", 0.0)), dev)
    who = dp.gather_identities(dist, dev)          # hostname / pid / PCI address of every rank's GPU: N ranks = N distinct devices?
    # (a platform that reports no PCI address falls back on the UUID, then on the device index the rank bound itself to)
    distinct = len({(w.get("hostname"), w.get("pci_bus_id") or w.get("uuid") or f"index {int(r[1])}") for w, r in zip(who, per_rank)})
    backend_used = dist.get_backend() if dist is not None else None
    if world > 1 and not args.oversubscribe and (distinct != world or backend_used != "nccl"):
        # an N-GPU line must be N devices talking RCCL: anything else is a functional check and has to say so (--oversubscribe)
        if rank == 0:
            print(f"bench.py: --gpus {world} ran on {distinct} distinct device(s) over {backend_used}; one rank per GPU over "
                  f"RCCL (backend nccl) is required unless --oversubscribe is given", file=sys.stderr)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)
    # every rank's kept-token index list, padded to L int32: rank 0 reports a checksum per rank
    kept_idx = torch.nonzero(ff.last_plan()["keep"]).reshape(-1)
    kept_all = dp.gather_kept_indices(dist, kept_idx, L, dev)

    result = None
    if rank == 0:
        spread = step_spread(step, max(20, min(args.steps, 100)))
        alg = algorithmic_bytes(L, L_out, info["nv"], d, hidden.element_size(), HEAD_DIM)
        achieved = alg[dominant] / (dense_us * 1e-6) / 1e9
        ms_per_step = t_max / args.steps * 1e3
        headline = (F, P, d) == (FRAMES, PATCHES, DIM)
        traffic, traffic_source = (None, None)
        if headline:
            if world == 1 and not args.no_pmc:
                traffic, traffic_source = live_traffic(dominant)
            if traffic is None:
                live_note = traffic_source
                traffic, traffic_source = profiled_traffic(dominant)
                if live_note and traffic_source is not None:
                    traffic_source["live_attempt"] = live_note
        result = {
            "metric": BASELINE_METRIC,
            "value": tok_all / t_max,
            "unit": "tokens/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "ipc_mode": dp.ipc_mode(), "dp_attempt": dp.attempt(),
            "distinct_devices": distinct,
            "timed_region": "per rank: opening barrier -> K steps -> the rank's own synchronize; the closing barrier is outside "
                            "every clock; ms_per_step = max over ranks (all_reduce MAX)",
            "per_rank": [{"rank": int(r[0]), "gpu": int(r[1]), "tokens_in": int(r[2]), "tokens_out": int(r[3]),
                          "ms_per_step": r[4], "seed": int(r[5]), "similarities_above_threshold": int(r[6]), "hostname": w.get("hostname"), "pid": w.get("pid"),
                          "pci_bus_id": w.get("pci_bus_id"), "numa_node": w.get("numa_node"), "cpus": w.get("cpus"),
                          "step_us": {"min": r[7], "median": r[8], "p90": r[9], "max": r[10]},
                          "kept_indices": {"n": int(kx.numel()), "sum": int(kx.to(torch.int64).sum()),
                                           "first": kx[:4].tolist(), "last": kx[-4:].tolist()}}
                         for r, w, kx in zip(per_rank, who, kept_all)],
            "steps": args.steps,
            "warmup": args.warmup,
            "before_the_loop": "1 priming step + per-kernel stage timing (the roofline figure), no idle gap before the warm-up steps",
            "ms_per_step": ms_per_step,
            "step_us": spread,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"C2: one FrameFusion.forward merge call on [1, {F}x{P}, {d}] bf16, "
                                   f"cost={COST} thr={THRESHOLD} lb={RATIO_LB}, p_change={args.p_change} "
                                   f"({'top-k' if info['branch'] else 'threshold'} branch), the default instance "
                                   f"(exactly sized outputs, as replace_framefusion_forward builds it; views of input-length buffers: extra.view_outputs_step_us), "
                                   + ("one sample per GPU" if len(mine) == 1 else f"{len(mine)} samples per GPU, two in flight (FrameFusionPair)"),
                       "tokens_in": L, "tokens_out": L_out, "samples_per_gpu_per_step": len(mine),
                       "samples_per_step": args.samples if args.samples > 0 else world,
                       "tokens_processed_per_s": (args.samples if args.samples > 0 else world) * L * args.steps / t_max,
                       "parallelism": f"dp{world} (independent samples)"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": alg[dominant], "kernel_us": dense_us,
                         "kernel_us_own_event_pair": kernel_us[dominant]},
            "kernels_us": kernel_us,
            "step_roofline": {"algorithmic_bytes": alg["step"] * len(mine), "achieved": alg["step"] * len(mine) / (ms_per_step * 1e-3) / 1e9,
                              "frac": alg["step"] * len(mine) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and headline and not args.no_extra:
            # the same step with prepare() fed the way the reference's packers do it: start / end index as 1-element
            # DEVICE tensors (llava_video/modeling_llava_video.py:332-333), read back once inside prepare()
            start_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            end_dev = start_dev + L - 1

            def step_packer():
                flip[0] ^= 1
                ff.prepare(ptype, P, start_dev, end_dev, L, L)
                return ff(hidden_alt if flip[0] else hidden, [cos, sin], None)[0]
            ff_views = ffa.FrameFusion(COST, THRESHOLD, RATIO_LB, **VIEWS)  # the opt-in form: views of input-length buffers

            def step_views():
                flip[0] ^= 1
                ff_views.prepare(ptype, P, 0, L, L, L)
                return ff_views(hidden_alt if flip[0] else hidden, [cos, sin], None)[0]
            views = step_spread(step_views, 40)
            assert step().shape[1] == L_out and step().untyped_storage().nbytes() == L_out * d * hidden.element_size()
            result["extra"] = {"packer_scalars_step_us": step_spread(step_packer, 40), "exact_outputs_step_us": spread,
                               "view_outputs_step_us": views, "configs": extra_configs(dev)}
            result["extra"]["call_a_plus_call_b"] = call_a_plus_call_b(ffa, dev)
            torch.cuda.empty_cache()        # (the cascades above leave a zoo of cached block sizes behind)
            result["extra"]["two_samples_per_gpu"] = two_samples_per_gpu(ffa, dev, F, P, d, args.p_change, args.seed,
                                                                         max(20, min(args.steps, 100)), 10)
            assert ff.last_call["L_out"] == L_out
        if args.e2e and world == 1:
            for shape in (sorted(E2E_SHAPES) if args.e2e == "all" else [args.e2e]):
                result.setdefault("extra", {})[f"e2e_prefill_{shape}"] = e2e_prefill(dev, shape)
        if not args.no_cpu_baseline and world == 1:         # reported at N = 1 only (other ranks would wait)
            result["cpu_baseline"], result["eager_gpu_baseline"] = baselines(hidden, ptype, cos, sin, P, L, L_out,
                                                                              args.cpu_calls, ms_per_step)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def multi_sample_step(ffa, dev, mine, args, F, P, d, L, cos, sin, hidden0, hidden0_alt, ptype0, ff0):
    """(step, reduced_of) for a rank that owns the samples `mine` (indices into the job's sample list): every step reduces
    all of them - consecutive samples two at a time through FrameFusionPair (one host thread, two streams), an odd one out
    alone.  Each sample has its own instance and two copies of its input (alternated like the headline loop)."""
    from framefusion_amd import dp
    from framefusion_amd.synth import video_tokens
    work = [dict(h=hidden0, h2=hidden0_alt, pt=ptype0, ff=ff0)]
    for idx in mine[1:]:
        h, pt = video_tokens(F, P, d, p_change=args.p_change, sigma=SIGMA, seed=dp.sample_seed(args.seed, idx), dtype=torch.bfloat16,
                             device=str(dev))
        work.append(dict(h=h, h2=h.clone(), pt=pt, ff=ffa.FrameFusion(COST, THRESHOLD, RATIO_LB, **VIEWS)))
    # consecutive samples alternate between the two streams of a pair; a sample's call is collected only after the NEXT
    # sample's call has been submitted - across step boundaries too, so the last call of a step is collected by the next step
    # (`drain` collects the very last one: inside the timed region, before the closing synchronize)
    pairs = []
    for j in range(0, len(work), 2):
        second = work[j + 1]["ff"] if j + 1 < len(work) else ffa.FrameFusion(COST, THRESHOLD, RATIO_LB, **VIEWS)      # (odd one out: half a pair)
        pairs.append(ffa.FrameFusionPair(work[j]["ff"], second, dev, sync_with_current=False))
    flip, waiting, last = [0], [], [None] * len(work)

    def collect_one():
        j = waiting.pop(0)
        last[j] = pairs[j // 2].collect(j & 1)[0]

    def step():
        flip[0] ^= 1
        for j, w in enumerate(work):
            w["ff"].prepare(w["pt"], P, 0, L, L, L)
            pairs[j // 2].submit(j & 1, w["h2"] if flip[0] else w["h"], [cos, sin], None)
            waiting.append(j)
            if len(waiting) > 1:
                collect_one()
        return last

    def drain():
        while waiting:
            collect_one()
        return last
    step.drain = drain
    return step, (lambda outs: sum(L - o.shape[1] for o in outs))


def step_spread(step, n):
    """min / median / max of the step on the GPU clock: events between consecutive steps of a second, untimed
    run of the same loop (host and device in the same lock-step as the timed region)."""
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for _ in range(3):
        step()
    marks[0].record()
    for i in range(n):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    us = sorted(marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(n))
    return {"min": us[0], "median": statistics.median(us), "max": us[-1], "p90": us[int(0.9 * (n - 1))], "samples": n}


def stage_times(ff, _lib, hidden, hidden_alt, ptype, cos, sin, P, L, d, info, dev, steps):
    """Per-kernel durations through the stage entry points of the C ABI, in the order of a real step (each
    kernel sees the cache state its predecessor leaves behind), hipEvents on the launch stream, launches
    queued ahead so host latency is not billed to a kernel.  The stand-alone plan stage also builds the
    select tables and the inverse order itself (the fused step gets both from the similarity kernel): its
    number is an upper bound of the in-step plan kernel (profiles/ has the in-step timeline)."""
    lib = _lib.load()
    sc = ff._scratch[(dev.type, dev.index)]
    order_buf = info["order"]
    stream = _lib.stream_ptr()
    elt = hidden.element_size()
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
    reps = 60            # (~22 ms of the step's own kernels: the GPU has left its post-idle transient when the loop starts)
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
    return kernel_us, dominant, both_us - kernel_us[names[q_dom - 1]]


def live_traffic(kernel):
    """(bytes, source): HBM bytes per launch of `kernel` measured NOW, by two child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC has four counter slots, FETCH_SIZE
    takes three and WRITE_SIZE two - MI355X_MICROARCH.md "rocprofv3 PMC slots"), mean over the kernel's launches, corrected as
    that guide prescribes: 2 x FETCH_SIZE (gfx950 tallies 64 B per 128-B request of a 16 B / lane streaming read) + WRITE_SIZE, both
    reported in KiB.  (None, why) when rocprofv3 is not there, a pass fails or takes longer than two minutes."""
    import csv, glob, shutil, signal, subprocess, tempfile
    pat = {"merge_compact": "k_merge_compact<", "similarity": "k_pair_similarity<"}[kernel]
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"live": False, "why": "rocprofv3 not on PATH"}
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return None, {"live": False, "why": "this run is itself being profiled: no nested rocprofv3"}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extra", "--no-pmc"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    means, launches = {}, {}
    work = tempfile.mkdtemp(prefix="ff_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=120)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, {"live": False, "why": f"the {counter} pass did not finish within 120 s"}
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, {"live": False, "why": f"the {counter} pass failed (exit {rc}, {len(files)} counter files)"}
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not vals:
                return None, {"live": False, "why": f"no {counter} rows for {pat}"}
            means[counter], launches[counter] = sum(vals) / len(vals), len(vals)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    total = 2.0 * means["FETCH_SIZE"] * 1024.0 + means["WRITE_SIZE"] * 1024.0
    return total, {"live": True, "how": "two child runs of this script under rocprofv3 --kernel-trace --pmc <counter> (separate passes), mean "
                                       "over the kernel's launches; bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB, gfx950 correction)",
                   "child": " ".join(child[1:]), "fetch_kib_mean": means["FETCH_SIZE"], "write_kib_mean": means["WRITE_SIZE"],
                   "launches": launches}


def profiled_traffic(kernel):
    """(bytes, source): HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_*):
    2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request for 16 B/lane streams, MI355X_MICROARCH.md section HBM) +
    WRITE_SIZE, both reported in KiB.  The passes are separate rocprofv3 runs (tools/prof_round.sh), not this run:
    `source` names the files and the build they profiled (profiles/rNN_pmc_meta.json: hash of the kernel sources,
    commit); when that build is not the one running now the number is withheld (None) instead of reported stale."""
    import glob
    from framefusion_amd import _lib
    pat = {"merge_compact": "k_merge_compact", "similarity": "k_pair_similarity"}[kernel]
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_meta.json")))
    if not metas:
        return None, {"error": "no profiles/r*_pmc_meta.json (run tools/prof_round.sh)"}
    meta = json.load(open(metas[-1]))
    tag = os.path.basename(metas[-1]).split("_pmc_meta")[0]
    source = {"files": [f"profiles/{tag}_pmc_fetch_summary.csv", f"profiles/{tag}_pmc_write_summary.csv"],
              "profiled_source_hash": meta.get("source_hash"), "profiled_commit": meta.get("commit"),
              "running_source_hash": _lib.source_hash()}
    if meta.get("source_hash") != _lib.source_hash():
        source["stale"] = "the kernel sources changed since the PMC passes: traffic withheld"
        return None, source
    total = 0.0
    for kind, scale in (("fetch", 2.0), ("write", 1.0)):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{kind}_summary.csv")
        if not os.path.exists(path):
            return None, source
        val = None
        for line in open(path):
            if pat in line:
                val = float(line.strip().split(",")[-1])
        if val is None:
            return None, source
        total += scale * val * 1024.0
    return total, source


def profiled_kernel_us(cfg):
    """Sum of the kernel durations of one back-to-back cascade of configuration `cfg` from the committed per-kernel timeline
    (profiles/rNN_timeline_<cfg>.txt: rocprofv3 kernel trace of tools/trace_config.py, tools/prof_configs.sh) - the part of
    `us_back_to_back` that is GPU work; the rest is gaps the host leaves.  Like roofline.traffic it is withheld when the
    timeline was taken on other kernel sources."""
    import glob, re
    from framefusion_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_timeline_{cfg}.txt")))
    if not files:
        return None, None
    text = open(files[-1]).read()
    src = {"file": os.path.relpath(files[-1], ROOT)}
    m = re.search(r"sources ([0-9a-f]{16})", text)
    if m is None or m.group(1) != _lib.source_hash():
        src["stale"] = "the kernel sources changed since this timeline was taken: kernel_us withheld"
        return None, src
    if "k_merge_resident" in text:
        src["note"] = ("the one-launch kernel waits inside the kernel for the host's mail: its traced duration contains the host's "
                       "reaction time (lengthened by the tracer); profiles/*_resident_stamps.txt has its device time")
    m = re.search(r"kernel_us ([0-9.]+)", text)
    return (float(m.group(1)) if m else None), src


def two_samples_per_gpu(ffa, dev, F, P, d, p_change, seed, steps, warmup):
    """Two independent samples in flight on ONE GPU, two ways: from ONE host thread through FrameFusionPair (submit / collect,
    two HIP streams), and from two threads, each with its own FrameFusion instance, stream and sample (the deployment the
    reference's demo uses for its replicas, llava_video_compare.py:217-223; the result-block poll runs in C with the interpreter
    lock released).  Either way one sample's similarity / merge pass fills the other's plan-kernel bubble and kernel ramps.
    Whole-GPU throughput of `steps` steps per sample; NOT the headline (one sample per GPU per step)."""
    import threading
    from framefusion_amd.synth import video_tokens, rotary_tables
    from framefusion_amd import pair as pair_mod
    work = []
    for t in range(2):
        h, pt = video_tokens(F, P, d, p_change=p_change, sigma=SIGMA, seed=seed + 100 + t, dtype=torch.bfloat16, device=str(dev))
        L = h.shape[1]
        cos, sin = rotary_tables(L, HEAD_DIM, torch.bfloat16, device=str(dev))
        work.append(dict(h=h, h2=h.clone(), pt=pt, cos=cos, sin=sin, L=L, ff=ffa.FrameFusion(COST, THRESHOLD, RATIO_LB, **VIEWS),
                         stream=torch.cuda.Stream(device=dev), out=None))
    def fresh(**kw):
        for w in work:
            w["ff"] = ffa.FrameFusion(COST, THRESHOLD, RATIO_LB, **(kw or VIEWS))

    def one_thread():
        # FrameFusionPair - sample 1's call is submitted (ff_ctx_merge_submit, its own stream) before sample 0's is collected and
        # vice versa: always one call enqueued ahead of the one being waited for
        pair = ffa.FrameFusionPair(work[0]["ff"], work[1]["ff"], dev, sync_with_current=False)

        def calls(n):
            for i in range(2 * n):
                w = work[i & 1]
                w["ff"].prepare(w["pt"], P, 0, w["L"], w["L"], w["L"])
                yield (i & 1, w["h2"] if (i >> 1) & 1 else w["h"], [w["cos"], w["sin"]], None)
        for _ in pair.run(calls(warmup)):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [None, None]
        for i, o in enumerate(pair.run(calls(steps))):
            outs[i & 1] = o
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, sum(w["L"] - o[0].shape[1] for w, o in zip(work, outs))

    def two_threads():
        start, stop = threading.Barrier(3), threading.Barrier(3)

        def run(w):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(w["stream"]):
                def step(i):
                    w["ff"].prepare(w["pt"], P, 0, w["L"], w["L"], w["L"])
                    return w["ff"](w["h2"] if i & 1 else w["h"], [w["cos"], w["sin"]], None)[0]
                for i in range(warmup):
                    step(i)
                w["stream"].synchronize()
                start.wait()
                for i in range(steps):
                    w["out"] = step(i)
                w["stream"].synchronize()
                stop.wait()
        threads = [threading.Thread(target=run, args=(w,)) for w in work]
        for th in threads:
            th.start()
        start.wait()
        t0 = time.perf_counter()
        stop.wait()
        dt = time.perf_counter() - t0
        for th in threads:
            th.join()
        return dt, sum(w["L"] - w["out"].shape[1] for w in work)

    # the two forms alternate twice (whichever runs second in a process has measured up to 12 % slower - allocator pools of
    # other streams, not the form): every number is reported, the summary is each form's best
    runs = {"one_host_thread_pair": [], "two_threads": []}
    fresh()
    one_thread()                            # (untimed: the pair streams' allocator pools get their blocks now - a first round measured
    #                                          243-267 us where every later one of the same process measures 239-241)
    for _ in range(2):
        for name, fn in (("one_host_thread_pair", one_thread), ("two_threads", two_threads)):
            fresh()
            dt, reduced = fn()
            runs[name].append({"us_per_pair_of_steps": dt / steps * 1e6, "tokens_reduced_per_s": reduced * steps / dt})
    best = {k: max(v, key=lambda r: r["tokens_reduced_per_s"]) for k, v in runs.items()}
    fresh(compact_outputs=True)             # the default exactly sized outputs: K1 + plan at submit, merge kernel at collect
    dt, reduced = one_thread()
    exact = {"us_per_pair_of_steps": dt / steps * 1e6, "tokens_reduced_per_s": reduced * steps / dt}
    return {"samples_in_flight": 2, "steps_per_sample": steps, "order": "(one untimed pair round,) pair, threads, pair, threads (fresh instances each time)",
            "one_host_thread_pair": {**best["one_host_thread_pair"], "all_us": [r["us_per_pair_of_steps"] for r in runs["one_host_thread_pair"]],
                                     "how": "framefusion_amd.FrameFusionPair (ff_ctx_merge_submit / _collect), two HIP streams, no threads"},
            "two_threads": {**best["two_threads"], "all_us": [r["us_per_pair_of_steps"] for r in runs["two_threads"]]},
            "one_host_thread_pair_exact_outputs": exact,
            "pair_streams": sorted(set(pair_mod.STREAM_SOURCE.values())),
            "us_per_pair_of_steps": best["one_host_thread_pair"]["us_per_pair_of_steps"],
            "tokens_reduced_per_s": best["one_host_thread_pair"]["tokens_reduced_per_s"]}


def cascade(ffa, dev, F, P, d, p_change, thr, pre, post, heads, kv_heads, num, mrope, sigma_hi=1.6, reps=6, seed=1234,
            idle_before_b2b_s=0.0, defer=True):
    """Every FrameFusion.forward call of ONE prefill (call A, then call B per layer until merging and pruning are
    finished), the importance of the prune call computed by the HIP attention-hook kernel from synthetic q / K
    (un-repeated GQA heads) - `defer`: inside the prune's own host call (the handle the adapters in framefusion_amd/models pass;
    round 5), else by the hook's own call before it (rounds 1-4).  GPU time of the whole cascade (one synchronise at the end), mean over `reps`."""
    from framefusion_amd.synth import video_tokens, rotary_tables
    h0, pt = video_tokens(F, P, d, p_change=p_change, sigma=SIGMA, sigma_hi=sigma_hi, seed=seed, pre=pre, post=post,
                          dtype=torch.bfloat16, device=str(dev))
    L = h0.shape[1]
    pe0 = rotary_tables(L, HEAD_DIM, torch.bfloat16, device=str(dev), mrope=mrope)
    gen = torch.Generator(device=dev).manual_seed(7)
    q = torch.randn(1, heads, num, HEAD_DIM, generator=gen, device=dev).to(torch.bfloat16)
    k_full = torch.randn(1, kv_heads, L, HEAD_DIM, generator=gen, device=dev).to(torch.bfloat16)
    ff = ffa.FrameFusion(COST, thr, RATIO_LB)
    elt, pe_outer = 2, (3 if mrope else 1)
    times, calls, bytes_alg = [], [], 0
    k_of = {L: k_full}                              # the layer's keys at the current sequence length
    def one_prefill():
        nonlocal calls, bytes_alg
        ff.prepare(pt, P, pre, pre + F * P - 1, F * P, L)
        h, pe = h0, [t for t in pe0]
        calls, bytes_alg = [], 0
        while not (ff.finish_merging and ff.finish_pruning) and len(calls) < 30:
            n_in = h.shape[1]
            w = None
            if ff.finish_merging and not ff.finish_pruning:            # what the attention hook hands over
                if n_in not in k_of:
                    k_of[n_in] = k_full[:, :, :n_in].contiguous()
                w = ffa.last_query_importance(q, k_of[n_in], num=num, is_causal=True, framefusion=ff, defer=defer)
            h, pe, _ = ff(h, pe, None, w)
            calls.append((ff.last_call["kind"], n_in, h.shape[1]))
            bytes_alg += call_bytes(ff.last_call["kind"], n_in, h.shape[1], ff.last_call["nv"], d, elt, HEAD_DIM, pe_outer,
                                    kv_heads=kv_heads)

    for rep in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_prefill()
        torch.cuda.synchronize()
        if rep >= 2:
            times.append((time.perf_counter() - t0) * 1e6)
    # the same cascade issued back to back (no idle GPU in front of a call, no drain behind it): what a prefill pays
    # when the calls sit between other kernels of the model
    n_b2b = 4 * reps
    # (one untimed pass first: with the host running ahead more output buffers are alive at once than in the isolated
    # runs above, and the allocator's first hipMalloc for them - tens of ms - would land in the timed pass: it did, once,
    # 3.8 ms "per cascade" at the 72B shape)
    for _ in range(n_b2b):
        one_prefill()
    torch.cuda.synchronize()
    if idle_before_b2b_s:                # (tools/trace_config.py: an idle gap that marks the block in a kernel trace)
        time.sleep(idle_before_b2b_s)
    t0 = time.perf_counter()
    for _ in range(n_b2b):
        one_prefill()
    torch.cuda.synchronize()
    us_b2b = (time.perf_counter() - t0) * 1e6 / n_b2b
    us = statistics.median(times)
    L_final = calls[-1][2]
    return {"tokens_in": L, "tokens_out": L_final, "calls": [f"{k}:{a}->{b}" for k, a, b in calls], "us": us,
            "us_back_to_back": us_b2b,
            "tokens_reduced_per_s": (L - L_final) / (us * 1e-6), "algorithmic_bytes": bytes_alg,
            "algorithmic_bytes_rule": "compulsory per call kind (bench.call_bytes): folding merge (L_in + L_out) rows; identity "
                                      "merge Nv rows read; prune K + 2 x L_out rows",
            "hbm_frac": bytes_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "hbm_frac_back_to_back": bytes_alg / (us_b2b * 1e-6) / 1e9 / HBM_PEAK_GBS}


def call_a_plus_call_b(ffa, dev, reps=100, rounds=2):
    """What the adapters in framefusion_amd/models issue at layer 0 of a prefill in the threshold regime: call A (merge), then call B
    with the decoder's residual add fused in (rows = T(attn_out + residual), modeling_qwen2.py:64-67) - LLaVA-Video-7B layout,
    prefills back to back, host included; the one-launch kernel (call B through its sums instance) against the three launches,
    same process, alternating."""
    from framefusion_amd.synth import video_tokens, rotary_tables
    F, P, d, pre, post = 64, 210, 3584, 14, 20
    h0, pt = video_tokens(F, P, d, p_change=0.5, sigma=SIGMA, sigma_hi=1.6, seed=1234, pre=pre, post=post, dtype=torch.bfloat16, device=str(dev))
    L = h0.shape[1]
    pe0 = rotary_tables(L, HEAD_DIM, torch.bfloat16, device=str(dev))
    ff = ffa.FrameFusion(COST, THRESHOLD, 0.02)
    attn, calls = {}, []

    def prefill():
        ff.prepare(pt, P, pre, pre + F * P - 1, F * P, L)
        h, pe, _ = ff(h0, list(pe0), None)
        calls[:] = [f"A:{L}->{h.shape[1]}" + ("*" if ff.last_call["one_launch"] else "")]
        n = h.shape[1]
        if n not in attn:
            attn[n] = torch.randn(1, n, d, device=dev, dtype=torch.bfloat16) * 0.1
        out, _, _ = ff(attn[n], pe, None, None, residual=h)
        calls.append(f"B:{n}->{out.shape[1]}" + ("*" if ff.last_call["one_launch"] else ""))
    us = {True: [], False: []}
    seen = {}
    was = ffa.FrameFusion.one_launch
    try:
        for _ in range(rounds):
            for one in (True, False):
                ffa.FrameFusion.one_launch = one
                for _ in range(20):
                    prefill()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    prefill()
                torch.cuda.synchronize()
                us[one].append((time.perf_counter() - t0) / reps * 1e6)
                seen[one] = list(calls)
    finally:
        ffa.FrameFusion.one_launch = was
    return {"workload": "LLaVA-Video-7B layout [1, 14+64x210+20, 3584] bf16, p_change=0.5: call A, then call B with residual=, per prefill, "
                        "back to back (* = went out as the one-launch kernel)",
            "calls_default": seen[True], "calls_three_launches": seen[False],
            "us_per_prefill_default": us[True], "us_per_prefill_three_launches": us[False]}


def extra_configs(dev):
    """BASELINE.json's other single-GPU configurations, each as a whole prefill cascade (host included: the
    calls are issued back to back, one synchronise at the end)."""
    import framefusion_amd as ffa
    out = []
    # C2 in the threshold regime (merge, identity, prune)
    def with_kernel_us(r, cfg):
        r["kernel_us"], r["kernel_us_source"] = profiled_kernel_us(cfg)
        return r
    r = cascade(ffa, dev, FRAMES, PATCHES, DIM, 0.5, THRESHOLD, 14, 20, 32, 8, 1, False)
    out.append({"workload": "C2 cascade: [1, 14+64x576+20, 4096] bf16, p_change=0.5 (threshold branch), thr=0.6", **with_kernel_us(r, "c2thr")})
    # C3: Qwen2-VL-7B, 128 frames = 64 temporal grids x 195 tokens, M-RoPE containers, num = 4 importance, threshold sweep
    sweep = []
    for thr in (0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9):
        r = cascade(ffa, dev, 64, 195, 3584, 0.5, thr, 15, 12, 28, 4, 4, True, sigma_hi=1.8, reps=4, seed=77)
        sweep.append({"similarity_lower_bound": thr, **(with_kernel_us(r, "c3") if thr == 0.6 else r)})
    tot_us = sum(s["us"] for s in sweep)
    out.append({"workload": "C3: Qwen2-VL-7B shape [1, 15+64x195+12, 3584] bf16, M-RoPE [3,1,L,128], num=4, "
                            "similarity_lower_bound sweep 0.3..0.9 (one prefill cascade each)",
                "sweep": sweep, "us": tot_us,
                "tokens_reduced_per_s": sum(s["tokens_in"] - s["tokens_out"] for s in sweep) / (tot_us * 1e-6),
                "hbm_frac": sum(s["algorithmic_bytes"] for s in sweep) / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS})
    # C5: LLaVA-Video-72B, d = 8192, H = 64 / H_kv = 8: merge, then importance inside the attention hook + prune
    r = cascade(ffa, dev, 64, 576, 8192, 0.95, THRESHOLD, 14, 20, 64, 8, 1, False, sigma_hi=None)
    out.append({"workload": "C5: LLaVA-Video-72B shape [1, 14+64x576+20, 8192] bf16, H=64/H_kv=8: merge call, fused "
                            "attention importance + prune", **with_kernel_us(r, "c5")})
    r = cascade(ffa, dev, 64, 576, 8192, 0.2, THRESHOLD, 14, 20, 64, 8, 1, False, sigma_hi=None)
    out.append({"workload": "C5 top-k regime: [1, 14+64x576+20, 8192] bf16, one merge call", **with_kernel_us(r, "c5topk")})
    # the real LLaVA-Video-7B token layout (14 x 15 per frame, d = 3584): the host, not HBM, sets the pace here
    r = cascade(ffa, dev, 64, 210, 3584, 0.2, THRESHOLD, 14, 20, 28, 4, 1, False, sigma_hi=None)
    out.append({"workload": "LLaVA-Video-7B real shape [1, 14+64x210+20, 3584] bf16, one merge call (top-k)", **with_kernel_us(r, "7b")})
    return out


class _EagerFrameFusion(torch.nn.Module):
    """The torch port of the reference's FrameFusion (oracle/ff_oracle.py, here on GPU tensors) behind the attribute surface
    the adapters use - the "reference eager" leg of e2e_prefill (baseline code: never part of the product path)."""
    supports_residual = False

    def __init__(self, oracle):
        super().__init__()
        self.__dict__["o"] = oracle
        self.time_s = 0.0

    def __getattr__(self, name):
        return getattr(self.__dict__["o"], name)

    def prepare(self, *a, **kw):
        return self.o.prepare(*a, **kw)

    def _expect_importance(self, *a, **kw):
        return None, None

    def forward(self, hidden_states, position_embeddings, attention_mask, self_attn_weights=None):
        active = hidden_states.shape[1] > 1 and not (self.o.finish_merging and self.o.finish_pruning)
        if active:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = self.o.forward(hidden_states, position_embeddings, attention_mask, self_attn_weights)
        if active:
            torch.cuda.synchronize()
            self.time_s += time.perf_counter() - t0
        return out


E2E_SHAPES = {   # name: (d, layers, heads, kv heads, MLP, patches per frame, reps, family)
    "7b": (3584, 28, 28, 4, 18944, 210, 3, "qwen2"),         # LLaVA-Video-7B-Qwen2 (BASELINE configs[1]); 14 x 15 tokens per frame
    "72b": (8192, 80, 64, 8, 29568, 576, 1, "qwen2"),        # LLaVA-Video-72B's LLM (configs[4]): 145 GB of bf16 weights on ONE MI355X
    # Qwen2-VL-7B (configs[2]): 128 frames = 64 temporal grids x 195 merged patches (13 x 15), M-RoPE [3, 1, L, 128] position
    # embeddings, num = 4 importance queries; the regimes are a similarity_lower_bound sweep instead of two p_change values
    "qwen2vl": (3584, 28, 28, 4, 18944, 195, 2, "qwen2_vl"),
}
QWEN2VL_SWEEP = (0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)


def mrope_position_ids(frames, grid_h, grid_w, pre, post, device):
    """[3, 1, L] (temporal, height, width) position ids of a Qwen2-VL prompt with one video: text tokens advance all three
    together, a video token takes (start + frame, start + row, start + column), the text behind continues from the largest id
    (what transformers' Qwen2VLModel.get_rope_index computes for such a prompt)."""
    t = torch.arange(frames).view(-1, 1, 1).expand(frames, grid_h, grid_w).reshape(-1)
    h = torch.arange(grid_h).view(1, -1, 1).expand(frames, grid_h, grid_w).reshape(-1)
    w = torch.arange(grid_w).view(1, 1, -1).expand(frames, grid_h, grid_w).reshape(-1)
    vis = torch.stack([t, h, w]) + pre
    head = torch.arange(pre).view(1, -1).expand(3, -1)
    nxt = int(vis.max()) + 1
    tail = (torch.arange(post) + nxt).view(1, -1).expand(3, -1)
    return torch.cat([head, vis, tail], dim=1).view(3, 1, -1).to(device)


def e2e_prefill(dev, shape="7b", frames=64, pre=14, post=20, regimes=((P_CHANGE, None), (0.5, 1.6))):
    """BASELINE.json configs[1] end to end ("LLaVA-Video-7B-Qwen2, 64 frames, cost=0.3, 1xMI355X bf16 - HIP sim+merge vs
    reference eager"): a random-weight Qwen2 decoder stack of that model's LLM shape (d = 3584, 28 layers, 28 / 4 heads,
    MLP 18944; no checkpoint exists offline) prefills 14 + 64 x 210 + 20 synthetic tokens (a) dense, (b) patched with
    apply_framefusion of THIS build, (c) with the same adapter driving the torch port of the reference's FrameFusion on the
    GPU.  Wall time of the LLM prefill (synchronised), per-layer sequence lengths, and the time spent inside
    FrameFusion.forward.  shape "72b": configs[4]; shape "qwen2vl": configs[2] - transformers' Qwen2VLTextModel (M-RoPE,
    num = 4 importance queries) of Qwen2-VL-7B's shape over a similarity_lower_bound sweep."""
    from transformers.cache_utils import DynamicCache
    import framefusion_amd as ffa
    from framefusion_amd.synth import video_tokens
    from oracle import ff_oracle as orc
    d, n_layers, n_heads, n_kv, mlp, patches, reps, family = E2E_SHAPES[shape]
    vl = family == "qwen2_vl"
    if vl:
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextModel
        from framefusion_amd.models.qwen2_vl import register_hf_qwen2_vl
        pre, post = 15, 12
        cfg = Qwen2VLTextConfig(vocab_size=1024, hidden_size=d, intermediate_size=mlp, num_hidden_layers=n_layers,
                                num_attention_heads=n_heads, num_key_value_heads=n_kv, max_position_embeddings=65536,
                                rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]})
    else:
        from transformers import Qwen2Config, Qwen2ForCausalLM
        from framefusion_amd.models.qwen2 import register_hf_qwen2
        cfg = Qwen2Config(vocab_size=1024, hidden_size=d, intermediate_size=mlp, num_hidden_layers=n_layers, num_attention_heads=n_heads,
                          num_key_value_heads=n_kv, max_position_embeddings=65536, rope_theta=1000000.0)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)          # (the 72B stack is 145 GB in bf16: it must never exist in fp32)
    try:
        with torch.device(dev):
            if vl:
                class Holder(torch.nn.Module):       # the bare text-decoder wrapper of models/qwen2_vl.py ("hf_qwen2_vl_text")
                    def __init__(self):
                        super().__init__()
                        self.model = Qwen2VLTextModel(cfg)
                model = Holder().eval()
            else:
                model = Qwen2ForCausalLM(cfg).eval()
    finally:
        torch.set_default_dtype(prev)
    n_vis = frames * patches
    L = pre + n_vis + post
    state = {"pos": mrope_position_ids(frames, 13, 15, pre, post, dev) if vl else None}

    def prefill(prepare):
        times, lengths = [], None
        for _ in range(reps + 1):
            if prepare is not None:
                prepare()
            cache = DynamicCache(config=cfg)
            kw = {"position_ids": state["pos"]} if vl else {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                out = model.model(inputs_embeds=state["emb"], past_key_values=cache, use_cache=True, **kw)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            lengths = getattr(model.model, "framefusion_lengths", None) or [out.last_hidden_state.shape[1]] * cfg.num_hidden_layers
            del cache, out
        return statistics.median(times[1:]), lengths

    def runs(lengths):                 # [a, a, a, b, b] -> "3 x a, 2 x b"
        out, i = [], 0
        while i < len(lengths):
            j = i
            while j < len(lengths) and lengths[j] == lengths[i]:
                j += 1
            out.append(f"{j - i} x {lengths[i]}")
            i = j
        return ", ".join(out)

    if vl:
        regimes = tuple((0.5, 1.8, thr) for thr in QWEN2VL_SWEEP)        # the C3 sample of extra_configs, threshold swept
        seed = 77
    else:
        regimes = tuple((pc, sh, THRESHOLD) for pc, sh in regimes)
        seed = 1234
    state["emb"], _ = video_tokens(frames, patches, d, p_change=regimes[0][0], sigma=SIGMA, sigma_hi=regimes[0][1], seed=seed,
                                   pre=pre, post=post, dtype=torch.bfloat16, device=str(dev))
    dense_ms, _ = prefill(None)
    (register_hf_qwen2_vl if vl else register_hf_qwen2)()
    ffa.apply_framefusion(model, cost=COST, similarity_lower_bound=THRESHOLD, ratio_lower_bound=RATIO_LB)
    hip_ff = model.framefusion

    def install(obj):
        model.framefusion = obj
        model.model.framefusion = obj
        for layer in model.model.layers:
            layer.framefusion = obj
            layer.self_attn.framefusion = obj
    results = []
    for p_change, sigma_hi, thr in regimes:
        emb, pt = video_tokens(frames, patches, d, p_change=p_change, sigma=SIGMA, sigma_hi=sigma_hi, seed=seed, pre=pre, post=post,
                               dtype=torch.bfloat16, device=str(dev))
        state["emb"] = emb
        hip_ff.similarity_lower_bound = thr
        eager = _EagerFrameFusion(orc.OracleFrameFusion(COST, thr, RATIO_LB))
        install(hip_ff)
        # both output forms (INTEGRATION.md "Output buffers"): exactly sized tensors like the reference's (compact_outputs, the
        # default since round 5) and views of input-length buffers (the opt-in fast path) - prefill time and peak memory of each
        modes = {}
        for compact in (True, False):
            hip_ff.compact_outputs = compact
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)
            ms, lens = prefill(lambda: hip_ff.prepare(pt, patches, pre, pre + n_vis - 1, n_vis, L))
            modes[compact] = (ms, lens, (torch.cuda.max_memory_allocated(dev) - base) / 2 ** 30)
        hip_ff.compact_outputs = type(hip_ff)().compact_outputs          # back to the class default for the legs below
        hip_ms, hip_lengths, _ = modes[hip_ff.compact_outputs]
        # wall time of the FrameFusion.forward calls of this build (each bracketed by synchronise: launch-from-idle and drain
        # included - an upper bound of what they add to the prefill)
        calls = []
        inner = hip_ff.forward

        def timed(*a, **kw):
            active = a[0].shape[1] > 1 and not (hip_ff.finish_merging and hip_ff.finish_pruning)
            if active:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            r = inner(*a, **kw)
            if active:
                torch.cuda.synchronize()
                calls.append((time.perf_counter() - t0) * 1e3)
            return r
        hip_ff.forward = timed
        prefill(lambda: (calls.clear(), hip_ff.prepare(pt, patches, pre, pre + n_vis - 1, n_vis, L)))
        hip_ff_ms, hip_ff_calls = sum(calls), len(calls)
        del hip_ff.forward                      # back to the class's forward
        install(eager)

        def prep_eager():
            eager.time_s = 0.0
            eager.prepare(pt, patches, pre, pre + n_vis - 1, n_vis, L)
        eager_ms, eager_lengths = prefill(prep_eager)
        results.append({"p_change": p_change, "similarity_lower_bound": thr, "hip_prefill_ms": hip_ms,
                        "hip_prefill_ms_exact_outputs": modes[True][0], "hip_prefill_ms_view_outputs": modes[False][0],
                        "prefill_peak_gib_above_weights_exact_outputs": modes[True][2],
                        "prefill_peak_gib_above_weights_view_outputs": modes[False][2],
                        "eager_reference_prefill_ms": eager_ms,
                        "prefill_speedup_vs_dense": dense_ms / hip_ms, "prefill_speedup_vs_eager_reference": eager_ms / hip_ms,
                        "framefusion_calls": hip_ff_calls, "hip_ms_inside_framefusion": hip_ff_ms,
                        "eager_ms_inside_framefusion": eager.time_s * 1e3,
                        "mean_kept_fraction_over_layers": sum(hip_lengths) / (len(hip_lengths) * L),
                        "lengths_hip": runs(hip_lengths), "lengths_eager": runs(eager_lengths)})
    del model
    torch.cuda.empty_cache()
    name = "Qwen2-VL-7B's text decoder (Qwen2VLTextModel, M-RoPE, num=4 importance queries)" if vl else f"Qwen2 LLM of LLaVA-Video-{shape.upper()}'s shape"
    return {"workload": f"{name} (d={d}, {n_layers} layers, {n_heads}/{n_kv} heads, MLP {mlp}, random "
                        f"weights), prefill of {pre}+{frames}x{patches}+{post} = {L} synthetic tokens, cost={COST}"
                        + (", similarity_lower_bound sweep" if vl else f", thr={THRESHOLD}"),
            "dense_prefill_ms": dense_ms, "regimes": results}


def baselines(hidden, ptype, cos, sin, P, L, L_out, calls, hip_ms):
    """(cpu_baseline, eager_gpu_baseline): the reference's torch path, as restated in oracle/ff_oracle.py
    ("port"), timed on this box's host cores and - the same code on GPU tensors - on the MI355X: what the
    reference costs through PyTorch-ROCm eager (~450 aten dispatches, 17 host syncs per call)."""
    from oracle import ff_oracle as orc
    h, pt, c, s = hidden.cpu(), ptype.cpu(), cos.cpu(), sin.cpu()

    def one(hh, pp, cc, ss):
        f = orc.OracleFrameFusion(COST, THRESHOLD, RATIO_LB)
        f.prepare(pp, P, 0, L, L, L)
        o, _, _ = f.forward(hh, [cc, ss], None)
        return o.shape[1]

    # The index-heavy torch-CPU path does not scale with threads (SURVEY.md §6 measured 0.42-0.50 s per call on 8 threads;
    # 128 threads took 0.8-1.4 s on the GPU boxes): the baseline is the CPU's BEST, so sweep the thread count - one warm-up
    # call and one timed call each - and time `calls` calls at the fastest setting.
    all_threads = torch.get_num_threads()
    sweep = {}
    for n_thr in sorted({t for t in (8, 16, 32, all_threads) if 1 <= t <= max(all_threads, 8)}):
        torch.set_num_threads(n_thr)
        one(h, pt, c, s)
        t0 = time.perf_counter()
        one(h, pt, c, s)
        sweep[n_thr] = time.perf_counter() - t0
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    calls = max(1, min(calls, int(6.0 / sweep[threads]) or 1))          # ~6 s at the best setting
    t0 = time.perf_counter()
    for _ in range(calls):
        lo = one(h, pt, c, s)
    dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    cpu = {"value": (L - lo) * calls / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
           "thread_sweep_ms_per_call": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
           "sample": f"{calls} merge calls of the CPU oracle on the same [1, {L}, {hidden.shape[2]}] bf16 sample at the fastest "
                     f"of {sorted(sweep)} torch threads = {threads} ({dt / calls * 1e3:.0f} ms per call, torch {torch.__version__} "
                     f"CPU, {os.cpu_count()} logical cores)"}
    n = 5
    for _ in range(2):
        one(hidden, ptype, cos, sin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        lg = one(hidden, ptype, cos, sin)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / n * 1e3
    eager = {"value": (L - lg) / (eager_ms * 1e-3), "unit": "tokens/s", "ms_per_call": eager_ms, "tokens_out": lg,
             "kind": "torch port of the reference path, eager on the MI355X (same sample, same tie rule)",
             "speedup_of_this_build": eager_ms / hip_ms}
    return cpu, eager


if __name__ == "__main__":
    main()
