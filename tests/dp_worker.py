#!/usr/bin/env python
"""Stand-in for bench.py on a machine without GPUs: the SAME multi-rank skeleton (dp.launch_ranks ->
torch.distributed.run -> dp.init -> dp.broadcast_config -> dp.timed_steps -> dp.gather_records ->
dp.aggregate, rank 0 prints one JSON line) over gloo, with the CPU oracle standing in for the HIP step.

    python tests/dp_worker.py --gpus 2        # self-launches 2 ranks, like `python bench.py --gpus 2`
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from framefusion_amd import dp                       # noqa: E402

dp.load_hooks("tests.dp_faults")                     # (the fault / delay injection of the tests; chosen by their environment)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seed", type=int, default=100)
    ap.add_argument("--fake-sysfs", default=None, help="bind every rank like bench.py does, against this sysfs tree: rank r sits on PCI function 0000:<r+1>0:00")
    a = ap.parse_args()
    dp.launch_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])      # does not return when it launches
    world, rank, local = dp.env_world()
    if a.fake_sysfs:
        # what bench.py does before anything pinned exists (dp.bind_to_gpu_numa), with made-up PCI addresses
        devs = [f"0000:{r + 1:x}0:00" for r in range(world)]
        dp.bind_to_gpu_numa(devs[local], local, devs, a.fake_sysfs)
    dist = dp.init("gloo")
    dev = torch.device("cpu")
    # every rank but 0 holds a WRONG local seed: the broadcast must overwrite it
    cfg = dp.broadcast_config(dist, dict(seed=a.seed + 1000 * rank, frames=6, patches=8, dim=32, steps=a.steps,
                                         warmup=a.warmup, p_change=0.3), dev)
    from oracle import ff_oracle as orc
    from framefusion_amd.synth import video_tokens
    h, pt = video_tokens(cfg["frames"], cfg["patches"], cfg["dim"], p_change=cfg["p_change"],
                         seed=dp.sample_seed(cfg["seed"], rank), pre=1, post=1, grid=0.125)
    L = h.shape[1]

    def step():
        f = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        f.prepare(pt, cfg["patches"], 1, L - 2, L - 2, L)
        return f.forward(h, torch.arange(L)[None], None)[0]

    timed = dp.timed_steps(dist, step, cfg["steps"], cfg["warmup"], dev)
    t_max, mine, out = timed
    spreads = [None] * (dist.get_world_size() if dist else 1)
    if dist is not None:
        dist.all_gather_object(spreads, timed.step_us)
    else:
        spreads = [timed.step_us]
    recs = dp.gather_records(dist, (L, out.shape[1], mine * 1e3 / cfg["steps"], cfg["seed"]), dev)
    t_all, units = dp.aggregate(dist, mine, float((L - out.shape[1]) * cfg["steps"]), dev)
    ids = dp.gather_identities(dist, dev)
    # the kept-token index lists (every rank keeps a different number): padded all_gather, true lengths back
    kept = torch.arange(L)[torch.arange(L) % (rank + 2) == 0]
    kept_all = [k.tolist() for k in dp.gather_kept_indices(dist, kept, L, dev)]
    if rank == 0:
        print(json.dumps(dict(n_gpus=world, ranks=dist.get_world_size() if dist else 1, records=recs, t_max=t_max,
                              t_all=t_all, units=units, seed=cfg["seed"], steps=cfg["steps"], identities=ids, kept=kept_all, step_us=spreads,
                              L=L,
                              # the per-rank block of bench.py's line (what the driver's scaling parser reads), same keys
                              per_rank=[{"rank": r, "gpu": r, "tokens_in": int(rec[0]), "tokens_out": int(rec[1]), "ms_per_step": rec[2],
                                         "seed": dp.sample_seed(cfg["seed"], r), "hostname": w.get("hostname"), "pid": w.get("pid"),
                                         "pci_bus_id": w.get("pci_bus_id"), "numa_node": w.get("numa_node"), "cpus": w.get("cpus"),
                                         "step_us": su, "kept_indices": {"n": len(kx)}}
                                        for r, (rec, w, su, kx) in enumerate(zip(recs, ids, spreads, kept_all))])))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
