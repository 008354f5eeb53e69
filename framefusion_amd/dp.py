"""Data-parallel sharding of independent video samples.

One FrameFusion instance handles one sample (`bsz == 1`, reference main.py:203) and samples never
interact, so the path shards trivially: sample i -> rank i mod world, one process per GPU, no
collective on the data path.  RCCL (torch.distributed backend "nccl"; "gloo" in the CPU tests) only
carries the timing barrier and a few scalars: max elapsed time, summed token counts, and - for
callers that want them - the per-rank output lengths.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(world_size, rank, local_rank) as torch.distributed.run exports them."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str = "nccl", device: Optional[torch.device] = None):
    """Join the process group (no-op for a single process). Returns the dist module or None."""
    world, rank, local = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        if backend == "nccl" and device is not None:
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return dist


def shard(n_samples: int, world: int, rank: int) -> List[int]:
    """Indices of the samples this rank owns (round-robin)."""
    return list(range(rank, n_samples, world))


def sample_seed(base_seed: int, sample_index: int) -> int:
    return base_seed + sample_index


def aggregate(dist, elapsed_s: float, units: float, device) -> Tuple[float, float]:
    """Whole-job numbers: the slowest rank's time and the units all ranks processed."""
    if dist is None:
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t), float(u)


def gather_lengths(dist, l_in: int, l_out: int, device) -> List[Tuple[int, int]]:
    """Per-rank (L_in, L_out) of the last reduction, on every rank."""
    mine = torch.tensor([l_in, l_out], dtype=torch.int64, device=device)
    if dist is None:
        return [(l_in, l_out)]
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [(int(p[0]), int(p[1])) for p in parts]


def barrier(dist):
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
