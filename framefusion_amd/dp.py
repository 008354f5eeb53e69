"""Data-parallel sharding of independent video samples.

One FrameFusion instance handles one sample (`bsz == 1`, reference main.py:203) and samples never
interact, so the path shards trivially: sample i -> rank i mod world, one process per GPU, no
collective on the data path (the reference's own 8-way run is `accelerate launch --num_processes=8`,
README.md:146: independent replicas).  RCCL (torch.distributed backend "nccl"; "gloo" in the CPU
tests) carries three small things, all off the timed path:

  * `broadcast_config`  - rank 0's workload description (seed, shape, step counts) to every rank;
  * `gather_records`    - all_gather of one fixed-size float64 record per rank (L_in, L_out, ms, ...);
  * `aggregate`         - all_reduce MAX of the elapsed time / SUM of the units processed.

`launch_ranks` turns `python bench.py --gpus N` into N ranks (one per GPU) when the process was not
started by torch.distributed.run already.
"""
from __future__ import annotations

import os
import socket
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(world_size, rank, local_rank) as torch.distributed.run exports them."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def under_launcher() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks: int, script: str, argv: Sequence[str]) -> None:
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N script argv`
    (rendezvous on 127.0.0.1, a free port).  Does not return.  No-op when already under a launcher or
    when one rank is asked for."""
    if n_ranks <= 1 or under_launcher():
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def init(backend: str = "nccl", device: Optional[torch.device] = None, force: bool = False):
    """Join the process group (no-op for a single process unless `force`: a one-rank group, which still runs every
    collective below through the backend - the way to exercise RCCL itself on a 1-GPU box).  Returns the dist module
    or None."""
    world, rank, local = env_world()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        kwargs = {}
        if world <= 1:
            kwargs.update(world_size=1, rank=0)
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device                 # binds the communicator to this rank's GPU up front
        dist.init_process_group(backend, **kwargs)
    return dist


def shard(n_samples: int, world: int, rank: int) -> List[int]:
    """Indices of the samples this rank owns (round-robin)."""
    return list(range(rank, n_samples, world))


def sample_seed(base_seed: int, sample_index: int) -> int:
    return base_seed + sample_index


def _comm_device(dist, device) -> torch.device:
    """Collectives run on the GPU over RCCL and on the host over gloo."""
    if dist is not None and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def broadcast_config(dist, config: Dict[str, float], device, src: int = 0) -> Dict[str, float]:
    """Rank `src`'s workload description to every rank (one RCCL broadcast of len(config) float64).
    Every rank passes a dict with the SAME keys (its own parse of the command line); the values of
    rank `src` win, so a seed or size can never differ between the ranks of one job."""
    if dist is None:
        return dict(config)
    keys = sorted(config)
    t = torch.tensor([float(config[k]) for k in keys], dtype=torch.float64, device=_comm_device(dist, device))
    dist.broadcast(t, src=src)
    vals = t.cpu().tolist()
    return {k: (int(v) if float(v).is_integer() and isinstance(config[k], int) else v) for k, v in zip(keys, vals)}


def gather_records(dist, record: Sequence[float], device) -> List[List[float]]:
    """One fixed-size float64 record per rank, on every rank (RCCL all_gather)."""
    mine = torch.tensor([float(x) for x in record], dtype=torch.float64, device=_comm_device(dist, device))
    if dist is None:
        return [mine.tolist()]
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [p.cpu().tolist() for p in parts]


def aggregate(dist, elapsed_s: float, units: float, device) -> Tuple[float, float]:
    """Whole-job numbers: the slowest rank's time and the units all ranks processed."""
    if dist is None:
        return elapsed_s, units
    cd = _comm_device(dist, device)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=cd)
    u = torch.tensor([units], dtype=torch.float64, device=cd)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t), float(u)


def gather_lengths(dist, l_in: int, l_out: int, device) -> List[Tuple[int, int]]:
    """Per-rank (L_in, L_out) of the last reduction, on every rank."""
    return [(int(a), int(b)) for a, b in gather_records(dist, (l_in, l_out), device)]


def barrier(dist):
    """Every rank's device work done, then every rank here (then nothing left in flight)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


def timed_steps(dist, step, steps: int, warmup: int, device) -> Tuple[float, float, object]:
    """The bench contract: `warmup` untimed steps, then EXACTLY `steps` steps between
    barrier + synchronize on both sides.  Returns (max elapsed over ranks, this rank's elapsed, last result)."""
    import time
    out = None
    for _ in range(warmup):
        out = step()
    barrier(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier(dist)
    mine = time.perf_counter() - t0
    t_max, _ = aggregate(dist, mine, 0.0, device)
    return t_max, mine, out
