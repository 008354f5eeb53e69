"""Data-parallel sharding of independent video samples.

One FrameFusion instance handles one sample (`bsz == 1`, reference main.py:203) and samples never
interact, so the path shards trivially: sample i -> rank i mod world, one process per GPU, no
collective on the data path (the reference's own 8-way run is `accelerate launch --num_processes=8`,
README.md:146: independent replicas).  RCCL (torch.distributed backend "nccl"; "gloo" in the CPU
tests) carries three small things, all off the timed path:

  * `broadcast_config`  - rank 0's workload description (seed, shape, step counts) to every rank;
  * `gather_records`    - all_gather of one fixed-size float64 record per rank (L_in, L_out, ms, ...);
  * `aggregate`         - all_reduce MAX of the elapsed time / SUM of the units processed;
  * `gather_identities` - all_gather_object of (hostname, pid, PCI address of the GPU, IPC mode) per rank;
  * `gather_kept_indices` - all_gather of every rank's kept-token index list, padded to a fixed int32 capacity.

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


IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"      # "0": dmabuf IPC (what this image's host driver supports); unset: the runtime's default


def ipc_mode() -> str:
    """The IPC mode this process runs under, as it is reported in the bench line."""
    v = os.environ.get(IPC_VAR)
    return f"{IPC_VAR}={v}" if v is not None else f"{IPC_VAR} unset"


def attempt() -> int:
    """0: the first try of this job; 1: the ranks re-executed themselves with the other IPC mode (see `init`)."""
    return int(os.environ.get("FF_DP_ATTEMPT", "0"))


def _other_ipc_env(env: Dict[str, str]) -> Dict[str, str]:
    env = dict(env)
    if env.get(IPC_VAR) == "0":
        del env[IPC_VAR]
    else:
        env[IPC_VAR] = "0"
    return env


def launch_ranks(n_ranks: int, script: str, argv: Sequence[str]) -> None:
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N script argv`
    (rendezvous on 127.0.0.1, a free port).  Does not return.  No-op when already under a launcher or
    when one rank is asked for.  The IPC mode is the environment's (the GPU boxes export
    HSA_ENABLE_IPC_MODE_LEGACY=0); only when the variable is absent is the documented value filled in - and
    `init` falls back to the other mode if the first collective fails."""
    if n_ranks <= 1 or under_launcher():
        return
    env = dict(os.environ)
    env.setdefault(IPC_VAR, "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def _join(dist, backend: str, device, world: int, rank: int, force: bool):
    """init_process_group; `_probe` then runs ONE small collective through the backend (RCCL sets its transports up
    lazily: an IPC problem shows at the first collective, not at init)."""
    import datetime
    kwargs = {}
    if world <= 1:
        kwargs.update(world_size=1, rank=0)
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device                 # binds the communicator to this rank's GPU up front
    n = attempt()
    if n:
        # The ranks re-executed themselves: the rendezvous store may still hold the first attempt's keys (under
        # torch.distributed.run the AGENT hosts it and outlives the workers), so this attempt talks to it under its own
        # prefix; when rank 0 hosts the store, the re-executed rank 0 serves a fresh one on the same port.
        agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), max(world, 1),
                             is_master=(rank == 0 and not agent_store), timeout=datetime.timedelta(seconds=300),
                             wait_for_workers=False)
        kwargs.update(store=dist.PrefixStore(f"ff_attempt{n}", base), world_size=max(world, 1), rank=rank if world > 1 else 0)
    dist.init_process_group(backend, **kwargs)


def _probe(dist, backend: str, device, world: int):
    if os.environ.get("FF_DP_FAIL_FIRST_ATTEMPT") == "1" and attempt() == 0:   # (tests: exercise the fallback without a broken box)
        raise RuntimeError("injected failure of the first collective (FF_DP_FAIL_FIRST_ATTEMPT=1)")
    probe = torch.ones(1, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(probe)
    if backend == "nccl":
        torch.cuda.synchronize(device)
    if int(probe.item()) != max(world, 1):
        raise RuntimeError(f"first all_reduce over {backend} returned {probe.item()} for {world} ranks")


def init(backend: str = "nccl", device: Optional[torch.device] = None, force: bool = False):
    """Join the process group (no-op for a single process unless `force`: a one-rank group, which still runs every
    collective below through the backend - the way to exercise RCCL itself on a 1-GPU box).  Returns the dist module
    or None.

    IPC-mode fallback: HSA_ENABLE_IPC_MODE_LEGACY is read once, when the HSA runtime starts, so a process cannot change
    its mind.  If joining or the first collective fails on the first attempt, EVERY rank (they all see the failure of a
    collective) re-executes itself - same pid, so an outer torch.distributed.run keeps supervising it - with the variable
    flipped ("0" <-> unset) and FF_DP_ATTEMPT=1; a second failure is final.  `ipc_mode()` / `attempt()` say what ran."""
    world, rank, local = env_world()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if dist.is_initialized():
        return dist
    try:
        _join(dist, backend, device, world, rank, force)
        _probe(dist, backend, device, world)
    except Exception as e:
        # (only a script can be re-executed: `python -c ...` / an interactive session just gets the error)
        if attempt() == 0 and os.environ.get("FF_DP_NO_RETRY") != "1" and sys.argv and os.path.isfile(sys.argv[0]):
            env = _other_ipc_env(os.environ)
            env["FF_DP_ATTEMPT"] = "1"
            env["FF_DP_FIRST_ERROR"] = f"{type(e).__name__}: {str(e)[:300]}"
            print(f"[framefusion_amd.dp] rank {rank}: joining over {backend} failed with {ipc_mode()} "
                  f"({type(e).__name__}: {str(e)[:200]}); re-executing with the other IPC mode", file=sys.stderr)
            sys.stdout.flush()
            sys.stderr.flush()
            os.execve(sys.executable, [sys.executable] + sys.argv, env)
        raise
    return dist


def identity(device: Optional[torch.device]) -> Dict[str, object]:
    """Who this rank is: host, pid and the PCI address of its GPU - all_gathered into the report so that an N-GPU line
    proves N distinct devices."""
    rec = {"hostname": socket.gethostname(), "pid": os.getpid(), "ipc_mode": ipc_mode(), "attempt": attempt()}
    if device is not None and device.type == "cuda" and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(device)
        rec["pci_bus_id"] = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
        rec["device_name"] = p.name
        uuid = getattr(p, "uuid", None)
        if uuid is not None:
            rec["uuid"] = str(uuid)
    return rec


def gather_identities(dist, device) -> List[Dict[str, object]]:
    """`identity()` of every rank, on every rank (all_gather_object: pickled through the backend)."""
    mine = identity(device)
    if dist is None:
        return [mine]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return out


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


def gather_kept_indices(dist, kept: torch.Tensor, capacity: int, device) -> List[torch.Tensor]:
    """Every rank's kept-token indices (all_gather of the index lists, padded to `capacity` int32 with -1:
    <= 147 KB per rank at 64 x 576), on every rank, as CPU int32 tensors of their true lengths.  One fixed-size all_gather:
    a ring over xGMI moves (world - 1) x 147 KB per link, microseconds next to a prefill, and off the timed path."""
    cd = _comm_device(dist, device)
    mine = torch.full((capacity,), -1, dtype=torch.int32, device=cd)
    n = int(kept.numel())
    if n > capacity:
        raise ValueError(f"{n} kept indices for a capacity of {capacity}")
    mine[:n] = kept.to(device=cd, dtype=torch.int32)
    if dist is None:
        return [mine[:n].cpu()]
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    out = []
    for p in parts:
        p = p.cpu()
        out.append(p[: int((p >= 0).sum())])
    return out


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
