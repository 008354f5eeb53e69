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

`bind_to_gpu_numa` pins a rank to the cores of its GPU's NUMA node (sysfs) before anything pinned is allocated.

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


class _StdoutToStderr:
    """While RCCL sets itself up: file descriptor 1 points at stderr, and C stdio is flushed before it is put back.  RCCL
    prints a five-line banner ("RCCL version : ...") with printf on rank 0 when its first communicator is created; stdout of
    a bench run is ONE JSON line, and a buffered banner would otherwise surface behind it when the process exits."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _timeout_s() -> float:
    return float(os.environ.get("FF_DP_INIT_TIMEOUT", "300"))


def _join(dist, backend: str, device, world: int, rank: int, force: bool):
    """init_process_group; `_probe` then runs ONE small collective through the backend (RCCL sets its transports up
    lazily: an IPC problem shows at the first collective, not at init)."""
    import datetime
    kwargs = {"timeout": datetime.timedelta(seconds=_timeout_s())}
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


class Hooks:
    """The two places where a test may interfere with a job (a rank whose first collective fails, a rank that dawdles inside a
    barrier).  The product never installs anything: `load_hooks("tests.dp_faults")` - what `bench.py --dp-hooks` and
    tests/dp_worker.py call - imports a module and lets it replace `dp.hooks`."""

    def first_collective(self, rank: int, attempt: int) -> None:
        pass

    def in_barrier(self, rank: int, world: int) -> None:
        pass


hooks = Hooks()


def load_hooks(module_name: str) -> None:
    import importlib
    importlib.import_module(module_name).install(sys.modules[__name__])


def _probe(dist, backend: str, device, world: int):
    """The first collective, with a deadline: RCCL transport problems usually HANG rather than raise, so the all_reduce runs
    on a helper thread and a rank that is not through after FF_DP_INIT_TIMEOUT seconds counts as failed (the re-exec that
    follows replaces the process, hung thread included)."""
    import threading
    hooks.first_collective(int(os.environ.get("RANK", "0")), attempt())
    box: Dict[str, object] = {}

    def run():
        try:
            if backend == "nccl":
                torch.cuda.set_device(device)
            probe = torch.ones(1, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if backend == "nccl":
                torch.cuda.synchronize(device)
            box["value"] = float(probe.item())
        except Exception as e:                                  # noqa: BLE001 (reported by the caller's thread)
            box["error"] = e
    import time
    th = threading.Thread(target=run, daemon=True)
    th.start()
    deadline = time.monotonic() + _timeout_s()
    while th.is_alive() and time.monotonic() < deadline:
        th.join(0.25)
        if th.is_alive() and world > 1:
            bad = _failed_peer(dist, world)           # a peer that already gave up will never enter the collective
            if bad is not None:
                raise RuntimeError(f"rank {bad} failed to join: not waiting for the first all_reduce over {backend}")
    if th.is_alive():
        raise TimeoutError(f"the first all_reduce over {backend} did not complete within {_timeout_s():.0f} s")
    if "error" in box:
        raise box["error"]
    if int(box["value"]) != max(world, 1):
        raise RuntimeError(f"first all_reduce over {backend} returned {box['value']} for {world} ranks")


def _failed_peer(dist, world: int) -> Optional[int]:
    """The first rank whose verdict in the store (see `_agree`) is a failure, or None."""
    try:
        store = dist.distributed_c10d._get_default_store()
        for r in range(world):
            if store.check([f"ff_probe/{r}"]) and store.get(f"ff_probe/{r}") == b"0":
                return r
    except Exception:                                           # noqa: BLE001
        pass
    return None


def _agree(dist, ok: bool, world: int, rank: int) -> bool:
    """Every rank's verdict on the join, exchanged through the rendezvous STORE (TCP: independent of the backend under test):
    True only if all ranks got through.  A failure on SOME ranks would otherwise leave the others blocked in their next
    collective while the failed ones re-execute."""
    import datetime
    if world <= 1:
        return ok
    try:
        store = dist.distributed_c10d._get_default_store()
        store.set(f"ff_probe/{rank}", "1" if ok else "0")
        votes = []
        for r in range(world):
            store.wait([f"ff_probe/{r}"], datetime.timedelta(seconds=_timeout_s() + 30))
            votes.append(store.get(f"ff_probe/{r}") == b"1")
        return all(votes)
    except Exception:                                           # noqa: BLE001 (no store / a rank never voted: not agreed)
        return False


def init(backend: str = "nccl", device: Optional[torch.device] = None, force: bool = False):
    """Join the process group (no-op for a single process unless `force`: a one-rank group, which still runs every
    collective below through the backend - the way to exercise RCCL itself on a 1-GPU box).  Returns the dist module
    or None.

    IPC-mode fallback: HSA_ENABLE_IPC_MODE_LEGACY is read once, when the HSA runtime starts, so a process cannot change
    its mind.  If joining or the first collective fails - raises, or does not complete within FF_DP_INIT_TIMEOUT (300) s -
    on ANY rank (the ranks exchange their verdicts through the rendezvous store), EVERY rank re-executes itself - same
    pid, so an outer torch.distributed.run keeps supervising it - with the variable flipped ("0" <-> unset) and
    FF_DP_ATTEMPT=1; a second failure is final and leaves no process group behind.  `ipc_mode()` / `attempt()` say what ran."""
    world, rank, local = env_world()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if dist.is_initialized():
        return dist
    if (backend == "nccl" and attempt() == 0 and IPC_VAR not in os.environ and
            not (torch.cuda.is_available() and torch.cuda.is_initialized())):
        # the HSA runtime has not started in this process yet: the documented value still takes effect (the GPU boxes export
        # it; `launch_ranks` fills it in for its children; here for a launcher that did neither).  Never on the second
        # attempt: "unset" is then the mode being tried.
        os.environ[IPC_VAR] = "0"
    err: Optional[BaseException] = None
    joined = False
    try:
        with _StdoutToStderr():                  # (RCCL's banner, gloo's connection notes: not on the report's stdout)
            _join(dist, backend, device, world, rank, force)
            joined = True
            _probe(dist, backend, device, world)
    except Exception as e:                                      # noqa: BLE001
        err = e
    everyone = _agree(dist, err is None, world, rank) if joined else False
    if err is not None or not everyone:
        if err is None:
            err = RuntimeError("another rank failed to join (see its message)")
        if joined and not isinstance(err, TimeoutError) and "not waiting for the first all_reduce" not in str(err):
            # (a probe thread still inside a collective would make destroy block: the re-exec below replaces the process)
            try:                                                 # a half-initialised group must not look healthy to a later init()
                dist.destroy_process_group()
            except Exception:                                    # noqa: BLE001
                pass
        # (only a script can be re-executed: `python -c ...` / an interactive session just gets the error)
        if attempt() == 0 and os.environ.get("FF_DP_NO_RETRY") != "1" and sys.argv and os.path.isfile(sys.argv[0]):
            env = _other_ipc_env(os.environ)
            env["FF_DP_ATTEMPT"] = "1"
            env["FF_DP_FIRST_ERROR"] = f"{type(err).__name__}: {str(err)[:300]}"
            print(f"[framefusion_amd.dp] rank {rank}: joining over {backend} failed with {ipc_mode()} "
                  f"({type(err).__name__}: {str(err)[:200]}); re-executing with the other IPC mode", file=sys.stderr)
            sys.stdout.flush()
            sys.stderr.flush()
            os.execve(sys.executable, [sys.executable] + sys.argv, env)
        raise err
    return dist


def pci_address(device) -> Optional[str]:
    """dddd:bb:dd of a CUDA device (what `identity` reports and sysfs names the device by, function .0)."""
    if device is None or torch.device(device).type != "cuda" or not torch.cuda.is_available():
        return None
    p = torch.cuda.get_device_properties(device)
    return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_of_pci(pci: Optional[str], sysfs: str = "/sys") -> Tuple[Optional[int], List[int]]:
    """(NUMA node, its CPUs) of the PCI device `pci` ("dddd:bb:dd", function .0) as sysfs reports them:
    /sys/bus/pci/devices/<bdf>/numa_node and /sys/devices/system/node/node<N>/cpulist.  (None, []) when the platform does
    not say (no such device, numa_node = -1: a single-node or virtualised host)."""
    if not pci:
        return None, []
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci + ".0", "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, []
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            return node, parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, []


_BOUND: Dict[str, object] = {"numa_node": None, "cpus": None}


def bind_to_gpu_numa(device, local_rank: int = 0, local_devices: Optional[Sequence] = None, sysfs: str = "/sys") -> Dict[str, object]:
    """Pin this process (every thread it starts later: the Python host AND the C poll loop of ff_ctx_merge_finish) to the
    cores of its GPU's NUMA node - BEFORE the pinned result block is allocated, so that first-touch places the block on the
    node the GPU writes it from and the poller reads it locally.  Ranks whose GPUs share a node split the node's cores
    evenly (rank order), so eight spinning pollers and eight interpreters never compete for a core on a 2-socket host.
    `local_devices`: the devices of ALL ranks of this host in local-rank order (default: cuda:0 .. cuda:N-1 for N =
    LOCAL_WORLD_SIZE / WORLD_SIZE).  Returns {"numa_node", "cpus"} (None / None: the platform did not say, or
    FF_DP_NO_BIND=1: nothing was changed) - reported per rank in the bench line."""
    if os.environ.get("FF_DP_NO_BIND") == "1" or not hasattr(os, "sched_setaffinity"):
        return dict(_BOUND)
    node, cpus = numa_of_pci(pci_address(device) if not isinstance(device, str) else device, sysfs)
    if node is None or not cpus:
        return dict(_BOUND)
    if local_devices is None:
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        local_devices = [torch.device("cuda", r % n_dev) for r in range(n_local)] if n_dev else []
    sharing = [r for r, d in enumerate(local_devices)
               if numa_of_pci(pci_address(d) if not isinstance(d, str) else d, sysfs)[0] == node]
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0))) or sorted(cpus)
    if local_rank in sharing and len(sharing) > 1 and len(allowed) >= len(sharing):       # (at least one core each)
        at, per = sharing.index(local_rank), len(allowed) // len(sharing)
        allowed = allowed[at * per:(at + 1) * per]
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return dict(_BOUND)
    _BOUND.update(numa_node=node, cpus=compact_cpulist(allowed))
    return dict(_BOUND)


def compact_cpulist(cpus: Sequence[int]) -> str:
    """[0, 1, 2, 3, 8] -> "0-3,8"."""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def identity(device: Optional[torch.device]) -> Dict[str, object]:
    """Who this rank is: host, pid, the PCI address of its GPU and where `bind_to_gpu_numa` put it - all_gathered into the
    report so that an N-GPU line proves N distinct devices."""
    rec = {"hostname": socket.gethostname(), "pid": os.getpid(), "ipc_mode": ipc_mode(), "attempt": attempt(),
           "numa_node": _BOUND["numa_node"], "cpus": _BOUND["cpus"]}
    if device is not None and device.type == "cuda" and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(device)
        rec["pci_bus_id"] = pci_address(device)
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
        hooks.in_barrier(dist.get_rank(), dist.get_world_size())
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


class Timed(tuple):
    """(t_max, mine, last result) of `timed_steps`, plus `.step_us`: this rank's per-step times (min / median / p90 / max)."""
    step_us: Dict[str, float] = {}


def timed_steps(dist, step, steps: int, warmup: int, device) -> "Timed":
    """The bench contract: `warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides.
    A rank's clock runs from the opening barrier to ITS OWN synchronize behind the last step; the closing barrier comes
    after that and is in nobody's time (an RCCL all-reduce of 50-100 us would be 2-4 % of a 2.7 ms region and is not the
    path's).  Returns Timed(max over ranks of those clocks (all_reduce MAX), this rank's clock, last result)."""
    import time
    out = None
    for _ in range(warmup):
        out = step()
    barrier(dist)
    stamps = [0.0] * (steps + 1)
    t0 = stamps[0] = time.perf_counter()
    for i in range(steps):
        out = step()
        stamps[i + 1] = time.perf_counter()
    drain = getattr(step, "drain", None)          # a pipelined step leaves its last call in flight: collected inside the clock
    if drain is not None:
        out = drain()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    barrier(dist)
    t_max, _ = aggregate(dist, mine, 0.0, device)
    res = Timed((t_max, mine, out))
    # host stamps behind every step: the step returns when its result block has arrived, so in steady state consecutive
    # stamps are one device step apart (the last step's tail is in `mine`, not here)
    us = sorted((b - a) * 1e6 for a, b in zip(stamps, stamps[1:]))
    res.step_us = ({"min": us[0], "median": us[len(us) // 2], "p90": us[int(0.9 * (len(us) - 1))], "max": us[-1]} if us else {})
    return res
