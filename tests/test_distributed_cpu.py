"""CPU, world_size 2 over gloo: the multi-rank skeleton of bench.py, driven through the PRODUCT's code path
(framefusion_amd/dp.py): `python tests/dp_worker.py --gpus 2` launches its own two ranks exactly as
`python bench.py --gpus 2` does (dp.launch_ranks -> torch.distributed.run on 127.0.0.1), rank 0's config
is broadcast, every rank times its own independent sample, records are all_gathered and the whole-job
numbers are max-time / summed-units.  The data path itself has no collective."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(*argv, env_extra=None, env_drop=(), direct=False):
    """`direct`: start the two ranks with an external torch.distributed.run (what the driver does), so that dp.launch_ranks -
    which would fill HSA_ENABLE_IPC_MODE_LEGACY in - is not involved."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(env_drop)}
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), *argv]
    if direct:
        from framefusion_amd import dp
        n = argv[argv.index("--gpus") + 1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(dp.free_port())] + cmd[1:]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout           # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_self_launched_two_ranks_over_gloo():
    out = run_worker("--gpus", "2", "--steps", "3", "--warmup", "1", "--seed", "100")
    assert out["n_gpus"] == 2 and out["ranks"] == 2
    recs = out["records"]
    assert len(recs) == 2
    # rank 1 started with seed 1100 locally; the broadcast made rank 0's seed the job's seed
    assert [r[3] for r in recs] == [100.0, 100.0] and out["seed"] == 100
    # independent samples (seed + rank): both reduce something, and differently
    assert all(r[0] > r[1] for r in recs) and recs[0][1] != recs[1][1]
    assert out["units"] == sum((r[0] - r[1]) * out["steps"] for r in recs)
    assert out["t_max"] == out["t_all"] >= max(r[2] for r in recs) * out["steps"] / 1e3 * 0.5
    # all_gather of the kept-index lists (padded to L, true lengths restored): rank r kept every (r + 2)-th position
    assert out["kept"] == [list(range(0, out["L"], 2)), list(range(0, out["L"], 3))]


@pytest.mark.timeout(300)
@pytest.mark.timeout(300)
def test_ranks_fall_back_to_the_other_ipc_mode_when_the_first_collective_fails():
    """dp.init: a failure of the join / first collective on the first attempt makes every rank re-execute itself with
    HSA_ENABLE_IPC_MODE_LEGACY flipped and join again under a fresh store prefix - here over gloo, under the same
    torch.distributed.run the product starts, with the failure injected."""
    out = run_worker("--gpus", "2", "--steps", "2", "--warmup", "1", "--seed", "100",
                     env_extra={"FF_DP_FAIL_FIRST_ATTEMPT": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and len(out["records"]) == 2
    ids = out["identities"]
    assert [i["attempt"] for i in ids] == [1, 1] and all(i["ipc_mode"].endswith("unset") for i in ids)
    assert len({i["pid"] for i in ids}) == 2
    # ... and vice versa: a job that starts without the variable retries with it
    out = run_worker("--gpus", "2", "--steps", "2", "--warmup", "1", env_extra={"FF_DP_FAIL_FIRST_ATTEMPT": "1"},
                     env_drop=("HSA_ENABLE_IPC_MODE_LEGACY",), direct=True)
    assert [i["attempt"] for i in out["identities"]] == [1, 1]
    assert all(i["ipc_mode"] == "HSA_ENABLE_IPC_MODE_LEGACY=0" for i in out["identities"])
    # no failure: first attempt, the environment's mode untouched
    out = run_worker("--gpus", "2", "--steps", "2", "--warmup", "1", env_extra={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert [i["attempt"] for i in out["identities"]] == [0, 0]


@pytest.mark.timeout(300)
def test_the_closing_barrier_is_in_nobodys_time_and_one_failing_rank_takes_all_with_it():
    """(a) dp.timed_steps stops a rank's clock at its own synchronize, BEFORE the closing barrier: a rank that dawdles 400 ms
    inside that barrier (FF_DP_SLOW_BARRIER_MS) must not show in t_max.  (b) The join verdicts travel through the store: when
    the first collective fails on rank 1 ONLY, rank 0 re-executes as well instead of blocking in its next collective."""
    out = run_worker("--gpus", "2", "--steps", "3", "--warmup", "1", env_extra={"FF_DP_SLOW_BARRIER_MS": "400"})
    # (both barriers of the timed region are slowed: 0.8 s of wall time that no rank's clock may contain)
    assert out["t_max"] < 0.35, out["t_max"]
    assert all(set(s) == {"min", "median", "p90", "max"} for s in out["step_us"])
    out = run_worker("--gpus", "2", "--steps", "2", "--warmup", "1",
                     env_extra={"FF_DP_FAIL_FIRST_ATTEMPT": "1", "FF_DP_FAIL_RANK": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert [i["attempt"] for i in out["identities"]] == [1, 1]


def test_numa_binding_reads_sysfs(tmp_path):
    """dp.numa_of_pci / bind_to_gpu_numa against a fake sysfs tree: node and cpulist of the GPU's PCI function, ranks that
    share a node split its cores, a platform that does not say (-1) changes nothing."""
    from framefusion_amd import dp
    sysfs = tmp_path
    for bdf, node in (("0000:05:00", 0), ("0000:06:00", 0), ("0000:85:00", 1), ("0000:99:00", -1)):
        d = sysfs / "bus/pci/devices" / (bdf + ".0")
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    mine = sorted(os.sched_getaffinity(0))
    assert len(mine) >= 4
    for node, cpus in ((0, mine[: len(mine) // 2]), (1, mine[len(mine) // 2:])):
        d = sysfs / "devices/system/node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(dp.compact_cpulist(cpus) + "\n")
    assert dp.numa_of_pci("0000:05:00", str(sysfs)) == (0, mine[: len(mine) // 2])
    assert dp.numa_of_pci("0000:99:00", str(sysfs)) == (None, []) and dp.numa_of_pci("0000:77:00", str(sysfs)) == (None, [])
    assert dp.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and dp.compact_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    devs = ["0000:05:00", "0000:06:00", "0000:85:00"]
    try:
        got = dp.bind_to_gpu_numa("0000:85:00", 2, devs, str(sysfs))            # alone on node 1: the whole node
        assert got["numa_node"] == 1 and sorted(os.sched_getaffinity(0)) == mine[len(mine) // 2:]
        assert dp.identity(None)["numa_node"] == 1 and dp.identity(None)["cpus"] == dp.compact_cpulist(mine[len(mine) // 2:])
        os.sched_setaffinity(0, mine)
        got = dp.bind_to_gpu_numa("0000:06:00", 1, devs, str(sysfs))            # shares node 0 with rank 0: the second half of it
        half = mine[: len(mine) // 2]
        assert got["numa_node"] == 0 and sorted(os.sched_getaffinity(0)) == half[len(half) // 2:]
        os.sched_setaffinity(0, mine)
        before = dict(dp._BOUND)
        assert dp.bind_to_gpu_numa("0000:99:00", 0, devs, str(sysfs)) == before and sorted(os.sched_getaffinity(0)) == mine
    finally:
        os.sched_setaffinity(0, mine)
        dp._BOUND.update(numa_node=None, cpus=None)


@pytest.mark.timeout(600)
def test_eight_ranks_bind_to_eight_disjoint_cpu_sets_and_report_the_scaling_schema(tmp_path):
    """The N = 8 job the driver will start on an 8-GPU node, as far as a box without GPUs can run it: 8 gloo ranks through
    dp.launch_ranks / dp.init / dp.timed_steps, every rank bound (dp.bind_to_gpu_numa) against a fake 2-socket sysfs tree -
    GPUs 0-3 on node 0, 4-7 on node 1, each node holding half of this machine's cores - before anything else happens.  Eight
    disjoint cpu sets of equal size; the all_gathered per-rank block carries the keys bench.py's line has."""
    from framefusion_amd import dp
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 8:
        pytest.skip("needs 8 cores for 8 disjoint sets")
    mine = mine[: len(mine) // 8 * 8]
    half = len(mine) // 2
    for r in range(8):
        d = tmp_path / "bus/pci/devices" / f"0000:{r + 1:x}0:00.0"
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{r // 4}\n")
    for node, cpus in ((0, mine[:half]), (1, mine[half:])):
        d = tmp_path / "devices/system/node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(dp.compact_cpulist(cpus) + "\n")
    out = run_worker("--gpus", "8", "--steps", "2", "--warmup", "1", "--fake-sysfs", str(tmp_path))
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and len(out["per_rank"]) == 8
    sets = [set(dp.parse_cpulist(p["cpus"])) for p in out["per_rank"]]
    assert all(len(s) == len(mine) // 8 for s in sets)
    assert len(set().union(*sets)) == sum(len(s) for s in sets) == len(mine)            # disjoint, and nothing left over
    assert [p["numa_node"] for p in out["per_rank"]] == [0, 0, 0, 0, 1, 1, 1, 1]
    for r, p in enumerate(out["per_rank"]):
        assert set(p) >= {"rank", "gpu", "tokens_in", "tokens_out", "ms_per_step", "seed", "hostname", "pid", "pci_bus_id",
                          "numa_node", "cpus", "step_us", "kept_indices"}
        assert p["rank"] == r and p["tokens_out"] < p["tokens_in"] and p["ms_per_step"] > 0
        assert set(p["step_us"]) >= {"min", "median", "p90", "max"}
        assert sets[r] <= set(mine[:half] if r < 4 else mine[half:])
    assert len({p["pid"] for p in out["per_rank"]}) == 8 and len({p["seed"] for p in out["per_rank"]}) == 8
    assert out["t_max"] >= max(p["ms_per_step"] for p in out["per_rank"]) * out["steps"] / 1e3 * 0.999


def test_single_rank_does_not_launch():
    out = run_worker("--gpus", "1", "--steps", "2", "--warmup", "0")
    assert out["n_gpus"] == 1 and out["ranks"] == 1 and len(out["records"]) == 1


def test_single_process_helpers():
    from framefusion_amd import dp
    cpu = torch.device("cpu")
    assert dp.shard(5, 1, 0) == [0, 1, 2, 3, 4] and dp.shard(5, 2, 1) == [1, 3]
    assert dp.aggregate(None, 1.25, 7.0, cpu) == (1.25, 7.0)
    assert dp.gather_lengths(None, 10, 4, cpu) == [(10, 4)]
    assert dp.gather_records(None, (1, 2.5), cpu) == [[1.0, 2.5]]
    assert [k.tolist() for k in dp.gather_kept_indices(None, torch.tensor([0, 3, 4]), 8, cpu)] == [[0, 3, 4]]
    with pytest.raises(ValueError):
        dp.gather_kept_indices(None, torch.arange(9), 8, cpu)
    assert dp.broadcast_config(None, dict(seed=3, p=0.5), cpu) == dict(seed=3, p=0.5)
    assert dp.init("gloo") is None
    timed = dp.timed_steps(None, lambda: 7, 3, 1, cpu)
    t_max, mine, out = timed
    assert out == 7 and t_max == mine > 0 and set(timed.step_us) == {"min", "median", "p90", "max"}
    dp.launch_ranks(1, "unused", [])             # one rank: returns


def test_bench_refuses_to_run_without_a_gpu():
    """The product has no CPU path: bench.py must fail loudly, not fall back."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert res.returncode == 2 and "MI355X" in res.stderr


def test_nccl_init_plumbing_without_a_gpu(monkeypatch):
    """dp.init("nccl", device): what reaches torch.distributed (backend name, device_id=, the one-rank form, the
    rendezvous + HSA_ENABLE_IPC_MODE_LEGACY defaults), checked by capturing the call - and the real call on this GPU-less
    box fails where RCCL needs its device, not earlier (argument errors would surface before that)."""
    import torch.distributed as dist
    from framefusion_amd import dp
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: calls.append((backend, kw)))
    monkeypatch.setattr(dp, "_probe", lambda *a: None)           # (the first collective: nothing to run it on here)
    monkeypatch.setattr(dp, "_agree", lambda dist_, ok, world, rank: ok)
    monkeypatch.setenv("FF_DP_NO_RETRY", "1")
    for k in ("MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    dev = torch.device("cuda", 1)
    assert dp.init("nccl", dev) is dist
    import datetime
    tmo = {"timeout": datetime.timedelta(seconds=300)}                       # (hangs become errors: FF_DP_INIT_TIMEOUT)
    assert calls == [("nccl", {**tmo, "device_id": dev})]                    # world / rank come from the launcher's env
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    # unset + the HSA runtime not started yet in this process: the documented value is filled in (it still takes effect);
    # once CUDA is initialised init() only reports the mode
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    calls.clear()
    assert dp.init("gloo", dev) is dist and calls == [("gloo", tmo)]         # gloo: no device binding
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    assert dp.init("nccl", dev) is None and calls == []                      # one process: no group ...
    assert dp.init("nccl", dev, force=True) is dist                          # ... unless asked for: a one-rank RCCL group
    assert calls == [("nccl", {**tmo, "world_size": 1, "rank": 0, "device_id": dev})]
    monkeypatch.undo()
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(dp.free_port()), PYTHONPATH=ROOT)
        code = ("import torch; from framefusion_amd import dp\n"
                "try:\n    dp.init('nccl', torch.device('cuda', 0), force=True); print('JOINED')\n"
                "except Exception as e:\n    print('REFUSED', type(e).__name__, str(e)[:200])\n")
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        assert "REFUSED" in res.stdout and "JOINED" not in res.stdout, (res.stdout, res.stderr[-500:])
        assert "TypeError" not in res.stdout                                 # it got as far as the backend itself
