"""GPU: bench.py as the driver runs it, including N > 1.

* 2 ranks over RCCL (one per GPU) when the box has >= 2 GPUs - `python bench.py --gpus 2` must start its own
  ranks, broadcast rank 0's config, all_gather the per-rank records and print ONE JSON line;
* on a 1-GPU box the same N = 2 job runs with both ranks on GPU 0 and gloo carrying the scalars
  (`--oversubscribe --backend gloo`): every line of the multi-rank path except RCCL itself, with two
  processes driving the HIP kernels concurrently;
* the N = 1 line carries the contract's fields (roofline, cpu_baseline, ...)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, expect_rc=0):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    if expect_rc:
        assert res.returncode == expect_rc, (res.returncode, res.stderr[-2000:])
        return res.stderr
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout       # ONE line on stdout, nothing else (RCCL's banner included)
    return json.loads(lines[0])


def check_multi(out, n):
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["scaling"] == "weak"
    assert [r["rank"] for r in out["per_rank"]] == list(range(n))
    assert all(r["tokens_in"] == 64 * 576 and 0 < r["tokens_out"] < r["tokens_in"] for r in out["per_rank"])
    reduced = sum(r["tokens_in"] - r["tokens_out"] for r in out["per_rank"])
    assert out["value"] == pytest.approx(reduced * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"]), rel=1e-6)
    assert "cpu_baseline" not in out                     # N = 1 only


@pytest.mark.timeout(900)
def test_two_ranks_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    out = run_bench("--gpus", "2", "--steps", "10", "--warmup", "3")
    check_multi(out, 2)
    assert out["collective_backend"] == "nccl"
    assert sorted(r["gpu"] for r in out["per_rank"]) == [0, 1] and out["distinct_devices"] == 2


@pytest.mark.timeout(900)
def test_one_rank_group_over_rccl():
    """What a 1-GPU box can show of the RCCL path: `--force-dist` forms a ONE-rank process group on the "nccl" backend
    (dp.init(..., device_id=...) with HSA_ENABLE_IPC_MODE_LEGACY=0), and rank 0's config broadcast, the all_gather of the
    per-rank records and the all_reduce of time / units all execute inside RCCL on the MI355X before the line is printed.
    (Two ranks on one device are refused by RCCL itself: "Duplicate GPU detected".)"""
    out = run_bench("--steps", "10", "--warmup", "3", "--force-dist", "--no-cpu-baseline", "--no-extra", "--no-pmc")
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["collective_backend"] == "nccl"
    assert [r["rank"] for r in out["per_rank"]] == [0] and out["per_rank"][0]["tokens_in"] == 64 * 576
    assert 0 < out["per_rank"][0]["tokens_out"] < 64 * 576 and out["value"] > 0


@pytest.mark.timeout(900)
def test_two_ranks_sharing_one_gpu_over_gloo():
    out = run_bench("--gpus", "2", "--steps", "10", "--warmup", "3", "--backend", "gloo", "--oversubscribe")
    check_multi(out, 2)
    assert out["collective_backend"] == "gloo"


@pytest.mark.timeout(1200)
def test_eight_ranks_sharing_one_gpu_over_gloo():
    """The N = 8 control flow of the driver's scaling run, executed once on whatever the box has: 8 processes (rank r -> GPU
    r % device_count) drive the kernels concurrently, rank 0's config is broadcast, 8 records + 8 identities are
    all_gathered, one JSON line comes out.  (RCCL itself needs 8 devices; gloo carries the scalars here.)"""
    # the last rank dawdles 300 ms inside every barrier (FF_DP_SLOW_BARRIER_MS): the closing barrier of the timed region is
    # in nobody's clock, so 5 steps of eight ranks sharing one GPU (~1.2 ms each) stay far below it
    out = run_bench("--gpus", "8", "--steps", "5", "--warmup", "2", "--backend", "gloo", "--oversubscribe", "--dp-hooks", "tests.dp_faults",
                    env_extra={"FF_DP_SLOW_BARRIER_MS": "300"})
    check_multi(out, 8)
    assert out["collective_backend"] == "gloo"
    assert out["ms_per_step"] * out["steps"] < 150, out["ms_per_step"]
    for r in out["per_rank"]:
        su = r["step_us"]
        assert 0 < su["min"] <= su["median"] <= su["p90"] <= su["max"]
        assert "numa_node" in r and "cpus" in r             # (None / None where the platform does not say)
        assert r["ms_per_step"] <= out["ms_per_step"] * (1 + 1e-9)
    assert [r["seed"] for r in out["per_rank"]] == list(range(1234, 1242))            # seed + rank: 8 distinct samples
    # ... every rank a different video (the top-k branch cuts all of them to the same length: the samples show in their
    # threshold counts)
    assert len({r["similarities_above_threshold"] for r in out["per_rank"]}) == 8
    # the kept-index lists were all_gathered: 8 different ones, each of its rank's output length
    assert all(r["kept_indices"]["n"] == r["tokens_out"] for r in out["per_rank"])
    assert len({r["kept_indices"]["sum"] for r in out["per_rank"]}) == 8
    assert len({r["pid"] for r in out["per_rank"]}) == 8
    n_dev = torch.cuda.device_count()
    assert [r["gpu"] for r in out["per_rank"]] == [r % n_dev for r in range(8)]
    assert all(r["hostname"] and r["pci_bus_id"] for r in out["per_rank"])
    assert out["distinct_devices"] == min(8, n_dev)
    assert out["dp_attempt"] == 0 and "HSA_ENABLE_IPC_MODE_LEGACY" in out["ipc_mode"]


@pytest.mark.timeout(900)
def test_two_samples_per_gpu_go_through_the_pair():
    """`--samples 2` on one GPU: the rank owns two samples per step and keeps both in flight (FrameFusionPair); the line says so
    and its value counts both."""
    out = run_bench("--samples", "2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extra", "--no-pmc")
    cfg = out["config"]
    assert cfg["samples_per_gpu_per_step"] == 2 and cfg["samples_per_step"] == 2 and "FrameFusionPair" in cfg["workload"]
    one = cfg["tokens_in"] - cfg["tokens_out"]
    assert out["value"] == pytest.approx(2 * one / (out["ms_per_step"] * 1e-3), rel=1e-6)
    assert out["value"] > 1.5e8                               # (one sample per step runs at ~1.9e8 on an MI355X)


@pytest.mark.timeout(900)
def test_an_n_gpu_line_cannot_be_n_ranks_on_one_device():
    """`--gpus 2` without --oversubscribe must be two devices over RCCL: on a 1-GPU box every rank refuses before anything
    runs (and, should ranks ever land on one device anyway, the all_gathered identities refuse the line: exit 3)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a 1-GPU box")
    err = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", expect_rc=1)       # torch.distributed.run reports its children's failure
    assert "GPU(s) visible" in err


@pytest.mark.timeout(900)
def test_single_gpu_line_has_the_contract_fields():
    out = run_bench("--steps", "10", "--warmup", "3", "--cpu-calls", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["dtype"] == "bf16" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.3 < rf["frac"] < 1.0 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    # the traffic figure names where it comes from, and is withheld when the PMC passes are of another build
    ts = rf["traffic_source"]
    # measured live (two rocprofv3 --pmc child passes) where rocprofv3 is installed; else from the committed summaries, withheld
    # when those are of another build
    assert ts is not None
    if ts.get("live"):
        assert rf["traffic"] > 0 and ts["launches"]["FETCH_SIZE"] > 20 and ts["launches"]["WRITE_SIZE"] > 20
    else:
        assert "stale" in ts or "error" in ts or ts["profiled_source_hash"] == ts["running_source_hash"]
        assert (rf["traffic"] is None) == ("stale" in ts or "error" in ts)
    if rf["traffic"] is not None:
        assert 0.8 < rf["traffic"] / rf["algorithmic_bytes"] < 1.3
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert "workload" in out["config"] and "model" not in out["config"]
    ex = out["extra"]
    assert ex["two_samples_per_gpu"]["tokens_reduced_per_s"] > 0 and ex["packer_scalars_step_us"]["median"] > 0
    seven_b = [c for c in ex["configs"] if "LLaVA-Video-7B" in c["workload"]][0]
    assert 0 < seven_b["us_back_to_back"] <= seven_b["us"] * 1.2
