"""CPU, world_size 2 over gloo: the data-parallel plumbing bench.py uses (sharding of independent
samples, max-time / summed-token aggregation, length gather).  The data path itself has no
collective, so this is all the N > 1 logic there is."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from framefusion_amd import dp
    from oracle import ff_oracle as orc
    from framefusion_amd.synth import video_tokens
    dist = dp.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    mine = dp.shard(5, world, rank)
    reduced = 0
    lens = (0, 0)
    for idx in mine:     # each rank reduces its own samples with the CPU oracle (no GPU here)
        h, pt = video_tokens(6, 8, 32, p_change=0.3, seed=dp.sample_seed(100, idx), pre=1, post=1, grid=0.125)
        f = orc.OracleFrameFusion(0.3, 0.6, 0.1)
        f.prepare(pt, 8, 1, 49, 48, h.shape[1])
        o, _, _ = f.forward(h, torch.arange(h.shape[1])[None], None)
        reduced += h.shape[1] - o.shape[1]
        lens = (h.shape[1], o.shape[1])
    dp.barrier(dist)
    t_max, total = dp.aggregate(dist, 0.5 + rank, float(reduced), torch.device("cpu"))
    gathered = dp.gather_lengths(dist, lens[0], lens[1], torch.device("cpu"))
    out.put((rank, mine, reduced, t_max, total, gathered))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_and_aggregation():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, m0, red0, t0, tot0, g0), (r1, m1, red1, t1, tot1, g1) = res
    assert m0 == [0, 2, 4] and m1 == [1, 3]                      # round-robin, disjoint, complete
    assert t0 == t1 == 1.5                                        # slowest rank's time everywhere
    assert tot0 == tot1 == float(red0 + red1) and red0 > 0 and red1 > 0
    assert g0 == g1 and len(g0) == 2 and all(a > b for a, b in g0)


def test_single_process_helpers():
    from framefusion_amd import dp
    assert dp.shard(5, 1, 0) == [0, 1, 2, 3, 4]
    assert dp.aggregate(None, 1.25, 7.0, torch.device("cpu")) == (1.25, 7.0)
    assert dp.gather_lengths(None, 10, 4, torch.device("cpu")) == [(10, 4)]
    assert dp.init("gloo") is None or True
