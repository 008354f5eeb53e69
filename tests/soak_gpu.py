"""Opt-in soak run (not collected by pytest): the random cascade sweep of test_gpu_random_sweep.py at LARGER shapes -
tens of thousands of tokens, so that the merge kernel's workgroup slot counts (17..53), several select slices and long
by-patch runs are all in play - for as many seeds as the time budget allows.

    python tests/soak_gpu.py [seconds = 240] [first seed = 0]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import framefusion_amd as ffa                                   # noqa: E402
from framefusion_amd.synth import video_tokens                  # noqa: E402
from oracle import ff_oracle as orc                             # noqa: E402
from tests import harness                                       # noqa: E402
from tests.conftest import same_bits                            # noqa: E402
from tests.test_gpu_random_sweep import DTYPES, positions       # noqa: E402

DEV = "cuda:0"


def draw(rng):
    dt = DTYPES[int(rng.integers(0, 3))]
    unit = 4 if dt == torch.float32 else 8
    F = int(rng.choice([16, 37, 64, 100, 150, 256]))
    P = int(rng.choice([49, 100, 196, 210, 400, 576]))
    d = unit * int(rng.choice([1, 2, 4, 8, 16]))
    return dict(dt=dt, F=F, P=P, d=d, pre=int(rng.choice([0, 3, 14])), post=int(rng.choice([0, 5, 20])),
                p_change=float(rng.choice([0.1, 0.3, 0.6, 0.9])), cost=float(rng.choice([0.2, 0.3, 0.5, 0.8])),
                thr=float(rng.choice([0.4, 0.6, 0.8])), lb=float(rng.choice([0.0, 0.05, 0.1])),
                container=str(rng.choice(["qwen2", "mrope", "ids"])), layers=int(rng.integers(1, 4)),
                heads=int(rng.choice([1, 4])), num=int(rng.choice([1, 4])), seed=int(rng.integers(0, 1 << 30)))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, done, skipped, seed = time.time(), 0, 0, first
    while time.time() - t0 < budget:
        rng = np.random.default_rng(50_000 + seed)
        seed += 1
        c = draw(rng)
        h, pt = video_tokens(c["F"], c["P"], c["d"], p_change=c["p_change"], sigma=0.3, sigma_hi=1.4, seed=c["seed"],
                             pre=c["pre"], post=c["post"], dtype=c["dt"], grid=0.125)
        L = h.shape[1]
        try:
            want, _ = harness.run_cascade(orc.OracleFrameFusion(c["cost"], c["thr"], c["lb"]), h.clone(), pt.clone(), c["P"],
                                          positions(c["container"], L, c["dt"]), None, c["layers"], c["heads"], c["num"],
                                          start=c["pre"], n_visual=c["F"] * c["P"])
        except (ValueError, AssertionError):
            skipped += 1
            continue
        pe = positions(c["container"], L, c["dt"])
        pe = pe.to(DEV) if isinstance(pe, torch.Tensor) else [t.to(DEV) for t in pe]
        got, _ = harness.run_cascade(ffa.FrameFusion(c["cost"], c["thr"], c["lb"]), h.to(DEV), pt.to(DEV), c["P"], pe, None,
                                     c["layers"], c["heads"], c["num"], start=c["pre"], n_visual=c["F"] * c["P"])
        assert len(got) == len(want), c
        for a, b in zip(got, want):
            assert (a["tag"], a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
                   (b["tag"], b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"]), (c, a["tag"])
            assert same_bits(a["hidden"].cpu(), b["hidden"]), (c, a["tag"])
            pa, pb = a["pos"], b["pos"]
            if isinstance(pb, torch.Tensor):
                assert torch.equal(pa.cpu(), pb), (c, a["tag"])
            else:
                assert all(same_bits(x.cpu(), y) for x, y in zip(pa, pb)), (c, a["tag"])
        done += 1
    print(f"soak: {done} cascades bit-exact ({skipped} skipped: budget / layout raises), seeds {first}..{seed - 1}, "
          f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
