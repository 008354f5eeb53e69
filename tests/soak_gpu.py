"""Opt-in soak run (not collected by pytest): the random cascade sweep of test_gpu_random_sweep.py at LARGER shapes -
tens of thousands of tokens, so that the merge kernel's workgroup slot counts (17..53), several select slices and long
by-patch runs are all in play - for as many seeds as the time budget allows.

    python tests/soak_gpu.py [seconds = 240] [first seed = 0]
    python tests/soak_gpu.py importance [seconds = 120] [first seed = 0]     # K5: random head layouts vs the oracle
    python tests/soak_gpu.py residual [seconds = 120] [first seed = 0]       # call B fused with the residual add
    python tests/soak_gpu.py pair [seconds = 120] [first seed = 0]           # two prefills interleaved through FrameFusionPair
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


def importance_soak(budget, first):
    """Grid-valued q / k (exact scores): every weight within 1 ulp of the oracle's, few of them different."""
    t0, done, seed, worst, worst32 = time.time(), 0, first, 0.0, 0.0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(90_000 + seed)
        seed += 1
        dt = DTYPES[int(rng.integers(0, 3))]
        dh = int(rng.choice([16, 24, 32, 64, 128, 256]))
        Hk = int(rng.choice([1, 2, 4, 8]))
        H = Hk * int(rng.choice([1, 2, 4, 7, 8]))
        num = int(rng.choice([1, 1, 4, 16]))
        S = int(rng.choice([17, 255, 256, 257, 1000, 4097, 12000, 33000]))
        if S <= num:
            continue
        g = torch.Generator().manual_seed(seed)
        q = harness.snap(0.5 * torch.randn(1, H, num, dh, generator=g), dt)
        k = harness.snap(0.5 * torch.randn(1, Hk, S, dh, generator=g), dt)
        want = orc.last_query_attention(q, k, num=num, is_causal=True, enable_gqa=True)
        got = ffa.scaled_dot_product_attention(q.to(DEV), k.to(DEV), None, num=num, is_causal=True, enable_gqa=True).cpu()
        cfg = (dt, dh, H, Hk, num, S, seed - 1)

        def close(a, b, ulps=1):
            if dt == torch.float32:
                return torch.allclose(a, b, rtol=2e-5 * ulps, atol=1e-30)       # (33 k exponentials added in two orders)
            # non-negative 16-bit values: neighbouring bit patterns are neighbouring values (subnormals included)
            return int((a.view(torch.int16).int() - b.view(torch.int16).int()).abs().max()) <= ulps

        assert close(got, want), cfg
        rate = float((got.float() != want.float()).float().mean())
        # the two sides add the row's exponentials in different orders: a relative 1e-6 on the sum moves ~1e-6 / ulp of the
        # probabilities across a rounding boundary (bf16 ulp 2^-8, fp16 2^-11); fp32 differs by a few ulps of exp everywhere
        assert dt == torch.float32 or rate <= (2e-3 if dt == torch.bfloat16 else 1.6e-2), (cfg, rate)
        worst = max(worst, rate if dt != torch.float32 else 0.0)
        if dt == torch.float32:
            worst32 = max(worst32, float(((got - want).abs() / want.abs().clamp_min(1e-30)).max()))
        imp = ffa.last_query_importance(q.to(DEV), k.to(DEV), num=num, is_causal=True).reshape(-1).cpu()
        # the fused importance is the head mean of THIS build's weights (1 ulp: the fp32 order of the mean); against the
        # oracle's it inherits their 1-ulp differences, which can add up to 2 across a binade boundary
        assert close(imp, torch.mean(got, dim=(1, 2))[0]), cfg
        assert close(imp, torch.mean(want, dim=(1, 2))[0], ulps=2), cfg
        done += 1
    print(f"importance soak: {done} head layouts within 1 ulp (worst 16-bit mismatch rate {worst:.2e}, worst fp32 relative difference {worst32:.1e}), "
          f"seeds {first}..{seed - 1}, {time.time() - t0:.0f} s")


def residual_soak(budget, first):
    """forward_residual(residual, attn_out) against the oracle fed with the eager sum: one merge call, then (when the
    state machine asks for it) the prune call, large shapes, bit for bit."""
    from tests.test_gpu_residual import split
    t0, done, seed = time.time(), 0, first
    while time.time() - t0 < budget:
        rng = np.random.default_rng(130_000 + seed)
        seed += 1
        c = draw(rng)
        h, pt = video_tokens(c["F"], c["P"], c["d"], p_change=c["p_change"], sigma=0.3, sigma_hi=1.4, seed=c["seed"],
                             pre=c["pre"], post=c["post"], dtype=c["dt"], grid=0.125, clip=2.0)
        L, nvis = h.shape[1], c["F"] * c["P"]
        o = orc.OracleFrameFusion(c["cost"], c["thr"], c["lb"])
        f = ffa.FrameFusion(c["cost"], c["thr"], c["lb"])
        o.prepare(pt.clone(), c["P"], c["pre"], c["pre"] + nvis, nvis, L)
        f.prepare(pt.to(DEV), c["P"], c["pre"], c["pre"] + nvis, nvis, L)
        ho, po, hg, pg = h, torch.arange(L)[None], None, torch.arange(L, device=DEV)[None]
        for layer in range(3):
            res, attn = split(ho, c["seed"] % 1000 + layer)
            w = harness.attention_stub(c["heads"], c["num"], ho.shape[1], c["dt"])
            need_w = o.finish_merging and not o.finish_pruning
            try:
                ho, po, _ = o.forward(res + attn, po, None, w if need_w else None)
            except ValueError:
                break
            hg, pg, _ = f.forward_residual(res.to(DEV), attn.to(DEV), pg, None, w.to(DEV) if need_w else None)
            assert (f.finish_merging, f.finish_pruning, f.sparsity_list) == (o.finish_merging, o.finish_pruning, o.sparsity_list), (c, layer)
            assert torch.equal(pg.cpu(), po) and same_bits(hg.cpu(), ho), (c, layer)
            ho = harness.layer_stub(ho, layer)
        done += 1
    print(f"residual soak: {done} prefills (up to 3 fused calls each) bit-exact, seeds {first}..{seed - 1}, {time.time() - t0:.0f} s")


def pair_soak(budget, first):
    """Two random prefills (different shapes, regimes, containers) interleaved through FrameFusionPair - every call of one
    sample submitted before the other sample's call is collected - against the same two prefills through independent
    instances: every call's rows, positions, flags and budget lists bit for bit."""
    from framefusion_amd.pair import FrameFusionPair
    from tests.test_gpu_pair import run_independent, run_paired, equal_logs
    t0, done, skipped, seed = time.time(), 0, 0, first
    while time.time() - t0 < budget:
        rng = np.random.default_rng(70_000 + seed)
        seed += 1
        cs = [draw(rng), draw(rng)]                   # (two unrelated samples: shapes, dtypes, regimes, containers all differ)
        samples = []
        for c in cs:
            h, pt = video_tokens(c["F"], c["P"], c["d"], p_change=c["p_change"], sigma=0.3, sigma_hi=1.4, seed=c["seed"],
                                 pre=c["pre"], post=c["post"], dtype=c["dt"], grid=0.125)
            L = h.shape[1]
            pe = positions(c["container"], L, c["dt"])
            pe = pe.to(DEV) if isinstance(pe, torch.Tensor) else [t.to(DEV) for t in pe]
            samples.append(dict(h=h.to(DEV), pt=pt.to(DEV), P=c["P"], pre=c["pre"], n=c["F"] * c["P"], L=L, pos=pe))
        mk = [lambda c=c: ffa.FrameFusion(c["cost"], c["thr"], c["lb"]) for c in cs]
        try:
            want = [run_independent(mk[x](), samples[x], cs[x]["layers"]) for x in (0, 1)]
        except ValueError:                       # "The cost is too small": the same raise path is unit-tested
            skipped += 1
            continue
        layers = max(cs[0]["layers"], cs[1]["layers"])
        if cs[0]["layers"] != cs[1]["layers"]:
            want = [run_independent(mk[x](), samples[x], layers) for x in (0, 1)]
        got = run_paired(FrameFusionPair(mk[0](), mk[1]()), samples[0], samples[1], layers)
        equal_logs(got[0], want[0])
        equal_logs(got[1], want[1])
        done += 1
    print(f"pair soak: {done} pairs of prefills bit-exact against independent instances ({skipped} skipped), seeds {first}..{seed - 1}, "
          f"{time.time() - t0:.0f} s")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "pair":
        return pair_soak(float(sys.argv[2]) if len(sys.argv) > 2 else 120.0, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "residual":
        return residual_soak(float(sys.argv[2]) if len(sys.argv) > 2 else 120.0, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "importance":
        return importance_soak(float(sys.argv[2]) if len(sys.argv) > 2 else 120.0, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
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
