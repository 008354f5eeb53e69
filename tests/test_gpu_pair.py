"""GPU: FrameFusionPair (FrameFusion.submit / collect, ff_ctx_merge_submit / _collect) - two samples in flight from one host thread give, bit for bit, what two
independent instances give (the reference's form: one instance per sample, script/demo/llava_video_compare.py:217-223)."""
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd.pair import FrameFusionPair
from framefusion_amd.synth import video_tokens, rotary_tables
from tests import harness
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sample(F, P, d, p_change, seed, pre, post, dtype, container):
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.4, seed=seed, pre=pre, post=post, dtype=dtype, grid=0.125)
    L = h.shape[1]
    if container == "ids":
        pos = torch.arange(L, device=DEV)[None]
    else:
        pos = [t.to(DEV) for t in rotary_tables(L, 16, dtype, mrope=(container == "mrope"))]
    return dict(h=h.to(DEV), pt=pt.to(DEV), P=P, pre=pre, n=F * P, L=L, pos=pos)


def prefill(ff, s, step_fn, layers, heads=4, num=1):
    """The reference's call protocol (tests/harness.run_cascade) with the FrameFusion call itself delegated to `step_fn`."""
    ff.prepare(s["pt"], s["P"], s["pre"], s["pre"] + s["n"], s["n"], s["L"])
    h, pe, mask = s["h"], (list(s["pos"]) if isinstance(s["pos"], list) else s["pos"]), None
    layer = -1                                   # call A, then call B per layer
    while True:
        attn_w = None
        if layer >= 0:
            if h.shape[1] > 1 and ff.finish_merging and not ff.finish_pruning:
                attn_w = harness.attention_stub(heads, num, h.shape[1], h.dtype, h.device)
            h = harness.layer_stub(h, layer)
        h, pe, mask = yield (h, pe, mask, attn_w)
        layer += 1
        if layer >= layers:
            return


def run_independent(ff, s, layers):
    log = []
    gen = prefill(ff, s, None, layers)
    args = next(gen)
    while True:
        out = ff(*args)
        log.append((out[0], out[1], bool(ff.finish_merging), bool(ff.finish_pruning), list(ff.sparsity_list)))
        try:
            args = gen.send(out)
        except StopIteration:
            return log


def run_paired(pair, sa, sb, layers):
    """Both prefills interleaved through the pair: sample 1's call is submitted before sample 0's is collected and vice
    versa - always one call enqueued ahead of the one being waited for."""
    logs = ([], [])
    gens = (prefill(pair.a, sa, None, layers), prefill(pair.b, sb, None, layers))
    args = [next(gens[0]), next(gens[1])]
    alive = [True, True]
    x = 0
    while True:
        if alive[x] and pair._pending[x] is None:
            pair.submit(x, *args[x])
        y = 1 - x
        if pair._pending[y] is not None:
            out = pair.collect(y)
            ff = pair.ffs[y]
            logs[y].append((out[0], out[1], bool(ff.finish_merging), bool(ff.finish_pruning), list(ff.sparsity_list)))
            try:
                args[y] = gens[y].send(out)
            except StopIteration:
                alive[y] = False
        if pair._pending[0] is None and pair._pending[1] is None and not (alive[0] or alive[1]):
            return logs
        x = y


def equal_logs(got, want):
    assert len(got) == len(want)
    for (h1, p1, fm1, fp1, sp1), (h2, p2, fm2, fp2, sp2) in zip(got, want):
        assert (fm1, fp1, sp1) == (fm2, fp2, sp2)
        assert h1.shape == h2.shape and same_bits(h1.cpu(), h2.cpu())
        if isinstance(p2, torch.Tensor):
            assert torch.equal(p1, p2)
        else:
            assert all(same_bits(x.cpu(), y.cpu()) for x, y in zip(p1, p2))


@pytest.mark.parametrize("dtype,container,shape_a,shape_b", [
    (torch.bfloat16, "qwen2", (12, 40, 512, 0.5, 4, 6), (9, 33, 512, 0.2, 0, 3)),          # different lengths, different regimes
    (torch.float16, "mrope", (8, 64, 256, 0.3, 2, 2), (8, 64, 256, 0.95, 2, 2)),           # one sample stops merging first
    (torch.float32, "ids", (5, 16, 64, 0.6, 1, 0), (21, 7, 128, 0.05, 0, 9)),
])
def test_pair_matches_two_independent_instances(dtype, container, shape_a, shape_b):
    sa = sample(*shape_a[:4], seed=11, pre=shape_a[4], post=shape_a[5], dtype=dtype, container=container)
    sb = sample(*shape_b[:4], seed=12, pre=shape_b[4], post=shape_b[5], dtype=dtype, container=container)
    want_a = run_independent(ffa.FrameFusion(0.3, 0.6, 0.1), sa, 4)
    want_b = run_independent(ffa.FrameFusion(0.3, 0.6, 0.1), sb, 4)
    # both output forms: exactly sized (default: K1 + plan at submit, merge kernel at collect) and views of input-length buffers
    # (everything enqueued at submit: ff_ctx_merge_submit / _collect)
    for views in (False, True):
        kw = dict(compact_outputs=False) if views else {}
        pair = FrameFusionPair(ffa.FrameFusion(0.3, 0.6, 0.1, **kw), ffa.FrameFusion(0.3, 0.6, 0.1, **kw))
        for _ in range(2):                      # a second prefill through the same instances (scratch reuse, order maintenance)
            got_a, got_b = run_paired(pair, sa, sb, 4)
            equal_logs(got_a, want_a)
            equal_logs(got_b, want_b)
    torch.cuda.synchronize()


def test_pair_at_the_headline_size_and_with_a_mask():
    """64 x 576 x 4096 bf16, two different videos: the pair equals two independent calls; a small pair with attention masks too."""
    F, P, d = 64, 576, 4096
    ss = []
    for seed in (1234, 1235):
        h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=seed, dtype=torch.bfloat16, device=DEV)
        L = h.shape[1]
        ss.append(dict(h=h, pt=pt, L=L, pos=list(rotary_tables(L, 128, torch.bfloat16, device=DEV))))
    want = []
    for s in ss:
        f = ffa.FrameFusion(0.3, 0.6, 0.1)
        f.prepare(s["pt"], P, 0, s["L"], s["L"], s["L"])
        o, pe, _ = f(s["h"], list(s["pos"]), None)
        want.append((o.clone(), [t.clone() for t in pe], f.patch_type.clone()))
    pair = FrameFusionPair(ffa.FrameFusion(0.3, 0.6, 0.1), ffa.FrameFusion(0.3, 0.6, 0.1))

    def calls(n):
        for i in range(n):
            x = i & 1
            pair.ffs[x].prepare(ss[x]["pt"], P, 0, ss[x]["L"], ss[x]["L"], ss[x]["L"])
            yield (x, ss[x]["h"], list(ss[x]["pos"]), None)
    for i, (o, pe, _) in enumerate(pair.run(calls(6))):
        wo, wpe, wpt = want[i & 1]
        f = pair.ffs[i & 1]
        assert torch.equal(o, wo) and all(torch.equal(x, y) for x, y in zip(pe, wpe))
        assert f.finish_merging and f.finish_pruning
    assert torch.equal(pair.a.patch_type, want[0][2]) and torch.equal(pair.b.patch_type, want[1][2])
    assert want[0][0].shape == want[1][0].shape and not torch.equal(want[0][0], want[1][0])          # two different videos
    # masks: gathered behind each sample's call on its own stream
    sa = sample(6, 20, 64, 0.4, seed=5, pre=2, post=3, dtype=torch.bfloat16, container="qwen2")
    sb = sample(7, 18, 64, 0.4, seed=6, pre=0, post=5, dtype=torch.bfloat16, container="qwen2")
    g = torch.Generator().manual_seed(0)
    masks = [torch.randn(1, 1, s["L"], s["L"], generator=g).to(torch.bfloat16).to(DEV) for s in (sa, sb)]
    want = []
    for s, m in zip((sa, sb), masks):
        f = ffa.FrameFusion(0.3, 0.6, 0.1)
        f.prepare(s["pt"], s["P"], s["pre"], s["pre"] + s["n"], s["n"], s["L"])
        want.append(f(s["h"], list(s["pos"]), m))
    pair = FrameFusionPair(ffa.FrameFusion(0.3, 0.6, 0.1), ffa.FrameFusion(0.3, 0.6, 0.1))
    for f, s in zip((pair.a, pair.b), (sa, sb)):
        f.prepare(s["pt"], s["P"], s["pre"], s["pre"] + s["n"], s["n"], s["L"])
    pair.submit(0, sa["h"], list(sa["pos"]), masks[0])
    pair.submit(1, sb["h"], list(sb["pos"]), masks[1])
    outs = [pair.collect(0), pair.collect(1)]
    for (o, pe, m), (wo, wpe, wm) in zip(outs, want):
        assert same_bits(o.cpu(), wo.cpu()) and same_bits(m.cpu(), wm.cpu()) and m.shape[-1] == o.shape[1]


def test_pair_refuses_misuse_and_survives_a_bad_sample():
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    with pytest.raises(ffa.FrameFusionHipError):
        FrameFusionPair(f, f)
    sa = sample(6, 20, 64, 0.4, seed=5, pre=2, post=3, dtype=torch.bfloat16, container="qwen2")
    pair = FrameFusionPair(ffa.FrameFusion(0.3, 0.6, 0.1), ffa.FrameFusion(0.3, 0.6, 0.1))
    for ff in pair.ffs:
        ff.prepare(sa["pt"], sa["P"], sa["pre"], sa["pre"] + sa["n"], sa["n"], sa["L"])
    pair.submit(0, sa["h"], list(sa["pos"]), None)
    with pytest.raises(ffa.FrameFusionHipError):
        pair.submit(0, sa["h"], list(sa["pos"]), None)                    # sample 0 already has a call in flight
    with pytest.raises(ffa.FrameFusionHipError):
        pair.a.submit(sa["h"], list(sa["pos"]), None)                     # ... and so has its instance
    with pytest.raises(ffa.FrameFusionHipError):
        pair.a(sa["h"], list(sa["pos"]), None)                            # (a plain call in between is refused as well)
    with pytest.raises(ffa.FrameFusionHipError):
        pair.collect(1)                                                   # nothing in flight for sample 1
    bad = sa["h"][:, :-1]                                                 # patch_type does not cover this sequence
    with pytest.raises(ffa.FrameFusionHipError):
        pair.submit(1, bad, list(sa["pos"]), None)
    oa = pair.collect(0)                                                  # sample 0 is unaffected
    want = ffa.FrameFusion(0.3, 0.6, 0.1)
    want.prepare(sa["pt"], sa["P"], sa["pre"], sa["pre"] + sa["n"], sa["n"], sa["L"])
    wo = want(sa["h"], list(sa["pos"]), None)[0]
    assert same_bits(oa[0].cpu(), wo.cpu())
    # both instances stay usable; submit / collect on a plain instance (PyTorch's current stream) equals forward
    for ff in pair.ffs:
        ff.prepare(sa["pt"], sa["P"], sa["pre"], sa["pre"] + sa["n"], sa["n"], sa["L"])
    pair.submit(1, sa["h"], list(sa["pos"]), None)
    pair.submit(0, sa["h"], list(sa["pos"]), None)
    assert same_bits(pair.collect(1)[0].cpu(), wo.cpu()) and same_bits(pair.collect(0)[0].cpu(), wo.cpu())
    solo = ffa.FrameFusion(0.3, 0.6, 0.1)
    solo.prepare(sa["pt"], sa["P"], sa["pre"], sa["pre"] + sa["n"], sa["n"], sa["L"])
    t = solo.submit(sa["h"], list(sa["pos"]), None)
    assert same_bits(solo.collect(t)[0].cpu(), wo.cpu())
    torch.cuda.synchronize()
