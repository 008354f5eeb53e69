"""GPU: every launch of a call goes to the caller's CURRENT stream (SURVEY.md §8b "all work on the caller's current
stream"; the reference's demo runs two model replicas from two threads, llava_video_compare.py:217-223, never sharing
an instance).  A prefill on a side stream, and two instances interleaved on two side streams, against the oracle."""
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from oracle import ff_oracle as orc
from tests import harness
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sample(seed, p_change):
    F, P, d, pre, post = 20, 64, 256, 4, 6
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.4, seed=seed, pre=pre, post=post,
                         dtype=torch.bfloat16, grid=0.125)
    return h, pt, P, pre, F * P


def oracle_cascade(h, pt, P, pre, nvis, layers=3):
    want, _ = harness.run_cascade(orc.OracleFrameFusion(0.3, 0.6, 0.1), h.clone(), pt.clone(), P,
                                  rotary_tables(h.shape[1], 16, torch.bfloat16), None, layers, 4, 1, start=pre, n_visual=nvis)
    return want


def check(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert (a["tag"], a["length"], a["finish_merging"], a["finish_pruning"], a["sparsity"]) == \
               (b["tag"], b["length"], b["finish_merging"], b["finish_pruning"], b["sparsity"])
        assert same_bits(a["hidden"].cpu(), b["hidden"])


def test_prefill_on_a_side_stream():
    h, pt, P, pre, nvis = sample(3, 0.5)
    want = oracle_cascade(h, pt, P, pre, nvis)
    side = torch.cuda.Stream(device=DEV)
    hd, ptd = h.to(DEV), pt.to(DEV)
    pe = [t.to(DEV) for t in rotary_tables(h.shape[1], 16, torch.bfloat16)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), hd, ptd, P, pe, None, 3, 4, 1, start=pre, n_visual=nvis)
    side.synchronize()
    check(got, want)


def test_two_instances_on_two_streams_interleaved():
    """Call by call alternation between two prefills that live on different streams: each instance keeps its own
    scratch, select tables and result block, so nothing of one shows up in the other."""
    sa, sb = sample(5, 0.2), sample(6, 0.6)
    wa, wb = oracle_cascade(*sa, layers=2), oracle_cascade(*sb, layers=2)
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    ffs = [ffa.FrameFusion(0.3, 0.6, 0.1), ffa.FrameFusion(0.3, 0.6, 0.1)]
    state = []
    for (h, pt, P, pre, nvis), f in zip((sa, sb), ffs):
        L = h.shape[1]
        f.prepare(pt.to(DEV), P, pre, pre + nvis, nvis, L)
        state.append(dict(h=h.to(DEV), pe=[t.to(DEV) for t in rotary_tables(L, 16, torch.bfloat16)], recs=[]))
    torch.cuda.synchronize()
    # the harness protocol by hand: call A, then per layer the stub + call B (with weights when the hook asks for them)
    for step in range(3):
        for x in (0, 1):
            f, st = ffs[x], state[x]
            with torch.cuda.stream(streams[x]):
                if step > 0:
                    st["h"] = harness.layer_stub(st["h"], step - 1)
                w = None
                if f.finish_merging and not f.finish_pruning:
                    w = harness.attention_stub(4, 1, st["h"].shape[1], torch.bfloat16, device=DEV)
                st["h"], st["pe"], _ = f(st["h"], st["pe"], None, w)
                st["recs"].append((st["h"].shape[1], f.finish_merging, f.finish_pruning, st["h"]))
    for s in streams:
        s.synchronize()
    for recs, want in ((state[0]["recs"], wa), (state[1]["recs"], wb)):
        assert len(recs) == len(want)
        for (length, fm, fp, hid), b in zip(recs, want):
            assert (length, fm, fp) == (b["length"], b["finish_merging"], b["finish_pruning"])
            assert same_bits(hid.cpu(), b["hidden"])


def test_two_threads_two_instances():
    """The reference's demo shape (llava_video_compare.py:217-223, 310-313): two Python threads, each with its own
    instance (and here its own stream on the one GPU).  The result-block poll of one thread must not starve the other."""
    import threading
    samples = [sample(8, 0.2), sample(9, 0.5)]
    wants = [oracle_cascade(*s) for s in samples]
    results, errors = [None, None], []

    def work(x):
        try:
            h, pt, P, pre, nvis = samples[x]
            stream = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(stream):
                pe = [t.to(DEV) for t in rotary_tables(h.shape[1], 16, torch.bfloat16)]
                for _ in range(5):                                   # several prefills per thread
                    got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), h.to(DEV), pt.to(DEV), P, list(pe), None, 3, 4, 1,
                                                 start=pre, n_visual=nvis)
            stream.synchronize()
            results[x] = got
        except Exception as e:                                       # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(x,)) for x in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a worker thread hangs"
    assert not errors, errors
    for got, want in zip(results, wants):
        check(got, want)


def test_two_gpus_in_one_process_from_two_threads():
    """The reference's demo layout (llava_video_compare.py:217-223, 310-313): two replicas on two GPUs, one thread each,
    one process.  The library's host-side per-device caches (function attributes for dynamic LDS, the merge kernel's
    occupancy) are keyed by device and the poll runs with the interpreter lock released, so the two prefills neither
    share state nor serialise.  Needs two devices (skipped on the 1-GPU boxes of this pool)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import threading
    results, errors = {}, []

    def work(dev_index, seed, p_change):
        try:
            dev = f"cuda:{dev_index}"
            h, pt, P, pre, nvis = sample(seed, p_change)
            pe = [t.to(dev) for t in rotary_tables(h.shape[1], 16, torch.bfloat16)]
            with torch.cuda.device(dev):
                got, _ = harness.run_cascade(ffa.FrameFusion(0.3, 0.6, 0.1), h.to(dev), pt.to(dev), P, pe, None, 3, 4, 1,
                                             start=pre, n_visual=nvis)
                torch.cuda.synchronize()
            results[dev_index] = (got, oracle_cascade(h, pt, P, pre, nvis))
        except Exception as e:                                    # pragma: no cover
            errors.append((dev_index, repr(e)))
    threads = [threading.Thread(target=work, args=(i, 40 + i, 0.3 + 0.3 * i)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        check(*results[i])
