"""GPU: the one-launch merge kernel (csrc/ff_resident.hip) against the three launches it replaces - same instance class, same
inputs, `one_launch` on / off - bit for bit: outputs, position tables, patch types, the plan's arrays, the order handed to the next
call.  The shapes aim at what is new in it: segments of ~50 slots per workgroup with runs that cross segment boundaries, rows that
end inside a 1 KiB tile, many non-visual rows, sequences of more than 16 384 positions, order-maintained second calls, outputs
sized for the wrong branch (the launch stops behind its plan) and no outputs at all.  The oracle comparison of the same path is
in test_gpu_parity / test_full_size_golden / test_gpu_random_sweep (the path is the default wherever it fits)."""
import threading

import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables
from tests import harness
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return x.to(DEV) if isinstance(x, torch.Tensor) else [t.to(DEV) for t in x]


def pair(cost=0.3, thr=0.6, lb=0.1, compact=True):
    a, b = ffa.FrameFusion(cost, thr, lb, compact_outputs=compact), ffa.FrameFusion(cost, thr, lb, compact_outputs=compact)
    b.one_launch = False
    return a, b


def prepare(ff, pt, P, pre, nvis, L):
    ff.prepare(pt, P, pre, pre + nvis, nvis, L)


def same_call(fa, fb, h, pe, expect_one_launch=True, residual=None):
    """One forward call through both instances; compares everything the call leaves behind.  `residual`: call B of a decoder
    layer with the add fused in (the rows are T(h + residual))."""
    pa = [t.clone() for t in pe] if isinstance(pe, list) else pe.clone()
    pb = [t.clone() for t in pe] if isinstance(pe, list) else pe.clone()
    oa, qa, _ = fa(h, pa, None, None, residual=residual)
    ob, qb, _ = fb(h, pb, None, None, residual=residual)
    assert fa.last_call["one_launch"] == expect_one_launch and not fb.last_call["one_launch"]
    assert oa.shape == ob.shape and same_bits(oa.cpu(), ob.cpu())
    if isinstance(qa, list):
        assert all(same_bits(x.cpu(), y.cpu()) for x, y in zip(qa, qb))
    else:
        assert torch.equal(qa, qb)
    assert torch.equal(fa.patch_type, fb.patch_type)
    assert (fa.finish_merging, fa.finish_pruning, fa.sparsity_list) == (fb.finish_merging, fb.finish_pruning, fb.sparsity_list)
    ca, cb = fa.last_call, fb.last_call
    for key in ("L_in", "L_out", "nv", "ftn", "count", "branch", "k"):
        assert ca[key] == cb[key], key
    la, lb_ = fa.last_plan(), fb.last_plan()
    for key in ("keep", "member", "sim", "order"):
        assert torch.equal(la[key].cpu(), lb_[key].cpu()), key
    if ca["L_out"] != ca["L_in"]:
        # the by-patch order of the compacted sequence and its inverse, as the next call will read them
        n = ca["L_out"]
        sa, sb = ca["scratch"], cb["scratch"]
        assert torch.equal(sa.order[:n].cpu(), sb.order[:n].cpu()) and torch.equal(sa.inv[:n].cpu(), sb.inv[:n].cpu())
    return oa, qa


SHAPES = [
    # F, P, d, pre, post, p_change, cost, thr       what it aims at
    (64, 210, 3584, 14, 20, 0.2, 0.3, 0.6),        # the LLaVA-Video-7B layout: top-k branch, 53 slots per workgroup
    (64, 210, 3584, 14, 20, 0.03, 0.3, 0.6),       # long runs (up to 63 members): runs cross many segment boundaries
    (64, 195, 3584, 15, 12, 0.5, 0.3, 0.6),        # the Qwen2-VL-7B layout: threshold branch first, then order-maintained calls
    (64, 180, 3584, 0, 0, 0.95, 0.3, 0.6),         # almost nothing folds
    (24, 576, 4096, 3, 5, 0.3, 0.3, 0.7),          # 8 column tiles (every wave holds rows)
    (40, 130, 3000, 300, 500, 0.4, 0.3, 0.6),      # rows of 6000 bytes (ragged last tile) + 800 non-visual rows
    (16, 576, 200, 9000, 9100, 0.3, 0.3, 0.6),     # 27 316 positions: two position words per thread; rows of 400 bytes
    (5, 40, 64, 1, 2, 0.3, 0.3, 0.6),              # less than one slot per workgroup
    (64, 224, 1024, 0, 0, 0.3, 0.5, 0.8),          # 14 336 slots = 56 per workgroup: every row place taken; two tiles
]


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}+{s[3]}+{s[4]}-{s[5]}")
def test_one_launch_equals_three_launches(shape, compact):
    F, P, d, pre, post, p_change, cost, thr = shape
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.4, seed=F * P + d, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    fa, fb = pair(cost, thr, 0.1, compact)
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    hd, pe = dev(h), dev(rotary_tables(L, 32, torch.bfloat16))
    layer = 0
    while not fa.finish_merging and layer < 6:
        hd, pe = same_call(fa, fb, hd, pe)
        hd = dev(harness.layer_stub(hd.cpu(), layer))
        layer += 1
    assert layer >= 1


def test_gaussian_rows_give_the_same_bits_too():
    """Not on the dyadic grid: the two paths sum the norms and dots in different orders (by wave and tile here, by tile and lane
    there), so their similarities may differ in the last place - on slots where they do not, every decision and every row agrees."""
    F, P, d, pre, post = 64, 210, 3584, 14, 20
    h, pt = video_tokens(F, P, d, p_change=0.2, seed=1234, pre=pre, post=post)
    L = h.shape[1]
    fa, fb = pair()
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    pe = dev(rotary_tables(L, 128, torch.bfloat16))
    oa, _, _ = fa(dev(h), [t.clone() for t in pe], None)
    ob, _, _ = fb(dev(h), [t.clone() for t in pe], None)
    sa, sb = fa.last_plan()["sim"].float().cpu(), fb.last_plan()["sim"].float().cpu()
    differ = (sa != sb).float().mean().item()
    assert differ < 2e-3, differ
    assert ((sa - sb).abs() <= 0.0079).all()                     # one bf16 ulp below 1
    if differ == 0.0:
        assert fa.last_call["L_out"] == fb.last_call["L_out"] and same_bits(oa.cpu(), ob.cpu())


def test_outputs_sized_for_the_other_branch_and_no_outputs_at_all():
    """Exactly sized outputs are allocated for the top-k branch's length; when the plan takes the threshold branch the kernel - which
    published l_out right behind its grid barrier - waits, rows in hand, for outputs of that length in the second mail slot (no
    second launch); likewise when no length can be guessed."""
    F, P, d, pre, post = 64, 195, 3584, 15, 12
    h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.8, seed=77, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    pe = dev(rotary_tables(L, 128, torch.bfloat16, mrope=True))
    for rep in range(2):            # (the first pass may find the caching allocator cold: a hipMalloc of 46 MB behind the result
        fa, fb = pair()             #  block can take longer than the kernel waits - it then gets the merge kernel instead, slot 3)
        for ff in (fa, fb):
            prepare(ff, dev(pt), P, pre, F * P, L)
        same_call(fa, fb, dev(h), pe)
        assert fa.last_call["branch"] == 0 and fa.last_call["applied"] == 2 and fa.last_call["mail_slot"] in (2, 3)
    assert fa.last_call["branch"] == 0 and fa.last_call["applied"] == 2 and fa.last_call["mail_slot"] == 2, {k: fa.last_call[k] for k in ("branch", "applied", "mail_slot", "flow", "wait_ns")}    # guessed the top-k length, got the threshold set
    # no guess: prepare() scalars that do not describe whole frames -> K0 path for the first call (three launches), the second
    # call is order-maintained and goes out as one launch without a length to allocate for
    fa, fb = pair()
    for ff in (fa, fb):
        ff.prepare(dev(pt), P, pre, pre + F * P - 1, F * P - 1, L)                # (no layout hint: the length is not F * P)
    o, q = same_call(fa, fb, dev(h), pe, expect_one_launch=False)
    o2 = dev(harness.layer_stub(o.cpu(), 0))
    same_call(fa, fb, o2, q)
    assert fa.last_call["one_launch"]
    assert fa.last_call["L_out"] == o2.shape[1] or fa.last_call["mail_slot"] == 2


def test_a_host_that_answers_too_late_gets_the_merge_kernel_as_a_launch_of_its_own(monkeypatch):
    """The waiting kernel gives up after ~4 ms (FF_STAT_ACK = 4 seq + 3), ff_ctx_merge_apply sees that and enqueues the merge kernel
    behind it: same bits, one launch more."""
    import time
    F, P, d, pre, post = 64, 195, 3584, 15, 12
    h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.8, seed=77, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    fa, fb = pair()
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    pe = dev(rotary_tables(L, 128, torch.bfloat16, mrope=True))
    real = ffa.FrameFusion._merge_outputs
    calls = []

    def slow(self, st, L_cap=None):
        calls.append(L_cap)
        if len(calls) == 2 and self is fa:                   # (the allocation BEHIND the result: the kernel is waiting)
            time.sleep(0.012)
        return real(self, st, L_cap)
    monkeypatch.setattr(ffa.FrameFusion, "_merge_outputs", slow)
    same_call(fa, fb, dev(h), pe)
    assert fa.last_call["applied"] == 2 and fa.last_call["mail_slot"] == 3
    monkeypatch.setattr(ffa.FrameFusion, "_merge_outputs", real)
    # ... and the context goes on as if nothing had happened
    h2, _ = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=5, pre=pre, post=post, grid=0.125)
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    same_call(fa, fb, dev(h2), pe)
    assert fa.last_call["one_launch"] and fa.last_call["mail_slot"] in (1, 2)


def test_top_k_guess_holds_and_the_call_is_one_kernel():
    F, P, d, pre, post = 64, 210, 3584, 14, 20
    h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1234, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    fa, fb = pair()
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    pe = dev(rotary_tables(L, 128, torch.bfloat16))
    out, _ = same_call(fa, fb, dev(h), pe)
    assert fa.last_call["branch"] == 1 and fa.last_call["applied"] == 2 and fa.last_call["mail_slot"] == 1    # the guessed outputs were taken
    assert out.untyped_storage().nbytes() == out.numel() * out.element_size()    # exactly sized, no capacity buffer behind it


def test_what_does_not_fit_takes_the_three_launches():
    cases = [(torch.float16, 8, 16, 64), (torch.float32, 8, 16, 64), (torch.bfloat16, 64, 576, 4096), (torch.bfloat16, 8, 16, 8192)]
    for dt, F, P, d in cases:
        h, pt = video_tokens(F, P, d, p_change=0.3, seed=3, dtype=dt, grid=0.125)
        L = h.shape[1]
        ff = ffa.FrameFusion()
        prepare(ff, dev(pt), P, 0, F * P, L)
        ff(dev(h), dev(rotary_tables(L, 32, dt)), None)
        assert not ff.last_call["one_launch"], (dt, F, P, d)
    # the residual form (rows = hidden + residual) is a three-launch call as well
    h, pt = video_tokens(8, 16, 64, p_change=0.3, seed=3, grid=0.125)
    ff = ffa.FrameFusion()
    prepare(ff, dev(pt), 16, 0, 128, 128)
    ff(dev(h), dev(rotary_tables(128, 32, torch.bfloat16)), None, residual=dev(h))
    assert not ff.last_call["one_launch"]


def test_wrong_layout_hint_is_repeated_through_the_order_kernels():
    """Text inside the visual range: the hint fails on the device (every position is checked by the one-launch kernel as by K1),
    the library repeats the call through K0 and the three launches."""
    F, P, d, pre, post = 12, 33, 256, 4, 6
    h, pt = video_tokens(F, P, d, p_change=0.3, sigma=0.3, sigma_hi=1.4, seed=5, pre=pre, post=post, grid=0.125)
    pt = pt.clone()
    pt[0, pre + 40] = -1
    pt[0, pre + 77] = -1
    L = h.shape[1]
    fa, fb = pair()
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    pe = dev(rotary_tables(L, 32, torch.bfloat16))
    same_call(fa, fb, dev(h), pe)
    assert fa.last_call["unhinted"] and fb.last_call["unhinted"]


def test_two_one_launch_kernels_side_by_side_give_up_and_still_answer():
    """Two instances on two streams from two threads: each kernel needs every CU until its grid barrier, so two of them can wait
    for each other; one leaves after ~2 ms, the library repeats its call through the three launches.  Whatever happens inside,
    every call returns the bits of the serial run."""
    F, P, d, pre, post = 64, 210, 3584, 14, 20
    h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=99, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    hd, ptd = dev(h), dev(pt)
    pe = dev(rotary_tables(L, 128, torch.bfloat16))
    ref = ffa.FrameFusion()
    ref.one_launch = False
    prepare(ref, ptd, P, pre, F * P, L)
    want, _, _ = ref(hd, [t.clone() for t in pe], None)
    want = want.cpu()
    errors, gave_up = [], [0, 0]

    def worker(x):
        try:
            stream = torch.cuda.Stream()
            ff = ffa.FrameFusion()
            with torch.cuda.stream(stream):
                for _ in range(40):
                    prepare(ff, ptd, P, pre, F * P, L)
                    out, _, _ = ff(hd, [t.clone() for t in pe], None)
                    if not ff.last_call["one_launch"] or int(ff.last_call["scratch"].ctx.res_off) > 0:
                        gave_up[x] += 1
                    if not same_bits(out.cpu(), want):
                        errors.append("wrong bits")
        except Exception as e:                                     # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(x,)) for x in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_barrier_words_survive_many_calls_of_changing_size():
    """The barrier and reader counts are reset by their last arriver: hundreds of launches of different grids' worth of work on
    ONE context, never a memset in between, every result against the three launches."""
    fa, fb = pair(compact=False)
    gen = torch.Generator().manual_seed(1)
    for it in range(60):
        F = int(torch.randint(2, 40, (1,), generator=gen))
        P = int(torch.randint(32, 200, (1,), generator=gen))
        d = 8 * int(torch.randint(1, 64, (1,), generator=gen))
        pre, post = int(torch.randint(0, 30, (1,), generator=gen)), int(torch.randint(0, 30, (1,), generator=gen))
        h, pt = video_tokens(F, P, d, p_change=0.3, sigma=0.3, sigma_hi=1.4, seed=it, pre=pre, post=post, grid=0.125)
        L = h.shape[1]
        for ff in (fa, fb):
            prepare(ff, dev(pt), P, pre, F * P, L)
        same_call(fa, fb, dev(h), dev(torch.arange(L)[None]))
    assert not fa.last_call["scratch"].dirty


def test_a_guess_that_failed_for_a_call_index_is_not_repeated_in_the_next_prefill():
    """Three launches, exactly sized outputs, threshold branch: the first prefill guesses the top-k length for its first merge call,
    the blind merge kernel writes nothing (applied = 0) and is repeated; the next prefill of the instance sends the same call index
    through plan -> wait -> outputs -> merge kernel (no wasted launch) - same bits either way, and a call whose guess held keeps
    guessing."""
    F, P, d, pre, post = 16, 48, 512, 3, 5
    h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.8, seed=9, pre=pre, post=post, grid=0.125, dtype=torch.float16)
    L = h.shape[1]
    ff = ffa.FrameFusion()
    pe = dev(rotary_tables(L, 64, torch.float16))
    outs = []
    for rep in range(3):
        prepare(ff, dev(pt), P, pre, F * P, L)
        o, _, _ = ff(dev(h), [t.clone() for t in pe], None)
        assert not ff.last_call["one_launch"] and ff.last_call["branch"] == 0 and ff.last_call["L_out"] < L
        assert ff.last_call["flow"] == ("submit" if rep == 0 else "wait"), (rep, ff.last_call["flow"])
        outs.append(o.cpu())
    assert same_bits(outs[0], outs[1]) and same_bits(outs[0], outs[2])
    # the top-k branch on the same instance: index 0 is distrusted once, found right, trusted again
    h2, _ = video_tokens(F, P, d, p_change=0.1, sigma=0.3, seed=10, pre=pre, post=post, grid=0.125, dtype=torch.float16)
    flows = []
    for rep in range(3):
        prepare(ff, dev(pt), P, pre, F * P, L)
        ff(dev(h2), [t.clone() for t in pe], None)
        assert ff.last_call["branch"] == 1
        flows.append(ff.last_call["flow"])
    assert flows == ["wait", "submit", "submit"], flows


@pytest.mark.parametrize("case", ["k_exceeds_the_real_similarities", "threshold_below_ignore_token", "all_rows_equal", "nan_rows"])
def test_edge_cases_of_the_early_result(case):
    """The result block leaves before the plan has run: l_out = L - k + [k > real similarities] on the top-k branch (slot 0, the first
    of the IGNORE_TOKEN ties, never folds), L - count - [threshold below IGNORE_TOKEN] on the threshold branch.  The corners of that
    arithmetic, against the three launches (which count the members they actually take) - a disagreement would also trip the
    kernel's own assertion (FF_ERR_BIT_RESIDENT on the next call)."""
    cost, thr = 0.3, 0.6
    if case == "k_exceeds_the_real_similarities":
        F, P, d = 2, 64, 256                     # 64 real similarities, k = int(0.7 * 128) = 89: 25 IGNORE_TOKEN slots are taken
        h, pt = video_tokens(F, P, d, p_change=0.0, sigma=0.05, seed=4, pre=2, post=3, grid=0.125)
    elif case == "threshold_below_ignore_token":
        F, P, d = 6, 48, 256
        cost, thr = 0.9, -3.0                    # every slot is in the threshold set; budget large enough for the threshold branch
        h, pt = video_tokens(F, P, d, p_change=0.5, sigma=0.3, seed=5, pre=1, post=1, grid=0.125)
    elif case == "all_rows_equal":
        F, P, d = 8, 40, 256                     # every real similarity is exactly 1: one tie class, cut by position
        h, pt = video_tokens(F, P, d, p_change=0.0, sigma=0.0, seed=6, pre=0, post=0, grid=0.125)
        h[:] = h[:, :1]
    else:
        F, P, d = 8, 40, 256
        h, pt = video_tokens(F, P, d, p_change=0.3, sigma=0.3, seed=7, pre=3, post=2, grid=0.125)
        h[0, 3 + 5 * P + 7, 11] = float("nan")   # two similarities become NaN (largest key: taken first by top-k, never by threshold)
    L = h.shape[1]
    fa, fb = pair(cost, thr, 0.1)
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, int((pt[0] == -1).long().argmin()), F * P, L)
    pe = dev(rotary_tables(L, 32, torch.bfloat16))
    pa, pb = [t.clone() for t in pe], [t.clone() for t in pe]
    oa, _, _ = fa(dev(h), pa, None)
    ob, _, _ = fb(dev(h), pb, None)
    assert fa.last_call["one_launch"] and not fb.last_call["one_launch"]
    for key in ("L_out", "count", "branch", "k", "nv", "ftn"):
        assert fa.last_call[key] == fb.last_call[key], (case, key, fa.last_call[key], fb.last_call[key])
    assert oa.shape == ob.shape
    if case != "nan_rows":
        assert same_bits(oa.cpu(), ob.cpu())
    else:
        a, b = oa.cpu().view(torch.int16), ob.cpu().view(torch.int16)
        nan = torch.isnan(oa.cpu().float()) | torch.isnan(ob.cpu().float())
        assert torch.equal(a[~nan], b[~nan]) and torch.equal(torch.isnan(oa.cpu().float()), torch.isnan(ob.cpu().float()))
    assert torch.equal(fa.patch_type, fb.patch_type)
    # ... and the kernel's own assertion stayed quiet: the next call of the instance goes through
    prepare(fa, dev(pt), P, int((pt[0] == -1).long().argmin()), F * P, L)
    fa(dev(h), [t.clone() for t in pe], None)


@pytest.mark.parametrize("shape", [(64, 210, 3584, 14, 20, 0.5), (64, 195, 3584, 15, 12, 0.6), (24, 576, 4096, 3, 5, 0.5), (40, 130, 3000, 300, 500, 0.5),
                                   (64, 224, 1024, 0, 0, 0.6)],
                         ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}")
@pytest.mark.parametrize("mrope", [False, True])
def test_call_b_with_the_residual_add_fused_in(shape, mrope):
    """Call A (no residual, layout hint) and then call B of every layer with the decoder's residual add fused in (rows = T(hidden +
    residual), modeling_qwen2.py:64-67): call B comes with the order call A left behind and goes out as the one-launch kernel whose
    rows are sums - two requests per row, through registers - against the three launches, bit for bit."""
    F, P, d, pre, post, p_change = shape
    h, pt = video_tokens(F, P, d, p_change=p_change, sigma=0.3, sigma_hi=1.6, seed=F + P + d, pre=pre, post=post, grid=0.125)
    L = h.shape[1]
    fa, fb = pair(0.3, 0.6, 0.02)
    for ff in (fa, fb):
        prepare(ff, dev(pt), P, pre, F * P, L)
    hd, pe = dev(h), dev(rotary_tables(L, 64, torch.bfloat16, mrope=mrope))
    hd, pe = same_call(fa, fb, hd, pe)                                   # call A
    layer, fused = 0, 0
    gen = torch.Generator().manual_seed(11)
    while not fa.finish_merging and layer < 5:
        attn_out = dev((torch.randint(-8, 9, tuple(hd.shape), generator=gen).float() * 0.125).to(torch.bfloat16))
        before = hd.shape[1]
        hd, pe = same_call(fa, fb, attn_out, pe, residual=hd)            # call B: T(attn_out + residual)
        fused += 1
        assert fa.last_call["L_in"] == before
        hd = dev(harness.layer_stub(hd.cpu(), layer))
        layer += 1
    assert fused >= 1
