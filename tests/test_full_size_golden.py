"""Parity at BASELINE.json's stated sizes against the REAL reference (tests/golden/full.npz, written by
oracle/make_golden_full.py from /root/reference on gaussian inputs): C2 (64 x 576 x 4096, bench.py's own
workload to the seed, three regimes), C3 (Qwen2-VL-7B: 64 grids x {180, 195} x 3584), C5 (LLaVA-Video-72B:
64 x 576 x 8192), the real LLaVA-Video-7B layout (64 x 210 x 3584) and three prune calls fed by
utils.scaled_dot_product_attention (num = 1 and 4, GQA).

What "identical" means on un-quantised data:
  * similarities / importances are sums in another order: <= 1e-3 of them may differ, by 1-3 ulps (the
    observed rate is printed and bounded);
  * the merge / keep decision of every slot whose value is bit-identical AND is not in the tie class of
    the k-th value must be identical to the reference's;
  * inside the tie class torch.topk on CPU takes ARBITRARY members (the fixtures record that the
    reference's choice is not the lowest-index one); the number taken must match, the build takes the
    lowest indices (what torch.topk does on the MI355X: tests/test_gpu_topk_ties.py);
  * output rows whose run is the same on both sides are compared bit for bit (north-star bar: 1e-3 rel).
CPU (-m "not gpu"): the oracle against these vectors.  GPU: the HIP path.
"""
import hashlib

import numpy as np
import pytest
import torch

from framefusion_amd.synth import video_tokens
from oracle import ff_oracle as orc
from tests.conftest import Golden, from_bits, to_bits

DEV = "cuda:0"
MERGE = ["c2_topk", "c2_thr", "c2_low", "c3_p195", "c3_p180", "c5_topk", "c5_low", "llava7b"]
PRUNE = ["c5_prune", "c3_prune", "c2_prune"]
CPU_MERGE = ["c2_topk", "c2_thr", "c3_p195", "llava7b"]          # the d = 8192 cases run on the GPU only (time)


def merge_inputs(g, name):
    F, P, D, pre, post, seed = (int(v) for v in g[f"{name}/cfg"])
    p_change, sigma_hi, cost, thr, lb = (float(v) for v in g[f"{name}/fcfg"])
    h, pt = video_tokens(F, P, D, p_change=p_change, sigma=0.3, sigma_hi=None if sigma_hi < 0 else sigma_hi, seed=seed,
                         pre=pre, post=post, dtype=torch.bfloat16)
    return h, pt, F, P, pre, (cost, thr, lb)


def ulps(a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
    """distance in representable values between two bf16 bit arrays (same-sign finite values)"""
    def lin(x):
        x = x.astype(np.int32) & 0xffff
        return np.where(x & 0x8000, -(x & 0x7fff), x)
    return np.abs(lin(a_bits) - lin(b_bits))


def kth_value(values: torch.Tensor, k: int) -> float:
    v = values.float()
    v = torch.where(torch.isnan(v), torch.full_like(v, float("inf")), v)
    return float(torch.sort(v, descending=True).values[k - 1])


def check_merge(g, name, sim, member, kept, out_rows, order, flags, sparsity, label):
    """sim [Nv] bf16 (cpu), member [Nv] bool by by-patch slot, kept [L] bool, out_rows(r) -> rows of the
    output, order [Nv] long."""
    L, L_out, nv, ftn, count, branch, k, fm, fp = (int(v) for v in g[f"{name}/stats"])
    sim_ref_bits = g[f"{name}/sim"]
    sim_bits = to_bits(sim)
    diff = sim_bits != sim_ref_bits
    flips = int(diff.sum())
    print(f"[{label}] {name}: {flips} of {nv} similarities differ from the reference's "
          f"({flips / nv:.1e}), max {int(ulps(sim_bits, sim_ref_bits).max())} ulp")
    assert flips <= 1e-3 * nv
    assert int(ulps(sim_bits[diff], sim_ref_bits[diff]).max(initial=0)) <= 3
    assert list(flags) == [fm, fp]
    merged_ref = np.unpackbits(g[f"{name}/merged"])[:nv].astype(bool)
    kept_ref = np.unpackbits(g[f"{name}/kept"])[:L].astype(bool)
    member = member.numpy().astype(bool)
    unsure = diff.copy()
    if branch:
        assert sparsity == []
        assert int(member.sum()) == k == int(merged_ref.sum())                 # exactly k slots folded, both sides
        kb, tie_size, tie_taken, ref_lowest = (int(v) for v in g[f"{name}/tie"])
        sim_ref = from_bits(sim_ref_bits.copy(), torch.bfloat16).float().numpy()
        kth_ref = float(from_bits(np.asarray([kb], dtype=np.uint16).view(np.int16), torch.bfloat16).float()[0])
        kth_mine = kth_value(sim, k)
        simf = sim.float().numpy()
        tie = (sim_ref == kth_ref) | (simf == kth_mine)
        unsure |= tie
        # the number of tie members taken can move by one per flipped similarity
        assert abs(int((member & tie).sum()) - int((merged_ref & tie).sum())) <= flips
        if flips == 0:
            assert int((member & tie).sum()) == tie_taken and int(tie.sum()) == tie_size
        # the build's own rule: strictly greater all taken, ties in ascending by-patch index
        want = np.zeros(nv, dtype=bool)
        want[orc.topk_lowest_index(sim, k).numpy()] = True
        want[0] = False
        assert np.array_equal(member, want)
        assert ref_lowest == 0          # documents that the CPU reference itself does NOT take the lowest indices
    else:
        assert len(sparsity) == 1 and abs(sparsity[0] - float(g[f"{name}/sparsity"][0])) <= (flips + 0.5) / ftn
    assert np.array_equal(member[~unsure], merged_ref[~unsure])                # every decidable slot: same decision
    # kept mask by position: implied by the member flags
    kept = kept.numpy().astype(bool)
    pos_unsure = np.zeros(L, dtype=bool)
    pos_unsure[order.numpy()[unsure]] = True
    assert np.array_equal(kept[~pos_unsure], kept_ref[~pos_unsure])
    assert int(kept.sum()) == L_out or flips > 0 and abs(int(kept.sum()) - L_out) <= flips
    # sampled anchors: same run => same bits
    inv = np.full(L, -1, dtype=np.int64)
    inv[order.numpy()] = np.arange(nv)
    out_index = np.cumsum(kept) - 1
    comparable = 0
    for p, n_ref, row_ref in zip(g[f"{name}/rows_pos"], g[f"{name}/rows_run"], g[f"{name}/rows"]):
        if not kept[p]:
            continue
        n = 0
        if inv[p] >= 0:
            j = inv[p] + 1
            while j < nv and member[j]:
                n += 1
                j += 1
        if n != int(n_ref):
            continue
        comparable += 1
        row = to_bits(out_rows(int(out_index[p])))
        assert np.array_equal(row, row_ref), (name, int(p), n)
    assert comparable >= len(g[f"{name}/rows_pos"]) // 2
    return flips, bool(np.array_equal(kept, kept_ref))


# ------------------------------------------------------------------------------------------------------
# CPU: the oracle against the reference's vectors
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CPU_MERGE)
def test_oracle_matches_reference_full_size_merge(name):
    g = Golden("full")
    h, pt, F, P, pre, params = merge_inputs(g, name)
    L = h.shape[1]
    o = orc.OracleFrameFusion(*params)
    o.prepare(pt.clone(), P, pre, pre + F * P - 1, F * P, L)
    ho, po, _ = o.forward(h, torch.arange(L)[None], None)
    kept = torch.zeros(L, dtype=torch.bool)
    kept[po[0]] = True
    member = torch.zeros(o.last_sim.shape[1], dtype=torch.bool)
    member[o.last_merge_idx] = True
    member[0] = False
    flips, same_kept = check_merge(g, name, o.last_sim[0], member, kept, lambda r: ho[0, r], o.last_order[0],
                                   (int(o.finish_merging), int(o.finish_pruning)), o.sparsity_list, "oracle")
    if same_kept and flips == 0:
        assert hashlib.sha256(to_bits(ho).tobytes()).digest() == g[f"{name}/sha256"].tobytes()


def prune_inputs(g, name):
    H, Hk, dh, num, S, start, n_img, d, seed, orig_len, img_len = (int(v) for v in g[f"{name}/cfg"])
    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(1, H, num, dh, generator=gen).to(torch.bfloat16)
    kk = torch.randn(1, Hk, S, dh, generator=gen).to(torch.bfloat16)
    hid = torch.randn(1, S, d, generator=gen).to(torch.bfloat16)
    return q, kk, hid, (H, Hk, dh, num, S, start, n_img, d, orig_len, img_len)


def check_prune(g, name, imp, kept, label, w_rows=None):
    S, L_out, start, n_img, k = (int(v) for v in g[f"{name}/stats"])
    ref_bits = g[f"{name}/importance"]
    bits = to_bits(imp)
    diff = bits != ref_bits
    rate = float(diff.mean())
    print(f"[{label}] {name}: {int(diff.sum())} of {S} importances differ from the reference's ({rate:.1e}), "
          f"max {int(ulps(bits, ref_bits).max())} ulp")
    assert int(ulps(bits, ref_bits).max()) <= 1
    if w_rows is not None:
        wb, wr = to_bits(w_rows), g[f"{name}/w_rows"]
        wd = wb != wr
        print(f"[{label}] {name}: weights of two (head, query) rows: {int(wd.sum())} of {wd.size} differ ({wd.mean():.1e})")
        assert int(ulps(wb, wr).max()) <= 1
        assert wd.mean() <= 2e-3
    assert rate <= 5e-4          # measured 0.5 - 2.2e-4: the fp32 summation-order floor
    kept_ref = np.unpackbits(g[f"{name}/kept"])[:S].astype(bool)
    kept = kept.numpy().astype(bool)
    assert int(kept.sum()) == L_out == S - n_img + k
    kb, tie_size, tie_taken, ref_lowest = (int(v) for v in g[f"{name}/tie"])
    imp_ref = from_bits(ref_bits.copy(), torch.bfloat16).float().numpy()
    kth_ref = float(from_bits(np.asarray([kb], dtype=np.uint16).view(np.int16), torch.bfloat16).float()[0])
    vis = slice(start, start + n_img)
    kth_mine = kth_value(imp[vis], k)
    unsure = diff | (imp_ref == kth_ref) | (imp.float().numpy() == kth_mine)
    unsure[:start] = False
    unsure[start + n_img:] = False
    assert np.array_equal(kept[~unsure], kept_ref[~unsure])
    tie = unsure & ~diff
    assert abs(int((kept & tie).sum()) - int((kept_ref & tie).sum())) <= int(diff.sum())
    want = np.ones(S, dtype=bool)
    want[vis] = False
    want[start + orc.topk_lowest_index(imp[vis], k).numpy()] = True
    assert np.array_equal(kept, want)                     # the build's rule on its own importances
    return rate


@pytest.mark.parametrize("name", PRUNE)
def test_oracle_matches_reference_full_size_prune(name):
    g = Golden("full")
    q, kk, hid, (H, Hk, dh, num, S, start, n_img, d, orig_len, img_len) = prune_inputs(g, name)
    w = orc.last_query_attention(q, kk, num=num, is_causal=True, enable_gqa=True)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(torch.zeros(1, S, dtype=torch.long), 1, start, start + n_img, img_len, orig_len, finish_merging=True,
              sparsity_list=list(g[f"{name}/sparsity"]))
    ho, po, _ = o.forward(hid, torch.arange(S)[None], None, w)
    kept = torch.zeros(S, dtype=torch.bool)
    kept[po[0]] = True
    check_prune(g, name, torch.mean(w, dim=(1, 2))[0], kept, "oracle", torch.stack((w[0, 0, 0], w[0, H - 1, num - 1])))
    assert np.array_equal(to_bits(ho), to_bits(hid[:, po[0]]))


# ------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the reference's vectors
# ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", MERGE)
def test_hip_matches_reference_full_size_merge(name):
    import framefusion_amd as ffa
    g = Golden("full")
    h, pt, F, P, pre, params = merge_inputs(g, name)
    L = h.shape[1]
    f = ffa.FrameFusion(*params)
    f.prepare(pt.to(DEV), P, pre, pre + F * P - 1, F * P, L)
    hg, pg, _ = f(h.to(DEV), torch.arange(L, device=DEV)[None], None)
    plan = f.last_plan()
    nv = f.last_call["nv"]
    kept = plan["keep"].bool().cpu()
    assert torch.equal(torch.nonzero(kept).reshape(-1), pg[0].cpu())              # the gathered position ids ARE the kept set
    hg_cpu = hg[0].cpu()
    flips, same_kept = check_merge(g, name, plan["sim"].cpu(), plan["member"][:nv].bool().cpu(), kept,
                                   lambda r: hg_cpu[r], plan["order"].long().cpu(),
                                   (int(f.finish_merging), int(f.finish_pruning)), f.sparsity_list, "hip")
    if same_kept and flips == 0:
        assert hashlib.sha256(to_bits(hg).tobytes()).digest() == g[f"{name}/sha256"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRUNE)
def test_hip_matches_reference_full_size_prune(name):
    """importance computed by the HIP kernel from q / K (fused, [1, 1, 1, S]) -> the prune call."""
    import framefusion_amd as ffa
    g = Golden("full")
    q, kk, hid, (H, Hk, dh, num, S, start, n_img, d, orig_len, img_len) = prune_inputs(g, name)
    qd, kd = q.to(DEV), kk.to(DEV)
    imp = ffa.last_query_importance(qd, kd, num=num, is_causal=True)
    w = ffa.scaled_dot_product_attention(qd, kd, None, num=num, is_causal=True, enable_gqa=True)
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    f.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, img_len, orig_len,
              finish_merging=True, sparsity_list=list(g[f"{name}/sparsity"]))
    hd = hid.to(DEV)
    hg, pg, _ = f(hd, torch.arange(S, device=DEV)[None], None, imp)
    assert f.finish_pruning
    kept = torch.zeros(S, dtype=torch.bool)
    kept[pg[0].cpu()] = True
    check_prune(g, name, imp.reshape(-1).cpu(), kept, "hip",
                torch.stack((w[0, 0, 0], w[0, H - 1, num - 1])).cpu())
    assert torch.equal(hg, hd[:, pg[0]])
    # importance computed WITH the instance: the kernel also fills the select tables of the prune call
    f3 = ffa.FrameFusion(0.3, 0.6, 0.1)
    f3.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, img_len, orig_len,
               finish_merging=True, sparsity_list=list(g[f"{name}/sparsity"]))
    imp3 = ffa.last_query_importance(qd, kd, num=num, is_causal=True, framefusion=f3)
    assert torch.equal(imp3, imp) and getattr(imp3, "_ff_tables", None) is not None
    hg3, pg3, _ = f3(hd, torch.arange(S, device=DEV)[None], None, imp3)
    assert torch.equal(pg3, pg) and torch.equal(hg3, hg)
    # ... and the workspace is clean again: a merge call on the same instance behaves
    assert not f3._scratch[("cuda", 0)].dirty
    # hook + prune as ONE host call (prune_from_qk / the deferred handle the adapters pass): same kept set, same rows
    f4 = ffa.FrameFusion(0.3, 0.6, 0.1)
    f4.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, img_len, orig_len,
               finish_merging=True, sparsity_list=list(g[f"{name}/sparsity"]))
    hg4, pg4, _ = f4.prune_from_qk(hd, torch.arange(S, device=DEV)[None], None, qd, kd, num=num, is_causal=True)
    assert f4.finish_pruning and torch.equal(pg4, pg) and torch.equal(hg4, hg) and not f4._scratch[("cuda", 0)].dirty
    f4.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, img_len, orig_len,
               finish_merging=True, sparsity_list=list(g[f"{name}/sparsity"]))
    handle = ffa.last_query_importance(qd, kd, num=num, is_causal=True, framefusion=f4, defer=True)
    hg5, pg5, _ = f4(hd, torch.arange(S, device=DEV)[None], None, handle)
    assert torch.equal(pg5, pg) and torch.equal(hg5, hg)
    with pytest.raises(ffa.FrameFusionHipError):
        f4.prune_from_qk(hd, torch.arange(S, device=DEV)[None], None, qd, kd, num=num)       # no prune is due any more
    # the unfused form (weights [1, H, num, S] -> head mean inside the prune call) keeps the same tokens
    f2 = ffa.FrameFusion(0.3, 0.6, 0.1)
    f2.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, img_len, orig_len,
               finish_merging=True, sparsity_list=list(g[f"{name}/sparsity"]))
    _, pg2, _ = f2(hd, torch.arange(S, device=DEV)[None], None, w)
    assert torch.equal(pg2, pg)
