"""GPU: several samples overlapped on per-sample streams give exactly the results of one-by-one calls."""
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd.batch import forward_many
from framefusion_amd.synth import video_tokens, rotary_tables
from tests import harness
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_forward_many_matches_sequential():
    F, P, d = 16, 48, 1024
    samples = [video_tokens(F, P, d, p_change=pc, sigma_hi=1.5, seed=30 + i, pre=3, post=4, grid=0.125)
               for i, pc in enumerate((0.2, 0.5, 0.95, 0.3))]
    L = samples[0][0].shape[1]

    def fresh():
        out = []
        for h, pt in samples:
            f = ffa.FrameFusion(0.3, 0.6, 0.1)
            f.prepare(pt.to(DEV), P, 3, 3 + F * P, F * P, L)
            out.append(f)
        return out
    seq_ffs, many_ffs = fresh(), fresh()
    hs = [h.to(DEV) for h, _ in samples]
    want = [f(h, rotary_tables(L, 64, device=DEV), None) for f, h in zip(seq_ffs, hs)]
    def weights_for(f, h):
        if h.shape[1] > 1 and f.finish_merging and not f.finish_pruning:
            return harness.attention_stub(4, 1, h.shape[1], h.dtype, DEV)
        return None

    for round_ in range(2):                       # second round: merge again / prune / nothing left to do
        if round_ == 0:
            got = forward_many(many_ffs, hs, [rotary_tables(L, 64, device=DEV) for _ in hs])
        else:
            got = forward_many(many_ffs, [g[0] for g in got], [g[1] for g in got], None,
                               [weights_for(f, g[0]) for f, g in zip(many_ffs, got)])
            want = [f(w[0], w[1], None, weights_for(f, w[0])) for f, w in zip(seq_ffs, want)]
        torch.cuda.synchronize()
        for g, w, fa, fb in zip(got, want, many_ffs, seq_ffs):
            assert same_bits(g[0].cpu(), w[0].cpu())
            assert same_bits(g[1][0].cpu().contiguous(), w[1][0].cpu().contiguous())
            assert (fa.finish_merging, fa.finish_pruning, fa.sparsity_list) == (fb.finish_merging, fb.finish_pruning, fb.sparsity_list)
