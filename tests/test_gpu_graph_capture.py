"""GPU: ff_merge_step - the context-free, non-waiting form of the merge call (K1, plan, K4 from ONE host call) - is what the
header calls the capturable form: captured into a hipGraph (torch.cuda.CUDAGraph on the capture stream), replayed on NEW data in
the same buffers, and compared with FrameFusion.forward on that data, bit for bit."""
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_merge_step_is_captured_and_replayed():
    F, P, d = 16, 40, 256
    lib = _lib.load()
    h0, pt = video_tokens(F, P, d, p_change=0.3, sigma=0.3, sigma_hi=1.4, seed=1, grid=0.125, device=DEV)
    h1, _ = video_tokens(F, P, d, p_change=0.5, sigma=0.3, sigma_hi=1.4, seed=2, grid=0.125, device=DEV)
    L = h0.shape[1]
    hidden = h0.clone()
    out = torch.empty(1, L, d, dtype=hidden.dtype, device=DEV)
    ptype_out = torch.empty(1, L, dtype=torch.int64, device=DEV)
    i32 = lambda: torch.empty(L, dtype=torch.int32, device=DEV)
    u8 = lambda: torch.empty(L, dtype=torch.uint8, device=DEV)
    order, inv, order_next, inv_next, dst = i32(), i32(), i32(), i32(), i32()
    member, keep = u8(), u8()
    sim = torch.empty(L, dtype=torch.float32, device=DEV)
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=DEV)
    wsb = int(lib.ff_workspace_bytes(L, P))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    aux = (_lib.FFAux * _lib.MAX_AUX)()
    aux[0] = _lib.FFAux(pt.data_ptr(), ptype_out.data_ptr(), 8, 1)
    thr = float(torch.tensor(0.6, dtype=torch.bfloat16))
    sub = ffa.FrameFusion._compute_pruning_ratio([], 0.3)

    def step():
        _lib.check(lib.ff_merge_step(hidden.data_ptr(), None, out.data_ptr(), _lib.FF_BF16, L, d, L, pt.data_ptr(), P, 0, thr, sub, 0.1,
                                     order.data_ptr(), inv.data_ptr(), sim.data_ptr(), member.data_ptr(), dst.data_ptr(),
                                     keep.data_ptr(), stats.data_ptr(), None, 1, aux, 1, 0, F, order_next.data_ptr(),
                                     inv_next.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()), "ff_merge_step")

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()                                             # (warm-up outside the capture: module loading, attributes)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for data in (h1, h0, h1):
        hidden.copy_(data)
        g.replay()
        torch.cuda.synchronize()
        ff = ffa.FrameFusion(0.3, 0.6, 0.1)
        ff.prepare(pt, P, 0, L, L, L)
        want, _, _ = ff(data.clone(), rotary_tables(L, 16, torch.bfloat16, device=DEV), None)
        l_out = int(stats[_lib.STAT_LOUT])
        assert l_out == want.shape[1] and int(stats[_lib.STAT_ERROR]) == 0
        assert same_bits(out[:, :l_out].cpu(), want.cpu())
        assert torch.equal(ptype_out[0, :l_out], ff.patch_type[0])
    assert not ws[:16 * 260 * 4].any()                     # the workspace protocol holds under replay: the level-0 select table is clean again
