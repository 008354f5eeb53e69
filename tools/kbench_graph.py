#!/usr/bin/env python
"""The three launches of a merge call (ff_merge_step: K1, plan, K4) issued directly against the same three
launches replayed from a captured hipGraph (torch.cuda.CUDAGraph capture of the C-ABI call on the capture
stream).  64 x 576 x 4096 bf16 and the 7B shape 64 x 210 x 3584; hipEvent time per call, stream kept busy."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens

dev = "cuda:0"
lib = _lib.load()


def run(F, P, d):
    h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1234, device=dev)
    L = h.shape[1]
    out = torch.empty(1, L, d, dtype=h.dtype, device=dev)
    ptype_out = torch.empty(1, L, dtype=torch.int64, device=dev)
    i32 = lambda: torch.empty(L, dtype=torch.int32, device=dev)
    u8 = lambda: torch.empty(L, dtype=torch.uint8, device=dev)
    order, inv, order_next, inv_next, dst = i32(), i32(), i32(), i32(), i32()
    member, keep = u8(), u8()
    sim = torch.empty(L, dtype=torch.float32, device=dev)
    stats = torch.zeros(_lib.STAT_WORDS, dtype=torch.int64, device=dev)
    wsb = int(lib.ff_workspace_bytes(L, P))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    aux = (_lib.FFAux * _lib.MAX_AUX)()
    aux[0] = _lib.FFAux(pt.data_ptr(), ptype_out.data_ptr(), 8, 1)
    thr = float(torch.tensor(0.6, dtype=torch.bfloat16))

    def step(stream):
        rc = lib.ff_merge_step(h.data_ptr(), None, out.data_ptr(), _lib.FF_BF16, L, d, L, pt.data_ptr(), P, 0, thr, 0.7, 0.1,
                               order.data_ptr(), inv.data_ptr(), sim.data_ptr(), member.data_ptr(), dst.data_ptr(),
                               keep.data_ptr(), stats.data_ptr(), None, 1, aux, 1, 0, F, order_next.data_ptr(),
                               inv_next.data_ptr(), ws.data_ptr(), wsb, stream)
        _lib.check(rc, "ff_merge_step")

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    direct = timeit(lambda: step(_lib.stream_ptr()))
    l_direct = int(stats[_lib.STAT_LOUT])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(_lib.stream_ptr())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step(_lib.stream_ptr())
    torch.cuda.synchronize()
    graph = timeit(g.replay)
    assert int(stats[_lib.STAT_LOUT]) == l_direct
    print(f"{F} x {P} x {d}: {L} -> {l_direct}   direct launches {direct:7.1f} us/call   hipGraph replay {graph:7.1f} us/call "
          f"({graph - direct:+.1f} us)")


run(64, 576, 4096)
run(64, 210, 3584)
run(16, 210, 3584)
