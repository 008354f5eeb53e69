#!/usr/bin/env python
"""Does the merge kernel find the rows the similarity kernel just read in the 256 MiB Infinity Cache?
ff_merge_begin (K1) on buffer A, ff_merge_finish (plan + K4) on the SAME buffer vs on a copy B (same bytes, cold).
Run under tools/prof_cmd.sh and compare k_merge_compact (development tool).  argv[1]: same | copy"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens

mode = sys.argv[1] if len(sys.argv) > 1 else "same"
dev = "cuda:0"
lib = _lib.load()
F, P, d = 64, 576, 4096
h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1234, device=dev)
bufs = [h, h.clone(), h.clone(), h.clone()]
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
st = _lib.stream_ptr()
for it in range(60):
    a = bufs[(2 * it) % 4]
    b = a if mode == "same" else bufs[(2 * it + 1) % 4]
    _lib.check(lib.ff_merge_begin(a.data_ptr(), None, _lib.FF_BF16, L, d, pt.data_ptr(), P, 0, thr, order.data_ptr(),
                                  inv.data_ptr(), sim.data_ptr(), stats.data_ptr(), it + 1, 0, F, ws.data_ptr(), wsb, st), "begin")
    _lib.check(lib.ff_merge_finish(b.data_ptr(), None, out.data_ptr(), _lib.FF_BF16, L, d, L, thr, 0.7, 0.1, order.data_ptr(),
                                   inv.data_ptr(), sim.data_ptr(), member.data_ptr(), dst.data_ptr(), keep.data_ptr(),
                                   stats.data_ptr(), None, it + 1, aux, 1, order_next.data_ptr(), inv_next.data_ptr(),
                                   ws.data_ptr(), wsb, st), "finish")
torch.cuda.synchronize()
print(mode, "L_out", int(stats[_lib.STAT_LOUT]))
