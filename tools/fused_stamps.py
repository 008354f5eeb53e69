#!/usr/bin/env python
"""Phase stamps of the fused plan + merge launch (FF_FUSED_DBG bit 2; development tool): 100 MHz wall clock, relative to the
start of plan workgroup 0."""
import os, sys, statistics
os.environ["FF_FUSED"] = "1"
os.environ["FF_FUSED_DBG"] = str(int(os.environ.get("FF_FUSED_DBG", "3")) | 4)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib
from framefusion_amd.synth import video_tokens, rotary_tables
F, P, d = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 576, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = "cuda:0"
h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1234, dtype=torch.bfloat16, device=dev)
L = h.shape[1]; h2 = h.clone()
cos, sin = rotary_tables(L, 128, torch.bfloat16, device=dev)
ff = ffa.FrameFusion(0.3, 0.6, 0.1)
rows = []
lib = _lib.load()
import ctypes
for i in range(60):
    ff.prepare(pt, P, 0, L, L, L)
    out = ff(h2 if i & 1 else h, [cos, sin], None)[0]
    if i >= 20:
        torch.cuda.synchronize()
        sc = ff._scratch[("cuda", 0)]
        st = sc.stats.cpu().tolist()
        t0 = st[16]
        # the debug stamps sit in the workspace's scratch ints (ff_plan.hip: ws_scratch_ints)
        G = (L + 4095) // 4096 + 1
        off = ((4160 + 16 + 1024 + 16 + 2048 + 32 + 2 * G * 256) * 4 + 15) // 16 * 16 // 8
        ws = sc.ws.view(torch.int64).cpu()
        raw = ws[off:off + 3 * 4000]
        buf = raw.double() - float(t0)
        passed, ended, info = buf[0::3], buf[1::3], raw[2::3]
        ok = (passed > 0) & (passed < 1e6) & (ended > 0) & (ended < 1e6)
        passed, ended, info = passed[ok] / 100.0, ended[ok] / 100.0, info[ok]
        last_info = (ended, info >> 32, torch.nonzero(ok).reshape(-1))
        rows.append([(st[17] - t0) / 100.0, (st[18] - t0) / 100.0, float(passed.min()), float(passed.median()), float(passed.max()),
                     float(ended.min()), float(ended.median()), float(ended.max()), int(ok.sum()),
                     int((passed < (st[18] - t0) / 100.0 + 3.0).sum())])
        for _ in range(3):
            ff.prepare(pt, P, 0, L, L, L); ff(h, [cos, sin], None)
ended, length, wg = last_info
for mod, label in ((8, "XCD (grid index mod 8)"), (32, "grid index mod 32")):
    rows_ = []
    for k in range(mod):
        m = (wg % mod) == k
        rows_.append(f"{float(ended[m].mean()):.1f}")
    print(f"mean end time by {label}:", " ".join(rows_))
n_main_ = int(wg.max() + 1) // 2
print("mean end time, column group 0 / 1:", round(float(ended[wg < n_main_].mean()), 1), round(float(ended[wg >= n_main_].mean()), 1))
q8 = [round(float(ended[(wg % n_main_ >= a * n_main_ // 8) & (wg % n_main_ < (a + 1) * n_main_ // 8)].mean()), 1) for a in range(8)]
print("mean end time by eighth of the slot range (bx):", q8)
q = torch.quantile(ended, torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0], dtype=ended.dtype))
print("main wg end percentiles 10/25/50/75/90/95/99/100:", [round(float(x), 1) for x in q])
lf = length.double()
print(f"stream length (slots): mean {float(lf.mean()):.1f} sd {float(lf.std()):.1f} min {int(lf.min())} max {int(lf.max())}; corr(end time, length) = "
      f"{float(torch.corrcoef(torch.stack([ended, lf]))[0, 1]):.2f}")
for lo, hi in ((0, 25), (25, 32), (32, 36), (36, 40), (40, 44), (44, 50), (50, 100)):
    m = (lf >= lo) & (lf < hi)
    if int(m.sum()):
        print(f"  length {lo:2d}..{hi - 1:2d}: {int(m.sum()):5d} workgroups, end median {float(ended[m].median()):6.1f} us  max {float(ended[m].max()):6.1f}")
names = ["last-index plan wg done", "flag written", "main wg past wait: min", "  median", "  max", "main wg end: min", "  median", "  max", "main wgs stamped", "  past wait within 3 us of the flag"]
for k, n in enumerate(names):
    v = [r[k] for r in rows]
    print(f"{n:26s} median {statistics.median(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f}")
