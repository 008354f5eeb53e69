#!/usr/bin/env python
"""Where the host's time goes inside one default-instance merge call, in a back-to-back loop (the state `us_back_to_back`
is measured in): every crossing of the C ABI and every host stage of FrameFusion.forward is stamped (enter / exit, ns), and the
mean timeline over `--calls` calls is printed relative to the start of prepare().

    python tools/flow_stamps.py [--config 7b] [--calls 400] [--three]
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd import _lib, main as ffmain
from framefusion_amd.synth import video_tokens, rotary_tables
from trace_config import CONFIGS

LOG = []
now = time.perf_counter_ns


class LibProxy:
    def __init__(self, lib):
        self._lib = lib
        self._cache = {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        def wrapped(*a):
            t0 = now(); r = fn(*a); LOG.append((name, t0, now())); return r
        self.__dict__[name] = wrapped
        return wrapped


def wrap_method(cls, name):
    f = getattr(cls, name)
    def wrapped(self, *a, **kw):
        t0 = now(); r = f(self, *a, **kw); LOG.append((name, t0, now())); return r
    setattr(cls, name, wrapped)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b")
    ap.add_argument("--calls", type=int, default=400)
    ap.add_argument("--three", action="store_true")
    ap.add_argument("--views", action="store_true")
    ap.add_argument("--wg", action="store_true")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    dev = "cuda:0"
    h0, pt = video_tokens(c["F"], c["P"], c["d"], p_change=c["p_change"], sigma=0.25, sigma_hi=c["sigma_hi"], seed=c["seed"], pre=c["pre"],
                          post=c["post"], dtype=torch.bfloat16, device=dev)
    L = h0.shape[1]
    pe = rotary_tables(L, 128, torch.bfloat16, device=dev, mrope=c["mrope"])
    ffa.FrameFusion.one_launch = not a.three
    ff = ffa.FrameFusion(0.3, c["thr"], 0.1, **(dict(compact_outputs=False) if a.views else {}))
    F, P, pre = c["F"], c["P"], c["pre"]

    def call():
        ff.prepare(pt, P, pre, pre + F * P - 1, F * P, L)
        return ff(h0, list(pe), None)
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.calls):
        call()
    torch.cuda.synchronize()
    print(f"plain loop: {(time.perf_counter() - t0) / a.calls * 1e6:.1f} us per prepare+forward, L={L} -> {ff.last_call['L_out']}, "
          f"one_launch={ff.last_call.get('one_launch')} applied={ff.last_call.get('applied')}")
    if ff.last_call.get("one_launch"):
        st = ff.last_call["scratch"].stats.cpu().tolist()
        names = ["rows+sims", "barrier", "decision", "plan", "fold", "roles", "end"]
        subs = ["keymasks", "tieslot", "poswords", "published", "ldsrows", "vgprrows", "contin", "outputs"]
        print("  last kernel of the loop, workgroup 0 (us from its start): " +
              "  ".join(f"{n} {st[_lib.STAT_T_PLAN + x] / 100:.1f}" for x, n in enumerate(names)) + " | " +
              "  ".join(f"{n} {st[_lib.STAT_T_ORDER + x] / 100:.1f}" for x, n in enumerate(subs)), "mail slot", ff.last_call["mail_slot"])
    if a.wg:
        # a library built with EXTRA=-DFF_RES_WGSTAMPS: earliest / latest workgroup start and end of the kernels of a back-to-back
        # loop (device clock, 100 MHz), and the host's clock around the launch
        sc = ff.last_call["scratch"]
        dbg = sc.ws[20800 + 3072: 20800 + 3072 + 96].view(torch.int64)
        M = (1 << 64) - 1
        rows = []
        for _ in range(12):
            for _ in range(20):
                call()
            dbg.zero_()
            call()
            torch.cuda.synchronize()
            v = [x & M for x in dbg.tolist()]
            s_min, s_max, e_min, e_max, s0, e0 = M - v[0], v[1], M - v[2], v[3], v[4], v[5]
            rows.append(((s_max - s_min) / 100, (e_min - s_min) / 100, (e_max - s_min) / 100, (s0 - s_min) / 100, (e0 - s_min) / 100, (v[6] - s_min) / 100, (v[7] - s_min) / 100, (v[8] - s_min) / 100, v[9] / 100, v[10] / 100, v[11] / 100))
        for r in rows:
            print("  workgroup starts spread over %.1f us; first end %.1f, last end %.1f us after the first start; workgroup 0: start %.1f end %.1f; its spare wave: roles from %.1f, index roles done %.1f, auxiliary rows done %.1f; latest wave of the launch: longest wait for the prefetched rows %.1f, longest continuation %.1f, fold done %.1f" % r)
    # stamped loop
    real = _lib.load()
    _lib._lib = LibProxy(real)
    for m in ("prepare", "_merge_prepare", "_merge_outputs", "_no_outputs", "_merge_complete", "_scratch_for",
              "_merge_submitted"):
        wrap_method(ffa.FrameFusion, m)
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    rows = {}
    t_all = time.perf_counter()
    for _ in range(a.calls):
        del LOG[:]
        t0 = now()
        call()
        t1 = now()
        for i, (name, s, e) in enumerate(sorted(LOG, key=lambda x: x[1])):
            r = rows.setdefault((i, name), [0, 0, 0])
            r[0] += s - t0; r[1] += e - t0; r[2] += 1
        r = rows.setdefault((99, "forward returns"), [0, 0, 0]); r[0] += t1 - t0; r[1] += t1 - t0; r[2] += 1
    torch.cuda.synchronize()
    print(f"stamped loop: {(time.perf_counter() - t_all) / a.calls * 1e6:.1f} us per call")
    for (i, name), (s, e, n) in sorted(rows.items()):
        print(f"  {name:28s} enter {s / n / 1e3:7.1f}  exit {e / n / 1e3:7.1f}  ({(e - s) / n / 1e3:6.1f} us)  n={n}")
    _lib._lib = real


if __name__ == "__main__":
    main()
