#!/usr/bin/env python
"""Call A + call B (the decoder's residual add fused in: rows = T(attn_out + residual), modeling_qwen2.py:64-67) of one prefill, back
to back, one-launch kernel against the three launches (same process, alternating): what the adapters in framefusion_amd/models issue.

    python tools/flow_residual.py [--config 7b] [--p-change 0.5] [--reps 200]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from trace_config import CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b")
    ap.add_argument("--p-change", type=float, default=0.5)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    c = CONFIGS[a.config]
    dev = "cuda:0"
    F, P, d, pre = c["F"], c["P"], c["d"], c["pre"]
    h0, pt = video_tokens(F, P, d, p_change=a.p_change, sigma=0.25, sigma_hi=1.6, seed=c["seed"], pre=pre, post=c["post"], dtype=torch.bfloat16, device=dev)
    L = h0.shape[1]
    pe0 = rotary_tables(L, 128, torch.bfloat16, device=dev, mrope=c["mrope"])
    ff = ffa.FrameFusion(0.3, c["thr"], 0.02)
    attn = {}
    calls = []

    def prefill():
        ff.prepare(pt, P, pre, pre + F * P - 1, F * P, L)
        h, pe, _ = ff(h0, list(pe0), None)                                   # call A
        calls[:] = [(ff.last_call["L_in"], ff.last_call["L_out"], ff.last_call["one_launch"])]
        n = h.shape[1]
        if n not in attn:
            attn[n] = torch.randn(1, n, d, device=dev, dtype=torch.bfloat16) * 0.1
        h, pe, _ = ff(attn[n], pe, None, None, residual=h)                    # call B of layer 0
        calls.append((ff.last_call["L_in"], ff.last_call["L_out"], ff.last_call["one_launch"]))
    rows = {True: [], False: []}
    shown = {}
    for _ in range(a.rounds):
        for one in (True, False):
            ffa.FrameFusion.one_launch = one
            for _ in range(30):
                prefill()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                prefill()
            torch.cuda.synchronize()
            rows[one].append(round((time.perf_counter() - t0) / a.reps * 1e6, 1))
            shown[one] = list(calls)
    ffa.FrameFusion.one_launch = True
    print(json.dumps({"config": a.config, "p_change": a.p_change, "calls(one launch: L_in, L_out, one_launch)": shown[True],
                      "calls(three launches)": shown[False], "us_per_prefill_one_launch": rows[True], "us_per_prefill_three_launches": rows[False]}))


if __name__ == "__main__":
    main()
