#!/usr/bin/env python
"""One of bench.py's extra configurations (BASELINE.json configs[1], [2], [4] as the models run them) as a command for
`rocprofv3 --kernel-trace`: warm-up cascades, an idle gap, then `--reps` whole prefill cascades back to back (the state
`us_back_to_back` of the bench line is measured in).  tools/timeline_cascade.py turns the trace into a per-kernel timeline.

    python tools/trace_config.py --config 7b|c3|c5|c5topk|c2thr [--reps 20]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import framefusion_amd as ffa

CONFIGS = {   # the argument lists of bench.extra_configs
    "c2thr": dict(F=64, P=576, d=4096, p_change=0.5, thr=0.6, pre=14, post=20, heads=32, kv_heads=8, num=1, mrope=False, sigma_hi=1.6, seed=1234),
    "c3": dict(F=64, P=195, d=3584, p_change=0.5, thr=0.6, pre=15, post=12, heads=28, kv_heads=4, num=4, mrope=True, sigma_hi=1.8, seed=77),
    "c5": dict(F=64, P=576, d=8192, p_change=0.95, thr=0.6, pre=14, post=20, heads=64, kv_heads=8, num=1, mrope=False, sigma_hi=None, seed=1234),
    "c5topk": dict(F=64, P=576, d=8192, p_change=0.2, thr=0.6, pre=14, post=20, heads=64, kv_heads=8, num=1, mrope=False, sigma_hi=None, seed=1234),
    "c2": dict(F=64, P=576, d=4096, p_change=0.2, thr=0.6, pre=0, post=0, heads=32, kv_heads=8, num=1, mrope=False, sigma_hi=None, seed=1234),
    "7b128": dict(F=128, P=210, d=3584, p_change=0.2, thr=0.6, pre=14, post=20, heads=28, kv_heads=4, num=1, mrope=False, sigma_hi=None, seed=1234),
    "7b32": dict(F=32, P=210, d=3584, p_change=0.2, thr=0.6, pre=14, post=20, heads=28, kv_heads=4, num=1, mrope=False, sigma_hi=None, seed=1234),
    "7b": dict(F=64, P=210, d=3584, p_change=0.2, thr=0.6, pre=14, post=20, heads=28, kv_heads=4, num=1, mrope=False, sigma_hi=None, seed=1234),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), required=True)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    c = CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    # bench.cascade: reps + 2 isolated cascades, then 4 * reps back to back - the LAST block of the trace
    r = bench.cascade(ffa, dev, c["F"], c["P"], c["d"], c["p_change"], c["thr"], c["pre"], c["post"], c["heads"], c["kv_heads"], c["num"],
                      c["mrope"], sigma_hi=c["sigma_hi"], reps=max(2, a.reps // 4), seed=c["seed"], idle_before_b2b_s=0.05)
    print(json.dumps({"config": a.config, "back_to_back_cascades": 4 * max(2, a.reps // 4), **r}))


if __name__ == "__main__":
    main()
