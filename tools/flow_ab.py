#!/usr/bin/env python
"""A/B of the two forms of a merge call inside whole prefill cascades (bench.cascade, host included): the one-launch kernel
(FrameFusion.one_launch = True, the default) against the three launches, per configuration of bench.extra_configs.

    python tools/flow_ab.py [--configs 7b c3 ...] [--reps 6]
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import framefusion_amd as ffa
from trace_config import CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["7b", "c3", "7b32", "7b128"])
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for name in a.configs:
        c = CONFIGS[name]
        rows = {True: [], False: []}
        for _ in range(a.rounds):
            for one in (True, False):
                ffa.FrameFusion.one_launch = one
                r = bench.cascade(ffa, dev, c["F"], c["P"], c["d"], c["p_change"], c["thr"], c["pre"], c["post"], c["heads"], c["kv_heads"],
                                  c["num"], c["mrope"], sigma_hi=c["sigma_hi"], reps=a.reps, seed=c["seed"])
                rows[one].append((round(r["us"], 1), round(r["us_back_to_back"], 1)))
                calls = r["calls"]
        print(json.dumps({"config": name, "calls": calls, "one_launch(us,b2b)": rows[True], "three_launches(us,b2b)": rows[False]}), flush=True)
    ffa.FrameFusion.one_launch = True


if __name__ == "__main__":
    main()
