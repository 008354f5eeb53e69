#!/usr/bin/env python
"""How much does the one-thread pair depend on WHICH two HIP streams it runs on? (development probe: the pair on streams from
torch's pool, on streams created through the HIP runtime directly (torch.cuda.ExternalStream), on a normal + a high-priority
stream; the two-thread form in between; profiles/r05_pair_probe.txt)"""
import ctypes, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
F, P, d = 64, 576, 4096
work = []
for t in range(2):
    h, pt = video_tokens(F, P, d, p_change=0.2, sigma=0.3, seed=1334 + t, dtype=torch.bfloat16, device=str(dev))
    L = h.shape[1]
    cos, sin = rotary_tables(L, 128, torch.bfloat16, device=str(dev))
    work.append(dict(h=h, h2=h.clone(), pt=pt, cos=cos, sin=sin, L=L, stream=torch.cuda.Stream(device=dev), out=None))
steps, warmup = 100, 20
hip = ctypes.CDLL("libamdhip64.so.7")


def hip_stream(flags=1):                      # hipStreamNonBlocking
    s = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithFlags(ctypes.byref(s), ctypes.c_uint(flags))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def fresh():
    for w in work:
        w["ff"] = ffa.FrameFusion(0.3, 0.6, 0.1, compact_outputs=False)


def run_pair(pair):
    def calls(n):
        for i in range(2 * n):
            w = work[i & 1]
            pair.ffs[i & 1].prepare(w["pt"], P, 0, w["L"], w["L"], w["L"])
            yield (i & 1, w["h2"] if (i >> 1) & 1 else w["h"], [w["cos"], w["sin"]], None)
    for _ in pair.run(calls(warmup)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in pair.run(calls(steps)):
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def new_pair(streams=None):
    fresh()
    pair = ffa.FrameFusionPair(work[0]["ff"], work[1]["ff"], None, sync_with_current=False)
    if streams is not None:
        pair._streams[(dev.type, dev.index)] = streams
    return pair


def run_threads():
    fresh()
    start, stop = threading.Barrier(3), threading.Barrier(3)

    def run(w):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(w["stream"]):
            def step(i):
                w["ff"].prepare(w["pt"], P, 0, w["L"], w["L"], w["L"])
                return w["ff"](w["h2"] if i & 1 else w["h"], [w["cos"], w["sin"]], None)[0]
            for i in range(warmup):
                step(i)
            w["stream"].synchronize()
            start.wait()
            for i in range(steps):
                w["out"] = step(i)
            w["stream"].synchronize()
            stop.wait()
    threads = [threading.Thread(target=run, args=(w,)) for w in work]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    stop.wait()
    dt = time.perf_counter() - t0
    for th in threads:
        th.join()
    return dt / steps * 1e6


def show(tag, us, pair=None):
    extra = ""
    if pair is not None:
        extra = "  streams " + " ".join(hex(s.cuda_stream) for s in pair.streams)
    print(f"{tag:52s} {us:7.1f} us per two calls{extra}", flush=True)


p0 = new_pair(); show("pair (the process's pair streams: torch pool)", run_pair(p0), p0)
for n in range(4):
    p = new_pair((torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)))
    show(f"torch pool streams #{n}", run_pair(p), p)
for n in range(4):
    p = new_pair((hip_stream(), hip_stream()))
    show(f"hipStreamCreateWithFlags(nonblocking) x 2 #{n}", run_pair(p), p)
for n in range(3):
    p = new_pair((torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)))
    show(f"normal + high priority #{n}", run_pair(p), p)
p = new_pair((torch.cuda.current_stream(dev), hip_stream()))
show("PyTorch's current stream + one new stream", run_pair(p), p)
show("two threads", run_threads())
p0 = new_pair(); show("pair (the process's pair streams) after threads", run_pair(p0), p0)
