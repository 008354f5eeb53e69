"""GPU: the C ABI driven by a host that is NOT Python and does not know torch: tests/abi_host/host.cpp
(plain HIP runtime + libframefusion_hip.so, one ff_merge_step) is compiled with hipcc, run, and its
results compared with the Python host and with the CPU oracle on the very same input bytes."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

import framefusion_amd as ffa
from framefusion_amd import _lib
from oracle import ff_oracle as orc
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def fnv64(words) -> int:
    h = 1469598103934665603
    for v in words:
        h ^= int(v) & 0xFFFFFFFFFFFFFFFF
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    _lib.load()                                                     # builds the library if needed
    exe = tmp_path / "abi_host"
    libdir = os.path.join(ROOT, "framefusion_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_host", "host.cpp"), "-L", libdir, "-lframefusion_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, timeout=300)
    return exe


@pytest.mark.parametrize("F,P,d", [(12, 24, 128), (20, 37, 256)])
def test_cpp_host_matches_python_host_and_oracle(tmp_path, F, P, d):
    exe = build_host(tmp_path)
    blob = tmp_path / "input.bin"
    run = subprocess.run([str(exe), str(F), str(P), str(d), str(blob)], check=True, capture_output=True, text=True, timeout=120)
    lines = run.stdout.strip().splitlines()
    head = lines[0].split()
    facts = {head[i]: int(head[i + 1]) for i in range(0, len(head), 2)}
    hidden_fnv = int(lines[1].split()[1], 16)
    ptype_fnv = int(lines[2].split()[1], 16)
    keep_cpp = np.array([c == "1" for c in lines[3].split()[1]])

    pre, post = 5, 7
    L = pre + F * P + post
    assert facts["L"] == L and facts["NV"] == facts["FTN"] == F * P
    raw = np.fromfile(blob, dtype=np.int16).reshape(1, L, d)
    h = torch.from_numpy(raw.copy()).view(torch.bfloat16)
    pt = torch.full((1, L), -1, dtype=torch.int64)
    pt[0, pre:pre + F * P] = torch.arange(F * P) % P

    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), P, pre, pre + F * P - 1, F * P, L)
    ho, po, _ = o.forward(h.clone(), torch.arange(L)[None], None)
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    f.prepare(pt.to(DEV), P, pre, pre + F * P - 1, F * P, L)
    hg, pg, _ = f(h.to(DEV), torch.arange(L, device=DEV)[None], None)

    keep_o = np.zeros(L, dtype=bool)
    keep_o[po[0].numpy()] = True
    assert facts["LOUT"] == ho.shape[1] == hg.shape[1]
    assert np.array_equal(keep_cpp, keep_o)
    assert np.array_equal(keep_cpp, f.last_plan()["keep"].bool().cpu().numpy())
    assert facts["BRANCH"] == f.last_call["branch"] and facts["COUNT"] == f.last_call["count"]
    assert hidden_fnv == fnv64(ho.view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist())
    assert hidden_fnv == fnv64(hg.cpu().view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist())
    assert ptype_fnv == fnv64(o.patch_type.reshape(-1).tolist())


def bf16_words(t):
    return t.contiguous().view(torch.int16).numpy().astype(np.uint16).reshape(-1).tolist()


@pytest.mark.parametrize("F,P,d", [(16, 30, 256), (64, 576, 4096)])
def test_cpp_host_full_size_context_calls(tmp_path, F, P, d):
    """The call-context entry points (ff_ctx_merge) from the torch-free C++ host, at a small size and at C2's
    (64 x 576 x 4096 bf16, BASELINE.json configs[1]): first call with `addend` (base + addend = the sequence that is
    reduced), M-RoPE style [3, L, 128] cos / sin aux tensors and the layout hint; second call on the compacted
    output with the maintained order (order_valid = 1).  Kept sets, lengths, branch facts and FNV hashes of every
    output against the CPU oracle fed with the eager sum of the same bytes."""
    exe = build_host(tmp_path)
    blob = tmp_path / "input.bin"
    run = subprocess.run([str(exe), str(F), str(P), str(d), str(blob), "full"], check=True, capture_output=True, text=True, timeout=600)
    out = {}
    calls = []
    for line in run.stdout.strip().splitlines():
        parts = line.split()
        if parts[0] == "CALL":
            calls.append({parts[i]: int(parts[i + 1]) for i in range(2, len(parts), 2)})
        elif parts[0] == "L":
            out.update({parts[i]: int(parts[i + 1]) for i in range(0, len(parts), 2)})
        else:
            out[parts[0]] = parts[1] if len(parts) > 1 else ""
    pre, post = 5, 7
    L = pre + F * P + post
    assert out["L"] == L and len(calls) == 2 and calls[0]["UNHINTED"] == 0 and out["ORDER_REUSED"] == 1
    assert out["PAIR_OK"] == "1"            # two contexts in flight from the one host thread gave the first call's output twice
    h = torch.from_numpy(np.fromfile(blob, dtype=np.int16).reshape(1, L, d).copy()).view(torch.bfloat16)
    pt = torch.full((1, L), -1, dtype=torch.int64)
    pt[0, pre:pre + F * P] = torch.arange(F * P) % P
    # the tables of host.cpp (table_at), [3, 1, L, 128] like Qwen2-VL's M-RoPE cos / sin
    i = torch.arange(L)[None, :, None]
    c = torch.arange(128)[None, None, :]
    p = torch.arange(3)[:, None, None]
    tabs = [(((w * 5 + p * 7 + i * 3 + c) % 33 - 16).float() * 0.0625).to(torch.bfloat16)[:, None] for w in range(2)]

    o = orc.OracleFrameFusion(0.3, 0.6, 0.01)
    o.prepare(pt.clone(), P, pre, pre + F * P - 1, F * P, L)
    # the C++ host fixes the budget of each call (sub = 0.7, then 0.55) instead of deriving it from the cost: the
    # oracle's budget is overridden to the same numbers for the two calls
    subs = iter([0.7, 0.55])
    orc_budget = orc.budget
    orc.budget = lambda sparsity_list, cost, num_layers=28: next(subs)
    try:
        h1, pe1, _ = o.forward(h.clone(), [t.clone() for t in tabs], None)
        keep1 = o.last_keep.clone()
        facts1 = (o.finish_merging, list(o.sparsity_list))
        o.finish_merging, o.finish_pruning = False, False                 # (the host issues the second call regardless)
        h2, pe2, _ = o.forward(h1.clone(), pe1, None)
        keep2 = o.last_keep.clone()
    finally:
        orc.budget = orc_budget
    assert calls[0]["LOUT"] == h1.shape[1] and calls[0]["NV"] == calls[0]["FTN"] == F * P
    k1 = np.zeros(L, dtype=bool); k1[keep1.numpy()] = True
    assert np.array_equal(np.array([ch == "1" for ch in out["KEEP1"]]), k1)
    assert int(out["HIDDEN1_FNV"], 16) == fnv64(bf16_words(h1))
    assert calls[1]["LOUT"] == h2.shape[1]
    k2 = np.zeros(h1.shape[1], dtype=bool); k2[keep2.numpy()] = True
    assert np.array_equal(np.array([ch == "1" for ch in out["KEEP2"]]), k2)
    if h2.shape[1] != h1.shape[1]:
        assert int(out["HIDDEN2_FNV"], 16) == fnv64(bf16_words(h2))
        assert int(out["PTYPE2_FNV"], 16) == fnv64(o.patch_type.reshape(-1).tolist())
        for w in range(2):
            assert int(out[f"TABLE{w}_FNV"], 16) == fnv64(bf16_words(pe2[w]))
