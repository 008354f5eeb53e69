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


def wsum16(t):
    """host_prune.cpp's checksum of 16-bit words: sum of (index + 1) * word, mod 2^64."""
    w = t.contiguous().view(torch.int16).numpy().astype(np.uint16).reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (np.arange(w.size, dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64))


def build_prune_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    _lib.load()
    exe = tmp_path / "abi_host_prune"
    libdir = os.path.join(ROOT, "framefusion_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_host", "host_prune.cpp"), "-L", libdir, "-lframefusion_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, timeout=300)
    return exe


def parse(stdout):
    out = {}
    for line in stdout.strip().splitlines():
        parts = line.split()
        if len(parts) == 2:
            out[parts[0]] = parts[1]
        else:
            out.update({parts[i]: parts[i + 1] for i in range(0, len(parts), 2)})
    return out


@pytest.mark.parametrize("S,d,H,Hk,dh,num", [(6404, 3584, 28, 4, 128, 4),        # the Qwen2-VL-7B prune (configs[2]: num = 4, MFMA scores)
                                              (35072, 8192, 64, 8, 128, 1),       # the LLaVA-Video-72B prune (configs[4])
                                              (700, 64, 6, 2, 24, 2)])            # an odd head size (the three-launch importance)
def test_cpp_host_prune_half_matches_python_host_and_oracle(tmp_path, S, d, H, Hk, dh, num):
    """The prune half of the path (framefusion/main.py:61-101 fed by utils.py:27-57) through the C ABI from a host that is not
    Python: ff_ctx_last_query_importance + ff_ctx_prune, and ff_ctx_prune_from_qk - kept indices, row hashes and the importance
    against the Python host on the same bytes (bit for bit: the same library), the kept set against the CPU oracle."""
    exe = build_prune_host(tmp_path)
    start, n_img = 14, S - 34
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    sparsity = [0.4, 0.2]
    k = round(n_img * (1 - f._compute_pruning_ratio(sparsity, 0.3)))
    run = subprocess.run([str(exe), "prune"] + [str(x) for x in (S, d, H, Hk, dh, num, start, n_img, k)] + [str(tmp_path)],
                         check=True, capture_output=True, text=True, timeout=600)
    out = parse(run.stdout)
    L_out = S - n_img + k
    assert int(out["L_OUT"]) == L_out and int(out["WS_DIRTY_BYTES"]) == 0 and int(out["CTX_DIRTY"]) == 0
    assert out["HIDDEN_A_FNV"] == out["HIDDEN_B_FNV"] and out["KEPT_A_FNV"] == out["KEPT_B_FNV"]      # hook + prune == one crossing

    def load(name, shape):
        return torch.from_numpy(np.fromfile(tmp_path / name, dtype=np.int16).reshape(shape).copy()).view(torch.bfloat16)
    h, q, key = load("hidden.bin", (1, S, d)), load("q.bin", (1, H, num, dh)), load("k.bin", (1, Hk, S, dh))
    f.prepare(torch.zeros(1, S, dtype=torch.long, device=DEV), 1, start, start + n_img, n_img, S,
              finish_merging=True, finish_pruning=False, sparsity_list=list(sparsity))
    og, kept_g, _ = f.prune_from_qk(h.to(DEV), torch.arange(S, device=DEV)[None], None, q.to(DEV), key.to(DEV), num=num, is_causal=True)
    assert og.shape[1] == L_out
    assert int(out["HIDDEN_A_FNV"], 16) == wsum16(og.cpu())
    assert int(out["KEPT_A_FNV"], 16) == fnv64(kept_g.cpu().reshape(-1).tolist())
    imp = f.last_call["scratch"].sim(torch.bfloat16, S)
    assert int(out["IMP_FNV"], 16) == wsum16(imp.cpu())
    assert int(out["KEEP_FNV"], 16) == fnv64(f.last_plan()["keep"].cpu().reshape(-1).tolist())
    # the oracle: its importance may differ from the kernels' in the last place on a few keys (fp32 summation order), which can
    # move a key across the cut - everything else is kept identically
    w = orc.last_query_attention(q, key, num=num, is_causal=True, enable_gqa=True)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(torch.zeros(1, S, dtype=torch.long), 1, start, start + n_img, n_img, S, finish_merging=True, finish_pruning=False,
              sparsity_list=list(sparsity))
    ho, kept_o, _ = o.forward(h.clone(), torch.arange(S)[None], None, w)
    assert ho.shape[1] == L_out
    a_, b_ = set(kept_g.cpu().reshape(-1).tolist()), set(kept_o.reshape(-1).tolist())
    assert len(a_ ^ b_) <= max(2, L_out // 500), len(a_ ^ b_)
    if a_ == b_:
        assert torch.equal(og.cpu().view(torch.int16), ho.view(torch.int16))


def test_cpp_host_layout_entry_points_feed_a_merge_call(tmp_path):
    """ff_token_span -> ff_fill_patch_type (what the reference's packers build as a Python list,
    llava_video/modeling_llava_video.py:332-336) -> one merge call on that patch_type, from the C++ host; against the Python
    host and the oracle on the same bytes."""
    exe = build_prune_host(tmp_path)
    F, P, d, pre, post = 16, 40, 256, 9, 11
    run = subprocess.run([str(exe), "layout"] + [str(x) for x in (F, P, d, pre, post)] + [str(tmp_path)], check=True, capture_output=True,
                         text=True, timeout=300)
    out = parse(run.stdout)
    L = pre + F * P + post
    assert int(out["L"]) == L and int(out["UNHINTED"]) == 0 and int(out["APPLIED"]) == 1 and int(out["ONE_LAUNCH"]) == 1
    pt = torch.full((1, L), -1, dtype=torch.int64)
    pt[0, pre:pre + F * P] = torch.arange(F * P) % P
    assert int(out["PTYPE_FNV"], 16) == fnv64(pt.reshape(-1).tolist())
    h = torch.from_numpy(np.fromfile(tmp_path / "hidden.bin", dtype=np.int16).reshape(1, L, d).copy()).view(torch.bfloat16)
    o = orc.OracleFrameFusion(0.3, 0.6, 0.1)
    o.prepare(pt.clone(), P, pre, pre + F * P - 1, F * P, L)
    ho, _, _ = o.forward(h.clone(), torch.arange(L)[None], None)
    assert int(out["LOUT"]) == ho.shape[1]
    assert int(out["HIDDEN_FNV"], 16) == wsum16(ho)
    assert int(out["PTYPE_OUT_FNV"], 16) == fnv64(o.patch_type.reshape(-1).tolist())
    f = ffa.FrameFusion(0.3, 0.6, 0.1)
    f.prepare(pt.to(DEV), P, pre, pre + F * P - 1, F * P, L)
    hg, _, _ = f(h.to(DEV), torch.arange(L, device=DEV)[None], None)
    assert int(out["HIDDEN_FNV"], 16) == wsum16(hg.cpu())
    # the same call with its outputs by mail: first slot refused (wrong length), second taken - no second launch, same bits
    assert int(out["LATE_ONE_LAUNCH"]) == 1 and int(out["LATE_MAIL_SLOT"]) == 2
    assert out["LATE_HIDDEN_FNV"] == out["HIDDEN_FNV"] and out["LATE_PTYPE_OUT_FNV"] == out["PTYPE_OUT_FNV"]
