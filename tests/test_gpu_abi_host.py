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


@pytest.mark.parametrize("F,P,d", [(12, 24, 128), (20, 37, 256)])
def test_cpp_host_matches_python_host_and_oracle(tmp_path, F, P, d):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    _lib.load()                                                     # builds the library if needed
    exe = tmp_path / "abi_host"
    libdir = os.path.join(ROOT, "framefusion_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_host", "host.cpp"), "-L", libdir, "-lframefusion_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, timeout=300)
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
