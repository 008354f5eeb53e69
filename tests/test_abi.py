"""CPU: the C-ABI library builds/loads and exports exactly what include/framefusion_hip.h declares;
argument validation happens before any HIP call, so it can be exercised without a GPU."""
import ctypes as C
import os
import re

import pytest

from framefusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "framefusion_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ff_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert "ff_merge_step" in names and "ff_pair_similarity" in names
    assert sorted(_lib.PROTOTYPES) == names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert _lib.load().ff_abi_version() == _lib.ABI_VERSION == 6


def test_stat_enum_matches_binding():
    text = open(HEADER).read()
    vals = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"FF_STAT_([A-Z_]+)\s*=\s*(\d+)", text))
    assert vals["WORDS"] == _lib.STAT_WORDS
    for k in ("NV", "FTN", "COUNT", "BRANCH", "K", "MERGED", "LOUT", "BELOW_LB", "KTH_KEY", "TIES_TAKEN", "SEQ"):
        assert vals[k] == getattr(_lib, "STAT_" + k), k
    dt = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"FF_(F32|BF16|F16)\s*=\s*(\d+)", text))
    assert (dt["F32"], dt["BF16"], dt["F16"]) == (_lib.FF_F32, _lib.FF_BF16, _lib.FF_F16)


def test_error_strings_and_workspace():
    lib = _lib.load()
    assert lib.ff_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert len(lib.ff_error_string(code)) > 4
    assert lib.ff_workspace_bytes(36864, 576) >= 48
    assert lib.ff_workspace_bytes(-1, 1) == 0


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers / bad sizes are rejected before anything is enqueued
    assert lib.ff_build_order(None, 10, 4, None, None, None, None, 0, None) == -1
    assert lib.ff_pair_similarity(None, 1, 10, 64, None, None, None, None, None) == -1
    assert lib.ff_pair_similarity(16, 7, 10, 64, 16, 16, 16, 16, None) == -1          # unknown dtype
    assert lib.ff_pair_similarity(16, 1, 10, 3, 16, 16, 16, 16, None) == -2           # 6-byte rows: alignment
    assert lib.ff_pair_similarity(24, 1, 10, 64, 16, 16, 16, 16, None) == -2          # base not 16-byte aligned
    assert lib.ff_plan_merge(None, 1, None, 10, 0.6, 0.7, 0.1, None, None, None, None, None, 0, None) == -1
    assert lib.ff_plan_prune(16, 1, 10, 8, 5, 2, 16, 16, 16, 16, 16, 1 << 24, None) == -1  # start + n_img > S
    assert lib.ff_plan_prune(16, 1, 10, 2, 5, 2, 16, 16, 16, 16, 16, 64, None) == -4       # workspace too small
    assert lib.ff_merge_compact(None, None, 1, 10, 64, 10, None, None, 1, None, None, None, 0, None) == -1
    assert lib.ff_head_mean(None, 1, 4, 1, 10, None, None) == -1
    assert lib.ff_last_query_attention(16, 16, 1, 6, 4, 1, 10, 64, 0.1, 1, 16, None, 0, 0, None, 0, 16, 1 << 20, None) == -1  # H % H_kv
    # empty problems are a no-op
    assert lib.ff_build_order(16, 0, 4, 16, None, 16, 16, 1 << 24, None) == 0
    assert lib.ff_pair_similarity(16, 1, 0, 64, 16, 16, 16, 16, None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "CSRC", str(tmp_path / "no_sources"))       # nothing to build from either
    with pytest.raises(_lib.FrameFusionHipError, match="no CPU/eager fallback"):
        _lib.load()
