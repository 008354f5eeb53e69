import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def from_bits(arr: np.ndarray, dtype) -> torch.Tensor:
    """Fixture arrays hold bf16/fp16 as int16 bit patterns."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dtype in (torch.bfloat16, torch.float16):
        return t.view(dtype)
    return t


def to_bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy()
    return t.numpy()


def same_bits(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape == b.shape and a.dtype == b.dtype and bool(np.array_equal(to_bits(a), to_bits(b)))


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def cases(self):
        return sorted({k.split("/")[0] for k in self.z.files if "/" in k})

    def __getitem__(self, key):
        return self.z[key]

    def has(self, key):
        return key in self.z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
