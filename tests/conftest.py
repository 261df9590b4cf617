import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """Load a committed fixture (data only, allow_pickle stays False) as torch tensors."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        out = {}
        for k in f.files:
            a = f[k]
            out[k] = a if a.dtype.kind in "US" else torch.from_numpy(np.array(a))
        return out


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative error' of BASELINE.json (tensor-level, max-norm)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    denom = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / denom


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _poison_gpu_allocator(request):
    """Before every GPU test, fill a spread of freed allocator blocks with NaN: a kernel that reads memory nothing
    wrote (padding tiles, workspace tails, partial rows) then produces NaN instead of passing by luck on zero pages."""
    if "gpu" in request.keywords and torch.cuda.is_available():
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(48)]
        junk += [torch.full((n,), float("nan"), device="cuda")
                 for n in (64, 256, 1024, 4160, 8320, 16384, 65536, 262144) for _ in range(12)]
        del junk
    yield
