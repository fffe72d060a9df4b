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
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def group(npz, prefix):
    """Sub-dict of an npz whose keys start with prefix (prefix stripped), as torch tensors."""
    out = {}
    for k in npz.files:
        if k.startswith(prefix):
            v = npz[k]
            out[k[len(prefix):]] = torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v
    return out


def check_summary(t, npz, name, rtol=1e-3, atol=1e-5):
    f = t.detach().flatten().cpu()
    idx = torch.from_numpy(npz[f"{name}.idx"])
    ref = torch.from_numpy(npz[f"{name}.val"])
    assert tuple(t.shape) == tuple(npz[f"{name}.shape"]), (t.shape, npz[f"{name}.shape"])
    torch.testing.assert_close(f[idx].float(), ref, rtol=rtol, atol=atol)
    assert abs(f.double().mean().item() - float(npz[f"{name}.mean"])) <= atol + rtol * abs(float(npz[f"{name}.mean"])) + 1e-6


@pytest.fixture(scope="session")
def golden():
    return load_golden
