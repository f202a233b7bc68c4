import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
DT_CODE = {"bf16": 0, "fp16": 1, "f32": 2}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """-m gpu tests need a real MI355X: skip them (instead of failing) where there is none."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X): run with -m gpu on the GPU box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def pq_opt():
    """Set behaviour switches of libpq_hip.so for one test (pq_set_option; the environment is only read once) and
    restore the defaults afterwards."""
    from protoquant_amd import _lib
    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.set_option(name, value)
    yield set_
    for n in touched:
        _lib.set_option(n, "")          # "" = the default of every switch


def load_golden(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["name"] = os.path.basename(path)[:-4]
    d["dtype"] = str(d["dtype"])
    d["code"] = DT_CODE[d["dtype"]]
    d.setdefault("bias", None)
    return d


@pytest.fixture(params=GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def golden(request):
    return load_golden(request.param)


def _producer(kind):
    return sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "producer", kind + "_*.npz")))


def _load_npz(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["code"] = int(d["code"])
    return d


SILU_GOLDEN, RMS_GOLDEN = _producer("silu_mul"), _producer("rmsnorm")


@pytest.fixture(params=SILU_GOLDEN, ids=[os.path.basename(p)[:-4] for p in SILU_GOLDEN])
def producer_golden(request):
    return _load_npz(request.param)


@pytest.fixture(params=RMS_GOLDEN, ids=[os.path.basename(p)[:-4] for p in RMS_GOLDEN])
def rms_golden(request):
    return _load_npz(request.param)
