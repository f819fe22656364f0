import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def z():
    """The product package (zlib.es_amd/ loaded as zlibes_amd); builds the HIP library if missing."""
    import __graft_entry__ as ge

    mod = ge.load()
    if not os.path.exists(os.path.join(ROOT, "zlib.es_amd", "libzes_hip.so")):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import _oracle

    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def gpu(z):
    """Initialised device; fails (not skips) when the HIP path cannot run on a GPU box."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    z.init(0)
    return torch.device("cuda:0")


def have_reference():
    return os.path.exists("/root/reference/dist/cjs/zlib.js") and shutil.which("node") is not None
