"""The host side in the reference's own language: TypeScript façade (erased to zlib.js) over the
N-API addon, driven under Node with the reference suite's assertions (test/index.js)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_reference_suite_through_napi_facade(gpu):
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    host = os.path.join(ROOT, "zlib.es_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    out = subprocess.run([node, os.path.join(ROOT, "tests", "host_node_test.js")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "host checks passed" in out.stdout


def test_es_module_entry(gpu):
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    host = os.path.join(ROOT, "zlib.es_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    out = subprocess.run([node, os.path.join(ROOT, "tests", "host_node_esm_test.mjs")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "esm entry ok" in out.stdout
