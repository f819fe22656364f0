"""The host side in the reference's own language: TypeScript façade (erased to zlib.js) over the
N-API addon, driven under Node with the reference suite's assertions (test/index.js)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_reference_suite_through_napi_facade(gpu, z, tmp_path):
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    host = os.path.join(ROOT, "zlib.es_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    # inputs of BASELINE.json configs[3] (this GPU's 128 buffers): generator i % 3, seed 12345 + i, 1 MiB each
    mix = ("xorshift", "itext", "lowent4k")
    path = str(tmp_path / "batch1m.bin")
    with open(path, "wb") as f:
        for i in range(128):
            f.write(z.gen(mix[i % 3], 12345 + i, 1 << 20).tobytes())
    env = dict(os.environ, ZES_BATCH1M_INPUT=path)
    out = subprocess.run([node, os.path.join(ROOT, "tests", "host_node_test.js")], capture_output=True, text=True, timeout=240, env=env)
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
