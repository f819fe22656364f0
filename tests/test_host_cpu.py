"""CPU-side checks of the Node host layer: the addon builds and loads, the façade exports the
reference's names, and without a GPU it throws instead of falling back."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_napi_addon_loads_and_mirrors_the_reference_api(z):
    host = os.path.join(ROOT, "zlib.es_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    js = (
        "const z=require(%r);"
        "if(typeof z.deflate!=='function'||typeof z.inflate!=='function')process.exit(3);"
        "if(z.deflate.length!==1||z.inflate.length!==1)process.exit(4);"
        "try{z.deflate(new Uint8Array(1));process.exit(5);}catch(e){if(e.message!=='Data is corrupted')process.exit(6);}"
        "try{z.inflate(new Uint8Array(0));process.exit(7);}catch(e){if(e.message!=='Not compressed by deflate')process.exit(8);}"
        "console.log('ok');" % os.path.join(host, "zlib.js")
    )
    out = subprocess.run(["node", "-e", js], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stderr)


def test_generated_js_is_in_sync_with_ts():
    host = os.path.join(ROOT, "zlib.es_amd", "host")
    before = open(os.path.join(host, "zlib.js")).read()
    subprocess.check_call(["python3", os.path.join(host, "strip_types.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(host, "zlib.js")).read() == before
