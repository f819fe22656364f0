"""Build-time properties of the generated gfx950 code that a change of compiler version or of the source can break
silently (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not installed")
def test_lazy_matcher_prefetch_registers_are_touched_by_nothing_else(tmp_path):
    """k_lz_match_lazy issues its loop's three prefetch streams by hand into accumulation registers a0-a2 and waits for them
    by count (zes_deflate.hip, LZ_REQ): the generated code must name those registers in the requests and the fetches only,
    every fetch right behind its s_waitcnt, all three phases and both forms of the loop (tools/check_lazy_isa.py) — a copy
    or a reuse of a register whose data is still on its way reads garbage on some runs only."""
    asm = str(tmp_path / "zes_deflate.s")
    src = os.path.join(ROOT, "zlib.es_amd", "csrc", "zes_deflate.hip")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm, "-Wno-unused-function"],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lazy_isa.py"), asm], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "violations 0" in out.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not installed")
def test_no_kernel_spills_to_scratch(tmp_path):
    """Every kernel of the deflate side fits its registers (a spill to scratch memory in one of the block kernels would
    cost more than any of this round's gains)."""
    src = os.path.join(ROOT, "zlib.es_amd", "csrc", "zes_deflate.hip")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", str(tmp_path / "x.o"),
                          "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    scratch = [ln for ln in out.stderr.splitlines() if "ScratchSize [bytes/lane]" in ln]
    assert scratch and all(ln.rstrip().endswith(": 0 [-Rpass-analysis=kernel-resource-usage]") for ln in scratch), scratch
