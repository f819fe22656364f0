"""Streams of another encoder through the segment-parallel tier piece by piece (what streams of 48 MiB and more —
and everything beyond 512 MiB — take): forced onto small streams by ZES_SEG_PIECE_MB, which the library reads once,
hence a process of its own."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_foreign_streams_in_pieces():
    env = dict(os.environ, ZES_SEG_PIECE_MB="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_pieces_probe.py")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "pieces probe: ok" in p.stdout, p.stdout[-2000:]
    # every stream went through the tier (tier 2), none fell to the serial wavefront
    lines = [ln for ln in p.stdout.splitlines() if " MiB level " in ln]
    assert len(lines) == 5 and all(" tier 2 " in ln for ln in lines), p.stdout[-2000:]
    # room that runs out mid-stream: the size is still reported by this tier; a stream cut off: the serial tiers' error
    assert "too little room: need 25165824 (expected 25165824) tier 2 ok" in p.stdout, p.stdout[-2000:]
    assert "truncated stream: Lack of data length tier 4" in p.stdout, p.stdout[-2000:]
