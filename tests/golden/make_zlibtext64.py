#!/usr/bin/env python3
"""make_zlibtext64.py — golden for bench.py's `zlibtext64` leg (a stream another encoder made).

    python tests/golden/make_zlibtext64.py [/root/reference]

Writes the stream CPython's zlib (level 6) makes of the 64 MiB `itext` buffer (seed 12345) to a scratch file,
then has make_golden.js (GOLDEN_FOREIGN=...) run the REFERENCE's inflate over it: foreign_big.json keeps the
stream's length + sha256 (so a box with another zlib build notices it holds a different stream) and the
length + sha256 of what the reference decoded.  Only those numbers are committed.
"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402  (the generators live in the C-ABI library: zes_gen.c)

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
z = ge.load()
for kind, seed, n, level in (("itext", 12345, 64 << 20, 6),):
    data = z.gen(kind, seed, n)
    comp = zlib.compress(data.tobytes(), level)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "stream.zz")
        open(path, "wb").write(comp)
        json.dump({"name": "zlib%d_%s_%d_%d" % (level, kind, seed, n), "kind": kind, "seed": seed, "n": n, "level": level,
                   "zlib_version": zlib.ZLIB_RUNTIME_VERSION}, open(path + ".json", "w"))
        subprocess.run(["node", "--max-old-space-size=8192", os.path.join(HERE, "make_golden.js"), ref],
                       env=dict(os.environ, GOLDEN_FOREIGN=path), check=True)
