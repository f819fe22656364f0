"""Which periodic inputs give a stream whose scan chunks hold more survivors than the scan's LDS list (1024)?
Not a pytest; run on the GPU box with ZES_VERIFY_DBG=1: prints the survivor count of every inflate beside its case
(stream length <= 8192 bytes and survivors > 1024 = one chunk that overflowed).  usage: gpu_scan_overflow_probe.py"""
import os, sys, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
z = ge.load(); z.init(0)
os.environ["ZES_VERIFY_DBG"] = "1"
for kind in ("itext", "lowent4k", "xorshift"):
    for per in (1, 2, 3, 5, 27, 257, 258, 259, 4097):
        for n in (131072, 131074, 400000, 3000000):
            a = np.resize(z.gen(kind, 7 + per, n)[:per], n).copy()
            comp = z.deflate(a)
            print("case", kind, per, n, "c", len(comp), file=sys.stderr, flush=True)
            back = z.inflate(comp)
            assert back.tobytes() == a.tobytes()
            print("   tier", z.last_inflate_tier(), file=sys.stderr, flush=True)
            fz = np.frombuffer(pz.compress(a.tobytes(), 6), dtype=np.uint8)
            print("case zlib", kind, per, n, "c", len(fz), file=sys.stderr, flush=True)
            assert z.inflate(fz).tobytes() == a.tobytes()
            print("   tier", z.last_inflate_tier(), file=sys.stderr, flush=True)
print("ok")
