"""Host-pointer API (PCIe-inclusive) timing at 64 MiB (not a pytest; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
z = ge.load(); z.init(0)
n = 64 << 20
for kind in ("xorshift", "itext"):
    a = z.gen(kind, 12345, n)
    bd = bi = 1e9
    for it in range(4):
        t0 = time.perf_counter(); c = z.deflate(a); bd = min(bd, time.perf_counter() - t0)
        t0 = time.perf_counter(); b = z.inflate(c); bi = min(bi, time.perf_counter() - t0)
    print("%-9s host API: deflate %.2f ms (%.2f GiB/s)  inflate %.2f ms (%.2f GiB/s)  ok=%s" % (
        kind, bd * 1e3, n / bd / 2**30, bi * 1e3, n / bi / 2**30, bool((b == a).all())), flush=True)
