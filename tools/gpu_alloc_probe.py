"""zes_inflate vs zes_inflate_alloc on a 64 MiB host call (not a pytest)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
z = ge.load(); z.init(0); L = z.lib()
n = 64 << 20
for kind in ("xorshift", "itext"):
    src = z.gen(kind, 12345, n)
    a, comp, back = z.host_alloc(n), z.host_alloc(z.deflate_bound(n)), z.host_alloc(n)
    a[:] = src; comp[:] = 0; back[:] = 0
    clen, blen = C.c_uint64(), C.c_uint64()
    assert L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, comp.size, C.byref(clen)) == 0
    cb = z.ALLOC_FN(lambda u, i, need: back.ctypes.data)
    for name, fn in (("zes_inflate", lambda: L.zes_inflate(comp.ctypes.data, clen.value, back.ctypes.data, n, C.byref(blen), 0)),
                     ("zes_inflate_alloc", lambda: L.zes_inflate_alloc(comp.ctypes.data, clen.value, cb, None, C.byref(blen), 0)),
                     ("..._alloc, BOUND", lambda: L.zes_inflate_alloc(comp.ctypes.data, clen.value, cb, None, C.byref(blen), z.ZES_F_ALLOC_BOUND))):
        fn(); best = 1e9
        for _ in range(8):
            t0 = time.perf_counter(); rc = fn(); best = min(best, time.perf_counter() - t0)
            assert rc == 0 and blen.value == n and (back == src).all()
        print("%-9s %-18s %.3f ms  %.1f GiB/s" % (kind, name, best * 1e3, n / best / 2**30), flush=True)
