"""Host-pointer API (PCIe-inclusive) timing at 64 MiB (not a pytest; run on the GPU box).

Calls the C-ABI directly with output arrays that exist (and have been touched) before the clock starts: a binding
that allocates a fresh 128 MiB array per call pays the page faults of that array, not the library's time.
Rows: pageable caller memory (staged through the library's pinned ring) and pinned caller memory (zes_host_alloc).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

z = ge.load()
z.init(0)
L = z.lib()
n = int(os.environ.get("HOST_N_MB", "64")) << 20  # (HOST_N_MB: another size)
cap = z.deflate_bound(n)


def run(kind, pinned):
    src = z.gen(kind, 12345, n)
    if pinned:
        a, comp, back = z.host_alloc(n), z.host_alloc(cap), z.host_alloc(n)
        a[:] = src
    else:
        a, comp, back = src, np.zeros(cap, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    clen, blen = C.c_uint64(), C.c_uint64()
    bd = bi = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        rc = L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, cap, C.byref(clen))
        bd = min(bd, time.perf_counter() - t0)
        assert rc == 0
        t0 = time.perf_counter()
        rc = L.zes_inflate(comp.ctypes.data, clen.value, back.ctypes.data, n, C.byref(blen), 0)
        bi = min(bi, time.perf_counter() - t0)
        assert rc == 0 and blen.value == n
    ok = bool((back == src).all())
    print("%-9s %-8s c=%d host API: deflate %.2f ms (%.2f GiB/s)  inflate %.2f ms (%.2f GiB/s)  ok=%s" % (
        kind, "pinned" if pinned else "pageable", clen.value, bd * 1e3, n / bd / 2**30, bi * 1e3, n / bi / 2**30, ok), flush=True)
    if pinned:
        for x in (a, comp, back):
            z.host_free(x)


for kind in ("xorshift", "itext"):
    for pinned in (False, True):
        run(kind, pinned)
