"""Timeline of the pipelined host deflate (run on the GPU box with ZES_PIPE_DBG=1)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

z = ge.load()
z.init(0)
L = z.lib()
n = 64 << 20
cap = z.deflate_bound(n)
src = z.gen(sys.argv[1] if len(sys.argv) > 1 else "xorshift", 12345, n)
pinned = len(sys.argv) > 2
if pinned:
    a, comp = z.host_alloc(n), z.host_alloc(cap)
    a[:] = src
else:
    a, comp = src, np.ones(cap, dtype=np.uint8)
clen = C.c_uint64()
for i in range(3):
    print("call", i, file=sys.stderr, flush=True)
    assert L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, cap, C.byref(clen)) == 0
