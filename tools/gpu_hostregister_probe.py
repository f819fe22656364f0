"""Cost of hipHostRegister + H2D/D2H + hipHostUnregister on 64 MiB pageable arrays (is pinning the caller's buffer in
place cheaper than staging it through the pinned ring?)."""
import ctypes as C
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
n = 64 << 20
hip.hipInit(0)
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
arrs = [np.full(n, i + 1, dtype=np.uint8) for i in range(6)]
for i, a in enumerate(arrs):
    p = C.c_void_p(a.ctypes.data)
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(p, C.c_size_t(n), C.c_uint(0))
    t1 = time.perf_counter()
    rc1 = hip.hipMemcpy(d, p, C.c_size_t(n), C.c_int(1))  # H2D
    t2 = time.perf_counter()
    rc2 = hip.hipMemcpy(p, d, C.c_size_t(n), C.c_int(2))  # D2H
    t3 = time.perf_counter()
    rc3 = hip.hipHostUnregister(p)
    t4 = time.perf_counter()
    print("array %d: register %.3f  h2d %.3f  d2h %.3f  unregister %.3f ms (rc %d %d %d %d)" % (
        i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, rc, rc1, rc2, rc3), flush=True)
a = arrs[0]
p = C.c_void_p(a.ctypes.data)
for i in range(3):
    t1 = time.perf_counter()
    hip.hipMemcpy(d, p, C.c_size_t(n), C.c_int(1))
    t2 = time.perf_counter()
    hip.hipMemcpy(p, d, C.c_size_t(n), C.c_int(2))
    t3 = time.perf_counter()
    print("unregistered: h2d %.3f  d2h %.3f ms" % ((t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
