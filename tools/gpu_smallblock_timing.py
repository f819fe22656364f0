"""Inflate of zlib streams with very small blocks (memLevel 1-3): candidate thinning keeps the segment count bounded (not a pytest)."""
import os, sys, time, zlib as pz
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
n = 64 << 20
a = z.gen("itext", 12345, n); t = torch.from_numpy(a).cuda()
for level, mem in ((6, 1), (6, 3), (1, 1)):
    co = pz.compressobj(level, pz.DEFLATED, 15, mem)
    comp = torch.from_numpy(np.frombuffer(co.compress(a.tobytes()) + co.flush(), dtype=np.uint8).copy()).cuda()
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(comp, back); best = min(best, time.perf_counter() - t0)
    kt = {k: (round(ms, 3), nl) for k, ms, nl in z.last_kernel_times()}  # (ms, launches)
    print("level %d mem %d c=%d inflate %.2f ms %.3f GiB/s tier %d ok=%s %s" % (level, mem, comp.numel(), best * 1e3, n / best / 2**30, z.last_inflate_tier(), bool((b == t).all()), kt), flush=True)
