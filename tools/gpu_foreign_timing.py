"""Inflate of streams made by another encoder (CPython's zlib module): segment-parallel tier against the
serial wavefront (not a pytest; run on the GPU box).  usage: gpu_foreign_timing.py [MiB] [levels...]"""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np
import torch
z = ge.load(); z.init(0)
z.set_profiling(True)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
levels = [int(x) for x in sys.argv[2:]] or [6]
n = mib << 20
for kind in ("itext", "lowent4k", "xorshift"):
    a = z.gen(kind, 12345, n)
    t = torch.from_numpy(a).cuda()
    for level in levels:
        comp = torch.from_numpy(np.frombuffer(pz.compress(a.tobytes(), level), dtype=np.uint8).copy()).cuda()
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(comp, back); best = min(best, time.perf_counter() - t0)
        kt = {k: round(ms, 3) for k, ms, nl in z.last_kernel_times()}
        print("%-9s level %d c=%9d inflate %9.2f ms %7.3f GiB/s tier %d ok=%s %s" % (
            kind, level, comp.numel(), best * 1e3, n / best / 2**30, z.last_inflate_tier(), bool((b == t).all()), kt), flush=True)
    if kind == "itext":
        m = min(n, 8 << 20)
        comp = torch.from_numpy(np.frombuffer(pz.compress(a[:m].tobytes(), 6), dtype=np.uint8).copy()).cuda()
        back = torch.empty(m, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(comp, back, z.ZES_F_NO_FASTPATH); dt = time.perf_counter() - t0
        print("%-9s level 6 first %d MiB, serial wavefront: %9.2f ms %7.4f GiB/s tier %d ok=%s" % (
            kind, m >> 20, dt * 1e3, m / dt / 2**30, z.last_inflate_tier(), bool((b == t[:m]).all())), flush=True)
