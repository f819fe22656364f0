"""Single foreign streams (CPython zlib) of several sizes, levels and contents through the device entry point: time,
tier and kernel times (run on the GPU box) — a sweep for performance cliffs of the segment-parallel tier."""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
for kind in ("itext", "lowent4k", "xorshift"):
    for n in (1 << 20, 4 << 20, 16 << 20, 48 << 20):
        for level in (1, 6, 9):
            if kind == "xorshift" and level != 6:
                continue
            raw = z.gen(kind, 1000 + level, n)
            comp = np.frombuffer(pz.compress(raw.tobytes(), level), dtype=np.uint8)
            d = torch.from_numpy(comp.copy()).cuda(); out = torch.empty(n, dtype=torch.uint8, device="cuda")
            best = 1e9
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                b = z.inflate_tensor(d, out)
                best = min(best, time.perf_counter() - t0)
            ok = b.numel() == n and bool((b.cpu().numpy() == raw).all())
            kt = sorted(((k, round(ms, 2)) for k, ms, nl in z.last_kernel_times()), key=lambda kv: -kv[1])[:3]
            print("%-9s %3d MiB level %d: c/n %.3f  %8.2f ms %7.2f GiB/s tier %d ok=%s top %s" % (kind, n >> 20, level, len(comp) / n, best * 1e3, n / best / 2**30,
                                                                                            z.last_inflate_tier(), ok, kt), flush=True)
