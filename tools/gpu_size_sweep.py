"""Latency of single-buffer calls over sizes (not a pytest; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
for kind in ("xorshift", "itext"):
    for n in (4096, 65536, 1 << 20, 8 << 20, 64 << 20):
        a = z.gen(kind, 7, n); t = torch.from_numpy(a).cuda()
        out = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device="cuda")
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        best_d = best_i = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); comp = z.deflate_tensor(t, out); best_d = min(best_d, time.perf_counter() - t0)
        cc = comp.clone()
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(cc, back); best_i = min(best_i, time.perf_counter() - t0)
        ok = bool((b == t).all())
        print("%-9s n=%9d  deflate %8.3f ms (%6.2f GiB/s)  inflate %8.3f ms (%6.2f GiB/s) tier %d ok=%s" % (
            kind, n, best_d * 1e3, n / best_d / 2**30, best_i * 1e3, n / best_i / 2**30, z.last_inflate_tier(), ok), flush=True)
