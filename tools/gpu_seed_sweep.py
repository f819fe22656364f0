"""Round-trip time per seed, the 8 seeds bench.py uses at 8 GPUs (not a pytest; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
n = 64 << 20
for kind in sys.argv[1:] or ["xorshift"]:
    for seed in range(12345, 12353):
        a = z.gen(kind, seed, n); t = torch.from_numpy(a).cuda()
        out = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
        bd = bi = 1e9
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); comp = z.deflate_tensor(t, out); bd = min(bd, time.perf_counter() - t0)
            t0 = time.perf_counter(); b = z.inflate_tensor(comp, back); bi = min(bi, time.perf_counter() - t0)
        print("%s seed %d deflate %.2f ms inflate %.2f ms tier %d ok=%s" % (kind, seed, bd * 1e3, bi * 1e3, z.last_inflate_tier(), bool((b == t).all())), flush=True)
