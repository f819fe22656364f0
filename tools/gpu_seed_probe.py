"""Per-kernel inflate times for xorshift seeds (not a pytest; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
z.set_profiling(True)
n = 64 << 20
for seed in [int(x) for x in sys.argv[1:]] or [7]:
    a = z.gen("xorshift", seed, n); t = torch.from_numpy(a).cuda()
    out = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device="cuda")
    comp = z.deflate_tensor(t, out).clone(); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(comp, back); dt = time.perf_counter() - t0
    kt = {k: (round(ms, 3), nl) for k, ms, nl in z.last_kernel_times()}
    print("seed %d inflate %.2f ms tier %d ok=%s %s" % (seed, dt * 1e3, z.last_inflate_tier(), bool((b == t).all()), kt), flush=True)
