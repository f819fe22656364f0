"""One zlib-6 stream of 8 MiB text through the single-buffer device entry point with the T2 item dump (run on the GPU box)."""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 104
n = int(sys.argv[2]) if len(sys.argv) > 2 else (8 << 20)
raw = z.gen("itext", seed, n)
comp = np.frombuffer(pz.compress(raw.tobytes(), 6), dtype=np.uint8)
d = torch.from_numpy(comp.copy()).cuda(); out = torch.empty(n, dtype=torch.uint8, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = z.inflate_tensor(d, out)
    t = time.perf_counter() - t0
print("seed", seed, "c", len(comp), "%.2f ms" % (t * 1e3), "tier", z.last_inflate_tier(), {k: round(ms, 2) for k, ms, nl in z.last_kernel_times()}, flush=True)
