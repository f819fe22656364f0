"""Per-kernel times of deflate on a 64 MiB buffer (not a pytest; run on the GPU box): python tools/gpu_deflate_kernels.py [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
z = ge.load(); z.init(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "itext"
t = torch.from_numpy(z.gen(kind, 12345, 64 << 20)).cuda()
out = torch.empty(z.deflate_bound(t.numel()), dtype=torch.uint8, device="cuda")
z.deflate_tensor(t, out)
z.set_profiling(True)
acc = {}
for _ in range(3):
    z.deflate_tensor(t, out)
    for n, ms, l in z.last_kernel_times():
        acc[n] = acc.get(n, 0) + ms / 3
print(kind, " ".join("%s=%.3f" % kv for kv in sorted(acc.items(), key=lambda kv: -kv[1])), "total=%.3f" % sum(acc.values()))
