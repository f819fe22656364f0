"""Per-kernel times of inflate on a reference-format 64 MiB stream (not a pytest): python tools/gpu_inflate_kernels.py [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
z = ge.load(); z.init(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "itext"
t = torch.from_numpy(z.gen(kind, 12345, 64 << 20)).cuda()
comp = z.deflate_tensor(t).clone()
out = torch.empty(t.numel(), dtype=torch.uint8, device="cuda")
z.inflate_tensor(comp, out)
z.set_profiling(True)
acc = {}
for _ in range(5):
    z.inflate_tensor(comp, out)
    for n, ms, l in z.last_kernel_times():
        acc[n] = acc.get(n, 0) + ms / 5
print(kind, os.environ.get("ZES_VERIFY_DIV", "-"), " ".join("%s=%.3f" % kv for kv in sorted(acc.items(), key=lambda kv: -kv[1])), "total=%.3f" % sum(acc.values()))
