"""Runs inflate (and optionally deflate) of one 64 MiB workload a few times: target for rocprofv3 PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "xorshift"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 64 << 20
a = z.gen(kind, 12345, n); t = torch.from_numpy(a).cuda()
comp = z.deflate_tensor(t).clone()
back = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(iters):
    comp2 = z.deflate_tensor(t)
    b = z.inflate_tensor(comp, back)
print("ok", kind, comp.numel(), b.numel())
