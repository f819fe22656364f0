"""Occupancy experiment (not a pytest): text with period 48 KiB inside every block, so that a lazy matcher whose LDS copy
of the block wraps at 48 KiB computes the real thing.  python tools/gpu_wrap_experiment.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
z = ge.load(); z.init(0)
nblk = 512
parts = []
for b in range(nblk):
    a = z.gen("itext", 777 + b, 48 << 10)
    parts += [a, a, a[: 32 << 10]]
host = np.concatenate(parts)
t = torch.from_numpy(host).cuda()
out = torch.empty(z.deflate_bound(t.numel()), dtype=torch.uint8, device="cuda")
c = z.deflate_tensor(t, out)
print("compressed", c.numel(), hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest()[:16])
z.set_profiling(True)
acc = {}
for _ in range(3):
    z.deflate_tensor(t, out)
    for n, ms, l in z.last_kernel_times():
        acc[n] = acc.get(n, 0) + ms / 3
print(os.environ.get("ZES_LIB"), " ".join("%s=%.3f" % kv for kv in sorted(acc.items(), key=lambda kv: -kv[1])[:4]), "total=%.3f" % sum(acc.values()))
