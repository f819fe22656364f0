"""Range decoder by hand (run on the GPU box with ZES_RANGE_DBG=1): the ranges of one stream, one call each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import _oracle  # noqa: E402

z = ge.load()
z.init(0)
n = 9 * 131072 + 999
a = z.gen("itext", 61, n)
comp = _oracle.deflate(a)
starts, ends = _oracle.inflate_blocks(comp)
print("c", len(comp), "starts", starts, flush=True)
t = torch.from_numpy(comp).cuda()
world = 4
for r in range(world):
    lo, own = 8 * max(2, len(comp) * r // world), 8 * (len(comp) * (r + 1) // world)
    out = torch.zeros(n + 131072, dtype=torch.uint8, device="cuda")
    print(r, lo, own, z.inflate_range_tensor(t, lo, own, r == 0, out), flush=True)
