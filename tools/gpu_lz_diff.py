"""First differing LZ77 token between the GPU match finder and the oracle (not a pytest; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import _oracle
z = ge.load(); z.init(0)
kind, seed, n = (sys.argv[1] if len(sys.argv) > 1 else "itext"), int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 40000
a = z.gen(kind, seed, n)
start = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ln = min(n - start, 131072)
got = z.stage_lz77_tensor(torch.from_numpy(a).cuda(), start, ln)
want = _oracle.lz77_block(a, start, ln)
print("tokens", len(got), len(want))
pos = 0; nd = 0
for i in range(min(len(got), len(want))):
    g, w = int(got[i]), int(want[i])
    if g != w:
        f = lambda t: ("M len %d dist %d" % (((t >> 16) & 255) + 3, (t & 0x7fff) + 1)) if t >> 31 else "L %d" % t
        print("token %d at position %d: got %s want %s" % (i, pos, f(g), f(w)))
        nd += 1
        if nd >= 8: break
    pos += (((w >> 16) & 255) + 3) if w >> 31 else 1
print("differences shown", nd)
