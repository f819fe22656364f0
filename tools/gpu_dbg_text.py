import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, torch
z = ge.load(); z.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
a = z.gen("itext", 12345, n); t = torch.from_numpy(a).cuda()
comp = z.deflate_tensor(t).clone()
back = torch.empty(n, dtype=torch.uint8, device="cuda")
b = z.inflate_tensor(comp, back)
print("ok", b.numel() == n and bool((b == t).all()))
