"""Phase stamps of the deflate kernels on 64 MiB (ZES_DEBUG_PHASES; run on the GPU box): python tools/gpu_lazy_phases.py [kind]"""
import os, sys
os.environ["ZES_DEBUG_PHASES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
z = ge.load(); z.init(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "itext"
t = torch.from_numpy(z.gen(kind, 12345, 64 << 20)).cuda()
out = torch.empty(z.deflate_bound(t.numel()), dtype=torch.uint8, device="cuda")
z.deflate_tensor(t, out)
z.deflate_tensor(t, out)
