"""Cycle profile of the serial wavefront decoder (build the library with -DWD_PROFILE; not a pytest)."""
import os, sys, time, zlib as pz
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0)
for kind in ("itext", "lowent4k", "xorshift"):
    a = z.gen(kind, 12345, 4 << 20)
    comp = torch.from_numpy(np.frombuffer(pz.compress(a.tobytes(), 6), dtype=np.uint8).copy()).cuda()
    back = torch.empty(a.size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter(); b = z.inflate_tensor(comp, back, z.ZES_F_NO_FASTPATH); dt = time.perf_counter() - t0
    print(kind, "%.1f ms" % (dt * 1e3), bool((b.cpu().numpy() == a).all()), flush=True)
