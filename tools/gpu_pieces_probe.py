"""Another encoder's stream through the segment-parallel tier piece by piece (ZES_SEG_PIECE_MB=<MiB> forces the
piece path that streams of 512 MiB and more take; not a pytest, run on the GPU box)."""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
ok_all = True
for kind, n, level in (("itext", 48 << 20, 6), ("itext", 48 << 20, 1), ("lowent4k", 48 << 20, 6), ("xorshift", 16 << 20, 6), ("itext", 3 << 20, 9)):
    raw = z.gen(kind, 4242, n)
    comp = np.frombuffer(pz.compress(raw.tobytes(), level), dtype=np.uint8)
    d_in = torch.from_numpy(comp.copy()).cuda(); d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    best = 1e9
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = z.inflate_tensor(d_in, d_out)
        best = min(best, time.perf_counter() - t0)
    ok = out.numel() == n and bool((out.cpu().numpy() == raw).all())
    ok_all = ok_all and ok
    kt = {k: round(ms, 2) for k, ms, nl in z.last_kernel_times()}
    print("%-9s %3d MiB level %d c=%d: %8.2f ms tier %d ok=%s %s" % (kind, n >> 20, level, len(comp), best * 1e3, z.last_inflate_tier(), ok, kt), flush=True)
print("pieces probe:", "ok" if ok_all else "FAILED")
