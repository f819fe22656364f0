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
# room that runs out in the middle of a stream: the pieces behind are measured only and the call names the size it needs
raw = z.gen("itext", 4243, 24 << 20)
comp = np.frombuffer(pz.compress(raw.tobytes(), 6), dtype=np.uint8)
d_in = torch.from_numpy(comp.copy()).cuda(); small = torch.zeros(10 << 20, dtype=torch.uint8, device="cuda")
try:
    z.inflate_tensor(d_in, small)
    print("too little room: no error"); ok_all = False
except z.ZlibEsError as e:
    good = getattr(e, "need", 0) == raw.size and z.last_inflate_tier() == 2
    print("too little room: need %s (expected %d) tier %d %s" % (getattr(e, "need", None), raw.size, z.last_inflate_tier(), "ok" if good else "WRONG"))
    ok_all = ok_all and good
# a stream cut off in its last piece: not for this tier, the serial tiers give the reference's answer
cut = torch.from_numpy(comp[: comp.size - 70000].copy()).cuda(); big = torch.zeros(raw.size, dtype=torch.uint8, device="cuda")
try:
    z.inflate_tensor(cut, big)
    print("truncated stream: no error"); ok_all = False
except z.ZlibEsError as e:
    print("truncated stream: %s tier %d" % (e, z.last_inflate_tier()))
    ok_all = ok_all and z.last_inflate_tier() in (3, 4)
print("pieces probe:", "ok" if ok_all else "FAILED")
