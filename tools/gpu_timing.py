"""Per-kernel timing at 64 MiB for the three workloads (not a pytest; run on the GPU box)."""
import os, sys, time, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
z.set_profiling(True)
kinds = sys.argv[1:] or ["xorshift", "itext", "lowent4k"]
m = json.load(open(os.path.join(ROOT, "tests/golden/manifest.json")))
for kind in kinds:
    n = 64 << 20
    a = z.gen(kind, 12345, n); t = torch.from_numpy(a).cuda()
    out = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time(); comp = z.deflate_tensor(t, out); dt = time.time() - t0
    kt = {k: round(ms, 3) for k, ms, _ in z.last_kernel_times()}
    print("deflate 64MiB %-9s %7.2f ms %6.2f GiB/s c=%d %s" % (kind, dt * 1e3, n / dt / 2**30, comp.numel(), kt), flush=True)
    exp = [e for e in m["big"] if e["kind"] == kind][0]
    ok = hashlib.sha256(comp.cpu().numpy().tobytes()).hexdigest() == exp["deflate_sha256"]
    cc = comp.clone(); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(3):
        t0 = time.time(); b = z.inflate_tensor(cc, back); dt = time.time() - t0
    kt = {k: round(ms, 3) for k, ms, _ in z.last_kernel_times()}
    print("inflate 64MiB %-9s %7.2f ms %6.2f GiB/s %s" % (kind, dt * 1e3, n / dt / 2**30, kt), flush=True)
    print("   bit-exact deflate: %s   round trip: %s" % (ok, bool(b.numel() == n and bool((b == t).all()))), flush=True)
