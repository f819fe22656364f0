"""A zlib stream of more than 512 MiB from another encoder (CPython's zlib, level 1, ~1.7 GB of text): the
segment-parallel tier takes it piece by piece (tier 2), where round 2 left it to the serial wavefront.
Not a pytest (the compression alone takes half a minute on the host); run on the GPU box."""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
n = int(os.environ.get("BIG_N_MB", "1700")) << 20
t0 = time.time()
parts, co, chunk = [], pz.compressobj(1), 64 << 20
raw_gpu = torch.empty(n, dtype=torch.uint8, device="cuda")
for off in range(0, n, chunk):
    m = min(chunk, n - off)
    r = z.gen("itext", 9000 + off // chunk, m)
    raw_gpu[off:off + m] = torch.from_numpy(r).cuda()
    parts.append(co.compress(r.tobytes()))
parts.append(co.flush())
comp = np.frombuffer(b"".join(parts), dtype=np.uint8)
print("input %d MiB, stream %d MiB (%.1f s on the host)" % (n >> 20, len(comp) >> 20, time.time() - t0), flush=True)
assert len(comp) >= (1 << 29) or os.environ.get("BIG_ANY"), "the stream is shorter than 512 MiB: raise BIG_N_MB"
d_in = torch.from_numpy(comp.copy()).cuda(); d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = z.inflate_tensor(d_in, d_out)
    dt = time.perf_counter() - t0
    ok = out.numel() == n and bool((out == raw_gpu).all())
    kt = {k: round(ms, 2) for k, ms, nl in z.last_kernel_times()}
    print("inflate %.1f ms = %.2f GiB/s tier %d ok=%s %s" % (dt * 1e3, n / dt / 2**30, z.last_inflate_tier(), ok, kt), flush=True)
