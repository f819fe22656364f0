"""Batch timing (not a pytest; run on the GPU box): BASELINE configs[3] per-GPU share (128 x 1 MiB) and a
many-small-buffers case, through zes_*_batch_dev."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import torch
z = ge.load(); z.init(0)
z.set_profiling(True)
CASES = (("itext", 128, 1 << 20), ("xorshift", 128, 1 << 20), ("itext", 1024, 1 << 20), ("itext", 2048, 65536))
sel = [int(x) for x in sys.argv[1:]] or range(len(CASES))
for kind, count, n in [CASES[i] for i in sel]:
    bound = z.deflate_bound(n)
    bstride = (bound + 15) // 16 * 16
    raw = np.concatenate([z.gen(kind, 1000 + i, n) for i in range(count)])
    d_raw = torch.from_numpy(raw).cuda()
    d_comp = torch.empty(bstride * count, dtype=torch.uint8, device="cuda")
    d_back = torch.zeros(n * count, dtype=torch.uint8, device="cuda")
    in_off = [i * n for i in range(count)]
    c_off = [i * bstride for i in range(count)]
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        clen, st = z.deflate_batch_tensor(d_raw, in_off, [n] * count, d_comp, c_off, [bstride] * count)
        dt = time.time() - t0
    assert not any(st)
    kt = {k: round(ms, 3) for k, ms, _ in z.last_kernel_times()}
    print("deflate %4d x %7d %-9s %8.2f ms %6.2f GiB/s %s" % (count, n, kind, dt * 1e3, n * count / dt / 2**30, kt), flush=True)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        olen, st = z.inflate_batch_tensor(d_comp, c_off, clen, d_back, in_off, [n] * count)
        dt = time.time() - t0
    assert not any(st) and all(o == n for o in olen)
    kt = {k: round(ms, 3) for k, ms, _ in z.last_kernel_times()}
    print("inflate %4d x %7d %-9s %8.2f ms %6.2f GiB/s tier %d %s" % (count, n, kind, dt * 1e3, n * count / dt / 2**30, z.last_inflate_tier(), kt), flush=True)
    print("   round trip:", bool((d_back == d_raw).all()), flush=True)
