import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0); z.set_profiling(True)
for count, n in ((8, 8 << 20), (8, 8 << 20)):
    raws = [z.gen("itext", 100 + i, n) for i in range(count)]
    comps = [np.frombuffer(pz.compress(r.tobytes(), 6), dtype=np.uint8) for r in raws]
    in_off, pos = [], 0
    for cdat in comps:
        in_off.append(pos); pos += (len(cdat) + 15) // 16 * 16
    big = np.zeros(pos, dtype=np.uint8)
    for cdat, o in zip(comps, in_off):
        big[o:o + len(cdat)] = cdat
    d_in = torch.from_numpy(big).cuda(); d_out = torch.zeros(count * n, dtype=torch.uint8, device="cuda")
    out_off = [i * n for i in range(count)]
    best = 1e9
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        olen, st = z.inflate_batch_tensor(d_in, in_off, [len(x) for x in comps], d_out, out_off, [n] * count)
        best = min(best, time.perf_counter() - t0)
    kt = {k: round(ms, 2) for k, ms, nl in z.last_kernel_times()}
    print("%5d x %8d: %9.2f ms tier %d %s" % (count, n, best * 1e3, z.last_inflate_tier(), kt), flush=True)
