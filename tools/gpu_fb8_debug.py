import os, sys, time, zlib as pz
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
import numpy as np, torch
z = ge.load(); z.init(0)
count, n = 8, 8 << 20
raws = [z.gen("itext", 100 + i, n) for i in range(count)]
comps = [np.frombuffer(pz.compress(r.tobytes(), 6), dtype=np.uint8) for r in raws]
in_off, pos = [], 0
for cdat in comps:
    in_off.append(pos); pos += (len(cdat) + 15) // 16 * 16
big = np.zeros(pos, dtype=np.uint8)
for cdat, o in zip(comps, in_off):
    big[o:o + len(cdat)] = cdat
d_in = torch.from_numpy(big).cuda(); d_out = torch.zeros(count * n, dtype=torch.uint8, device="cuda")
olen, st = z.inflate_batch_tensor(d_in, in_off, [len(x) for x in comps], d_out, [i * n for i in range(count)], [n] * count)
print(st[:3], olen[:3])
