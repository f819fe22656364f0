"""The index of one block (sd[] / inv[], k_lz_sort or k_lz_index) against its definition (run on the GPU box):
python tools/gpu_index_check.py [kind] [n]   — uses ZES_DUMP_INDEX of zes_stage_lz77_dev."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["ZES_DUMP_INDEX"] = "/tmp/zes_index.bin"
import __graft_entry__ as ge
z = ge.load(); z.init(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "itext"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
a = z.gen(kind, 3, n)
z.stage_lz77_tensor(torch.from_numpy(a).cuda(), 0, n)
raw = open("/tmp/zes_index.bin", "rb").read()
flag = np.frombuffer(raw[:4], dtype=np.uint32)[0]
inv = np.frombuffer(raw[4:4 + 4 * 131072], dtype=np.uint32)
sd = np.frombuffer(raw[4 + 4 * 131072:], dtype=np.uint16)
print("flag %#x" % flag)
cnt = n - 2
b = a.astype(np.uint32)
key = b[:-2] | (b[1:-1] << 8) | (b[2:] << 16)
last = {}
bad = 0
slots = set()
for p in range(cnt):
    k = int(key[p]); q = last.get(k); last[k] = p
    want_has = q is not None and p - q <= 32768
    iv = int(inv[p])
    if not want_has:
        if iv != 0xFFFFFFFF:
            bad += 1
            if bad < 10: print("p", p, "should have no candidate, inv %#x" % iv)
        continue
    if iv == 0xFFFFFFFF:
        bad += 1
        if bad < 10: print("p", p, "has candidate", q, "but inv NONE")
        continue
    r = iv & 0x1FFFF
    if r in slots: print("slot twice", r)
    slots.add(r)
    if int(sd[r]) != p - q:
        bad += 1
        if bad < 10: print("p", p, "slot", r, "sd", int(sd[r]), "want", p - q)
# chains: slot r-1 must hold q
pos_of = {}
for p in range(cnt):
    iv = int(inv[p])
    if iv != 0xFFFFFFFF: pos_of[iv & 0x1FFFF] = p
last = {}
for p in range(cnt):
    k = int(key[p]); q = last.get(k); last[k] = p
    iv = int(inv[p])
    if iv == 0xFFFFFFFF or q is None or p - q > 32768: continue
    r = iv & 0x1FFFF
    # the slot before must belong to q unless q has no candidate itself (then its slot is unknown to inv): check via sd only
    if (r - 1) in pos_of and pos_of[r - 1] != q:
        bad += 1
        if bad < 20: print("p", p, "slot", r, "slot-1 holds", pos_of[r - 1], "want", q)
print("bad", bad)
H = (key.astype(np.uint64) * 0x9E3779) & 0xFFFFFF
cls = (H >> 13).astype(np.int64)
sizes = np.bincount(cls[:cnt], minlength=2048)
badcls = {}
last = {}
for p in range(cnt):
    k = int(key[p]); q = last.get(k); last[k] = p
    want_has = q is not None and p - q <= 32768
    iv = int(inv[p])
    ok = (iv == 0xFFFFFFFF) if not want_has else (iv != 0xFFFFFFFF and int(sd[iv & 0x1FFFF]) == p - q)
    if not ok: badcls[int(cls[p])] = badcls.get(int(cls[p]), 0) + 1
print("classes with errors (class: errors / size):", sorted((c, e, int(sizes[c])) for c, e in badcls.items())[:40])
print("class sizes > 1024:", [(int(c), int(sizes[c])) for c in np.nonzero(sizes > 1024)[0]], "max", sizes.max())
