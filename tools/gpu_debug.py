"""Ad-hoc GPU bring-up script (not collected by pytest): stage-by-stage comparison with the oracle."""
import os, sys, time, json, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import _oracle as O
import torch
z = ge.load(); z.init(0)
print(z.device_info(), flush=True)

def check(name, cond, extra=""):
    print(("PASS " if cond else "FAIL ") + name + " " + extra, flush=True)
    return cond

# adler
for n in [1, 15, 4096, 70000, 1 << 20, (1 << 22) + 13]:
    a = z.gen("xorshift", 7, n); t = torch.from_numpy(a).cuda()
    check("adler n=%d" % n, z.adler32_tensor(t) == O.adler32(a))
# huffman stage
hf = json.load(open(os.path.join(ROOT, "tests/golden/huffman.json")))
bad = 0
for e in hf:
    l = z.stage_huff_lengths(e["hist"], e["maxlen"])
    if list(l) != e["lens"]:
        bad += 1
        if bad < 4: print("huff mismatch", e["maxlen"], len(e["hist"]), list(l)[:40], e["lens"][:40])
check("huffman lengths (%d cases)" % len(hf), bad == 0)
# lz77 stage
for kind, seed, n, start, ln in [("itext", 7, 3000, 0, 3000), ("lowent4k", 3, 9000, 0, 9000), ("xorshift", 9, 2000, 0, 2000),
                                 ("itext", 12345, 300000, 0, 131072), ("itext", 12345, 300000, 131072, 131072),
                                 ("itext", 12345, 300000, 262144, 37856), ("lowent4k", 12345, 300000, 131072, 131072)]:
    a = z.gen(kind, seed, n); t = torch.from_numpy(a).cuda()
    got = z.stage_lz77_tensor(t, start, ln); want = O.lz77_block(a, start, ln)
    ok = len(got) == len(want) and (got == want).all()
    extra = ""
    if not ok:
        m = min(len(got), len(want)); d = np.nonzero(got[:m] != want[:m])[0]
        extra = "ntok %d vs %d first diff %s got %s want %s" % (len(got), len(want), d[:1], [hex(x) for x in got[d[:1][0]:d[:1][0]+3]] if len(d) else "", [hex(x) for x in want[d[:1][0]:d[:1][0]+3]] if len(d) else "")
    check("lz77 %s n=%d start=%d" % (kind, n, start), ok, extra)
z0 = np.zeros(70000, dtype=np.uint8); t = torch.from_numpy(z0).cuda()
got = z.stage_lz77_tensor(t, 0, 70000); want = O.lz77_block(z0, 0, 70000)
check("lz77 zeros", len(got) == len(want) and (got == want).all())
# full deflate
for kind, seed, n in [("itext", 1, 2), ("itext", 1, 100), ("xorshift", 2, 4096), ("itext", 3, 65536), ("lowent4k", 4, 131072),
                      ("itext", 12345, 300000), ("xorshift", 12345, 1 << 20), ("lowent4k", 12345, 1 << 20), ("itext", 12345, 1 << 20)]:
    a = z.gen(kind, seed, n); t = torch.from_numpy(a).cuda()
    try:
        comp = z.deflate_tensor(t).cpu().numpy()
    except Exception as ex:
        check("deflate %s n=%d" % (kind, n), False, repr(ex)); continue
    want = O.deflate(a)
    ok = comp.shape == want.shape and (comp == want).all()
    extra = ""
    if not ok:
        m = min(len(comp), len(want)); d = np.nonzero(comp[:m] != want[:m])[0]
        extra = "len %d vs %d first diff at %s" % (len(comp), len(want), d[:3])
    check("deflate %s n=%d" % (kind, n), ok, extra)
    # inflate of the oracle's stream
    wt = torch.from_numpy(want).cuda()
    for flags, nm in [(0, "fast"), (1, "seq")]:
        out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        try:
            back = z.inflate_tensor(wt, out, flags)
            ok = back.numel() == n and bool((back.cpu().numpy() == a).all())
            check("inflate[%s] %s n=%d" % (nm, kind, n), ok, "got %d bytes" % back.numel())
        except Exception as ex:
            check("inflate[%s] %s n=%d" % (nm, kind, n), False, repr(ex))
# KATs + malformed streams through the host API
kat = json.load(open(os.path.join(ROOT, "tests/golden/kat.json")))
for k in ["UNCOMPRESSED", "FIXED", "DYNAMIC"]:
    try:
        o = z.inflate(bytes.fromhex(kat["kat"][k]))
        check("KAT " + k, o.tobytes().hex() == kat["kat"]["RAW"])
    except Exception as ex:
        check("KAT " + k, False, repr(ex))
cases = json.load(open(os.path.join(ROOT, "tests/golden/inflate_cases.json")))
bad = 0; t0 = time.time()
for e in cases[::3]:
    try:
        o = z.inflate(bytes.fromhex(e["input"])); got = ("out", o.tobytes().hex())
    except z.ZlibEsError as ex:
        got = ("err", str(ex))
    exp = ("err", e["error"]) if "error" in e else ("out", e["output"])
    if got != exp:
        bad += 1
        if bad < 8: print("  MISMATCH", e["name"], exp[0], exp[1][:50], "| got", got[0], got[1][:50], flush=True)
check("malformed/foreign inflate cases (%d)" % len(cases[::3]), bad == 0, "bad=%d %.1fs" % (bad, time.time() - t0))
rd = open(os.path.join(ROOT, "tests/golden/ref_data/compressed.bin"), "rb").read()
raw = open(os.path.join(ROOT, "tests/golden/ref_data/raw.bin"), "rb").read()
t0 = time.time(); o = z.inflate(rd); check("compressed.bin -> raw.bin", o.tobytes() == raw, "%.2fs" % (time.time() - t0))
o = z.deflate(raw); check("deflate(raw.bin)", (o == O.deflate(raw)).all())
# timing at 64 MiB
z.set_profiling(True)
for kind in ["xorshift", "itext"]:
    n = 64 << 20
    a = z.gen(kind, 12345, n); t = torch.from_numpy(a).cuda()
    out = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time(); comp = z.deflate_tensor(t, out); dt = time.time() - t0
        print("deflate 64MiB %s: %.2f ms  %.2f GiB/s  c=%d" % (kind, dt * 1e3, n / dt / 2**30, comp.numel()), z.last_kernel_times(), flush=True)
    sha = hashlib.sha256(comp.cpu().numpy().tobytes()).hexdigest()
    m = json.load(open(os.path.join(ROOT, "tests/golden/manifest.json")))
    exp = [e for e in m["big"] if e["kind"] == kind][0]
    check("deflate 64MiB %s sha" % kind, sha == exp["deflate_sha256"] and comp.numel() == exp["deflate_len"])
    cc = comp.clone(); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(2):
        t0 = time.time(); b = z.inflate_tensor(cc, back); dt = time.time() - t0
        print("inflate 64MiB %s: %.2f ms  %.2f GiB/s" % (kind, dt * 1e3, n / dt / 2**30), z.last_kernel_times(), flush=True)
    check("inflate 64MiB %s roundtrip" % kind, b.numel() == n and bool((b == t).all()))
print("done")
