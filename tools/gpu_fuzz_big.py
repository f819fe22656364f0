"""Randomised run on multi-MiB streams (not a pytest; run on the GPU box): the segment-parallel tier on streams of
another encoder with random settings, flush points, damage and truncation, against the oracle's inflate.
usage: gpu_fuzz_big.py [seconds] [seed]"""
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import numpy as np, torch
import _oracle as oracle
z = ge.load(); z.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kinds = ("itext", "lowent4k", "xorshift")
t_end = time.time() + budget
n_cases = 0
tiers = {}

def ref_inflate(comp):
    try:
        return ("out", oracle.inflate(comp).tobytes())
    except oracle.OracleError as ex:
        return ("err", ex.code)

def gpu_inflate(comp):
    try:
        return ("out", z.inflate(comp).tobytes())
    except z.ZlibEsError as ex:
        return ("err", ex.code)

while time.time() < t_end:
    # a mixture: pieces of different kinds glued together
    pieces = []
    for _ in range(int(rng.integers(1, 6))):
        pieces.append(z.gen(kinds[int(rng.integers(3))], int(rng.integers(1 << 30)), int(rng.integers(1000, 6 << 20))))
    a = np.concatenate(pieces)
    n = len(a)
    level = int(rng.integers(1, 10)); mem = int(rng.integers(1, 10)); wb = int(rng.integers(9, 16))
    strat = int(rng.choice([pz.Z_DEFAULT_STRATEGY, pz.Z_DEFAULT_STRATEGY, pz.Z_FILTERED, pz.Z_HUFFMAN_ONLY, pz.Z_RLE]))
    co = pz.compressobj(level, pz.DEFLATED, wb, mem, strat)
    parts = []
    step = max(1, n // int(rng.integers(1, 9)))
    for o in range(0, n, step):
        parts.append(co.compress(a[o:o + step].tobytes()))
        if rng.integers(4) == 0:
            parts.append(co.flush(int(rng.choice([pz.Z_SYNC_FLUSH, pz.Z_FULL_FLUSH]))))
    parts.append(co.flush())
    fz = np.frombuffer(b"".join(parts), dtype=np.uint8)
    if os.environ.get("ZES_FUZZ_TRACE"):  # the input of the call that is about to run, should the GPU fault in it
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "fuzzbig_last.bin"), "wb").write(fz.tobytes())
        print("case %d: n=%d c=%d level=%d mem=%d wb=%d strat=%d" % (n_cases, n, len(fz), level, mem, wb, strat), flush=True)
    got = gpu_inflate(fz)
    t = z.last_inflate_tier(); tiers[t] = tiers.get(t, 0) + 1
    assert got == ("out", a.tobytes()), ("foreign", n, level, mem, wb, strat, t)
    # one damaged / truncated variant against the oracle
    bad = fz.copy()
    pos = int(rng.integers(2, len(bad)))
    bad[pos] ^= np.uint8(1 << int(rng.integers(8)))
    if rng.integers(3) == 0:
        bad = bad[:int(rng.integers(2, len(bad)))]
    if os.environ.get("ZES_FUZZ_TRACE"):
        open(os.path.join(ROOT, "gpurun_out", "fuzzbig_last.bin"), "wb").write(bad.tobytes())
        print("case %d: damaged, c=%d" % (n_cases, len(bad)), flush=True)
    got, exp = gpu_inflate(bad), ref_inflate(bad)
    if got != exp:
        name = os.path.join(ROOT, "gpurun_out", "fuzzbig_fail_%d.bin" % n_cases)
        os.makedirs(os.path.dirname(name), exist_ok=True)
        open(name, "wb").write(bad.tobytes())
        print("MISMATCH", name, "tier", z.last_inflate_tier(), got[0], exp[0], (got[1] if got[0] == "err" else len(got[1])),
              (exp[1] if exp[0] == "err" else len(exp[1])), flush=True)
        raise SystemExit(1)
    n_cases += 1
    if n_cases % 10 == 0:
        print("cases %d tiers %s" % (n_cases, tiers), flush=True)
print("fuzz ok: %d cases, tiers %s" % (n_cases, tiers), flush=True)
