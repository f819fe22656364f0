"""Randomised run on the host calls that decode in pieces (streams of 8 MiB and more, zes_inflate / zes_inflate_alloc):
streams of this library, damaged, truncated and lengthened copies, other encoders' streams, tight capacities — against
the source bytes and the oracle's inflate.  Not a pytest; run on the GPU box.  usage: gpu_fuzz_host_big.py [seconds] [seed]"""
import ctypes as C
import os, sys, time, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import numpy as np
import _oracle as oracle
z = ge.load(); z.init(0)
L = z.lib()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
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


def check(tag, comp, exp):
    global n_cases
    got = gpu_inflate(comp)
    t = z.last_inflate_tier(); tiers[t] = tiers.get(t, 0) + 1
    if got != exp:
        name = os.path.join(ROOT, "gpurun_out", "fuzzhost_fail_%d.bin" % n_cases)
        os.makedirs(os.path.dirname(name), exist_ok=True)
        open(name, "wb").write(np.asarray(comp).tobytes())
        print("MISMATCH", tag, name, "tier", t, got[0], exp[0], (got[1] if got[0] == "err" else len(got[1])),
              (exp[1] if exp[0] == "err" else len(exp[1])), flush=True)
        raise SystemExit(1)
    n_cases += 1


while time.time() < t_end:
    kind = ("xorshift", "itext", "lowent4k")[int(rng.integers(3))]
    # compressed sizes around the pipeline's thresholds (8 MiB of stream; outputs of 56 MiB and more get a piece more)
    target_c = int(rng.choice([9, 12, 20, 30, 50, 70])) << 20
    ratio = {"xorshift": 1.0, "itext": 0.35, "lowent4k": 0.04}[kind]
    n = min(int(target_c / ratio), 160 << 20) + int(rng.integers(0, 300000))
    a = z.gen(kind, int(rng.integers(1 << 30)), n)
    comp = z.deflate(a)
    src = a.tobytes()
    if len(comp) >= (8 << 20):
        check("own", comp, ("out", src))
        assert z.last_inflate_tier() == 1, (kind, n)
        # capacities: exact, and one byte short (the size comes back)
        back = np.zeros(n, dtype=np.uint8)
        blen = C.c_uint64()
        assert L.zes_inflate(comp.ctypes.data, len(comp), back.ctypes.data, n, C.byref(blen), 0) == 0 and blen.value == n and back.tobytes() == src
        assert L.zes_inflate(comp.ctypes.data, len(comp), back.ctypes.data, n - 1, C.byref(blen), 0) == -16 and blen.value == n
        # bytes behind the end
        check("longer", np.concatenate([comp, np.frombuffer(os.urandom(int(rng.integers(1, 2 << 20))), dtype=np.uint8)]), ("out", src))
        # truncated / damaged: what the reference does (the oracle needs ~1 s per 100 MiB: a few of these only)
        if n <= (48 << 20) or rng.integers(3) == 0:
            bad = comp.copy()
            if rng.integers(2):
                bad = bad[:int(rng.integers(len(bad) // 2, len(bad)))]
            else:
                bad[int(rng.integers(2, len(bad)))] ^= np.uint8(1 << int(rng.integers(8)))
            check("damaged", bad, ref_inflate(bad))
    # another encoder's stream of that size
    if rng.integers(2) == 0:
        m = min(n, 64 << 20)
        other = np.frombuffer(pz.compress(src[:m], int(rng.choice([1, 6]))), dtype=np.uint8).copy()
        check("foreign", other, ("out", src[:m]))
    print("cases %d tiers %s (last: %s n=%d c=%d)" % (n_cases, tiers, kind, n, len(comp)), flush=True)
print("fuzz ok: %d cases, tiers %s" % (n_cases, tiers), flush=True)
