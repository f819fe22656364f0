"""Randomised differential run on the GPU box: deflate vs the oracle, inflate of own / foreign / damaged streams vs the
oracle, single and batch entry points.  usage: gpu_fuzz.py [seconds] [seed]
(FUZZ_DAMAGE=2: several bit flips and overwritten spans per damaged copy; FUZZ_MAX_CASES=n: stop after n cases — the
fixed-seed slice tests/test_gpu_fuzz.py runs in the suite)"""
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
n_cases = n_batch = 0
max_cases = int(os.environ.get("FUZZ_MAX_CASES", "0"))

def ref_inflate(comp):
    try:
        return ("out", oracle.inflate(comp).tobytes())
    except oracle.OracleError as ex:
        return ("err", ex.code)

def gpu_inflate(comp, flags=0):
    try:
        return ("out", z.inflate(comp, flags).tobytes())
    except z.ZlibEsError as ex:
        return ("err", ex.code)

while time.time() < t_end and (not max_cases or n_cases < max_cases):
    kind = kinds[int(rng.integers(3))]
    n = int(rng.choice([2, 3, 100, 4097, 65536, 131071, 131072, 131074, 262144 + 5, 400000, 1 << 20, 3000000, int(rng.integers(2, 2500000))]))
    if n % 131072 in (0, 1) and n < 3:
        n = 5
    if n % 131072 == 1:
        n += 1
    a = z.gen(kind, int(rng.integers(1 << 30)), n)
    shape = int(rng.integers(6))
    if os.environ.get("FUZZ_CASES"):  # name every case on stderr (beside ZES_TRACE_KERNELS: which call a GPU fault belongs to)
        print("case", n_cases, kind, n, "shape", shape, file=sys.stderr, flush=True)
    if shape == 0:  # runs and short periods: maximal matches at tiny distances, chunked code-length runs
        per = int(rng.choice([1, 2, 3, 5, 27, 257, 258, 259, 4097]))
        a = np.resize(a[:per], n).copy()
    elif shape == 1 and n > 1000:  # a few distinct byte values only
        a = (a & np.uint8(int(rng.choice([1, 3, 15])))).copy()
    # own deflate vs the oracle (sizes the oracle finishes quickly)
    if n <= 1500000:
        exp = oracle.deflate(a).tobytes()
        got = z.deflate(a).tobytes()
        assert got == exp, ("deflate", kind, n)
        comp = np.frombuffer(exp, dtype=np.uint8)
    else:
        comp = z.deflate(a)
    assert z.inflate(comp).tobytes() == a.tobytes(), ("inflate own", kind, n)
    # another encoder's stream, random settings
    level = int(rng.integers(0, 10)); mem = int(rng.integers(1, 10)); wb = int(rng.integers(9, 16))
    strat = int(rng.choice([pz.Z_DEFAULT_STRATEGY, pz.Z_FILTERED, pz.Z_HUFFMAN_ONLY, pz.Z_RLE, pz.Z_FIXED]))
    co = pz.compressobj(level, pz.DEFLATED, wb, mem, strat)
    parts = []
    step = max(1, n // int(rng.integers(1, 6)))
    for o in range(0, n, step):
        parts.append(co.compress(a[o:o + step].tobytes()))
        if rng.integers(4) == 0:
            parts.append(co.flush(int(rng.choice([pz.Z_SYNC_FLUSH, pz.Z_FULL_FLUSH]))))
    parts.append(co.flush())
    fz = np.frombuffer(b"".join(parts), dtype=np.uint8)
    assert z.inflate(fz).tobytes() == a.tobytes(), ("inflate foreign", kind, n, level, mem, wb, strat)
    if os.environ.get("FUZZ_DUMP") and n_cases == int(os.environ["FUZZ_DUMP"]):
        od = os.path.join(ROOT, "gpurun_out")
        os.makedirs(od, exist_ok=True)
        np.save(os.path.join(od, "fuzz_dump_a.npy"), a)
        np.save(os.path.join(od, "fuzz_dump_comp.npy"), np.asarray(comp))
        np.save(os.path.join(od, "fuzz_dump_fz.npy"), fz)
    # damaged copies: same result as the oracle (error code or bytes), whichever tier ends up with it
    for src in (comp, fz):
        if len(src) < 8 or len(src) > 600000:
            continue
        bad = src.copy()
        pos = int(rng.integers(2, len(bad)))
        bad[pos] ^= np.uint8(1 << int(rng.integers(8)))
        if os.environ.get("FUZZ_DAMAGE") == "2":  # heavier damage (another sequence of cases per seed than the default)
            for _ in range(int(rng.integers(0, 4))):
                bad[int(rng.integers(2, len(bad)))] ^= np.uint8(1 << int(rng.integers(8)))
            if rng.integers(5) == 0:  # a span overwritten: zeros, ones or a copy from elsewhere in the stream
                ln = int(rng.integers(1, min(300, len(bad) - 2)))
                at = int(rng.integers(2, len(bad) - ln))
                kind2 = int(rng.integers(3))
                if kind2 == 0:
                    bad[at:at + ln] = 0
                elif kind2 == 1:
                    bad[at:at + ln] = 255
                else:
                    frm = int(rng.integers(0, len(bad) - ln))
                    bad[at:at + ln] = src[frm:frm + ln]
        if rng.integers(3) == 0:
            bad = bad[:int(rng.integers(2, len(bad)))]
        if os.environ.get("FUZZ_TRACE"):
            print("damaged", kind, n, pos, len(bad), "cases", n_cases, flush=True)
            if n_cases >= int(os.environ["FUZZ_TRACE"]):
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "fuzz_last.bin"), "wb").write(bad.tobytes())
        t0 = time.time()
        got, exp = gpu_inflate(bad), ref_inflate(bad)
        if got != exp:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            name = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d.bin" % n_cases)
            open(name, "wb").write(bad.tobytes())
            desc = lambda r: (r[0], len(r[1]) if r[0] == "out" else r[1])
            print("MISMATCH", name, "tier", z.last_inflate_tier(), "gpu", desc(got), "oracle", desc(exp), flush=True)
            if got[0] == exp[0] == "out":
                a1, a2 = np.frombuffer(got[1], dtype=np.uint8), np.frombuffer(exp[1], dtype=np.uint8)
                m = min(len(a1), len(a2))
                d = np.nonzero(a1[:m] != a2[:m])[0]
                print("   first difference at", int(d[0]) if len(d) else None, "of", len(a1), len(a2), flush=True)
            raise SystemExit(1)
        if time.time() - t0 > 5:
            print("slow damaged case: %.1f s" % (time.time() - t0), kind, n, pos, len(bad), flush=True)
    # device entry point on a view: random cut with the rest of the stream live behind it
    for src in (comp, fz):
        if len(src) < 16 or len(src) > 400000:
            continue
        d = torch.from_numpy(np.ascontiguousarray(src)).cuda()
        c = int(rng.integers(2, len(src) + 1)) if rng.integers(2) else len(src) - int(rng.integers(0, 10))
        exp = ref_inflate(np.ascontiguousarray(src[:c]))
        out = torch.empty((len(exp[1]) if exp[0] == "out" else n) + int(rng.integers(0, 64)), dtype=torch.uint8, device="cuda")
        try:
            got = ("out", z.inflate_tensor(d[:c], out).cpu().numpy().tobytes())
        except z.ZlibEsError as ex:
            got = ("err", ex.code)
        if got != exp:
            name = os.path.join(ROOT, "gpurun_out", "fuzz_view_%d.bin" % n_cases)
            os.makedirs(os.path.dirname(name), exist_ok=True)
            open(name, "wb").write(np.ascontiguousarray(src).tobytes())
            print("MISMATCH view", name, "c", c, "tier", z.last_inflate_tier(), got[0], exp[0], (got[1] if got[0] == "err" else len(got[1])),
                  (exp[1] if exp[0] == "err" else len(exp[1])), flush=True)
            raise SystemExit(1)
    # raw forms with an offset
    if n <= 1500000 and rng.integers(3) == 0:
        raw = z.deflate_raw(a)
        assert raw.tobytes() == oracle.deflate_raw(a).tobytes(), ("deflate_raw", kind, n)
        k = int(rng.integers(0, 40))
        boxed = np.concatenate([rng.integers(0, 256, k, dtype=np.uint8), raw, rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)])
        assert z.inflate_raw(boxed, k).tobytes() == a.tobytes(), ("inflate_raw", kind, n, k)
    n_cases += 1
    # now and then: a batch of mixed streams through the batch entry point
    if n_cases % 5 == 0:
        raws, comps = [], []
        for i in range(int(rng.integers(2, 24))):
            m = int(rng.integers(2, 600000))
            if m % 131072 == 1:
                m += 1
            r = z.gen(kinds[int(rng.integers(3))], int(rng.integers(1 << 30)), m)
            raws.append(r)
            comps.append(oracle.deflate(r) if rng.integers(2) else np.frombuffer(pz.compress(r.tobytes(), int(rng.integers(0, 10))), dtype=np.uint8))
        in_off, out_off, pos, opos = [], [], 0, 0
        for r, cdat in zip(raws, comps):
            in_off.append(pos); pos += (len(cdat) + 15) // 16 * 16
            out_off.append(opos); opos += (len(r) + 15) // 16 * 16
        big = np.zeros(pos, dtype=np.uint8)
        for cdat, o in zip(comps, in_off):
            big[o:o + len(cdat)] = cdat
        d_out = torch.zeros(opos, dtype=torch.uint8, device="cuda")
        olen, st = z.inflate_batch_tensor(torch.from_numpy(big).cuda(), in_off, [len(x) for x in comps], d_out, out_off, [len(r) for r in raws])
        host = d_out.cpu().numpy()
        for i, r in enumerate(raws):
            assert st[i] == 0 and olen[i] == len(r) and (host[out_off[i]:out_off[i] + len(r)] == r).all(), ("batch", i, len(r))
        # the same batch with some members damaged: status (or bytes) per member as the oracle has them
        dcomps = []
        for cdat in comps:
            cdat = np.array(cdat, copy=True)
            if rng.integers(3) == 0 and len(cdat) > 8:
                cdat[int(rng.integers(2, len(cdat)))] ^= np.uint8(1 << int(rng.integers(8)))
            dcomps.append(cdat)
        big2 = np.zeros(pos, dtype=np.uint8)
        for cdat, o in zip(dcomps, in_off):
            big2[o:o + len(cdat)] = cdat
        caps = [len(r) + 70000 for r in raws]
        out_off2, opos2 = [], 0
        for cp in caps:
            out_off2.append(opos2); opos2 += (cp + 15) // 16 * 16
        d_out2 = torch.zeros(opos2, dtype=torch.uint8, device="cuda")
        olen2, st2 = z.inflate_batch_tensor(torch.from_numpy(big2).cuda(), in_off, [len(x) for x in dcomps], d_out2, out_off2, caps)
        host2 = d_out2.cpu().numpy()
        for i, cdat in enumerate(dcomps):
            exp = ref_inflate(cdat)
            if exp[0] == "err":
                assert st2[i] == exp[1], ("batch damaged status", i, st2[i], exp[1])
            elif len(exp[1]) > caps[i]:
                assert st2[i] == -16 and olen2[i] == len(exp[1]), ("batch damaged nospace", i, st2[i], olen2[i], len(exp[1]))
            else:
                assert st2[i] == 0 and olen2[i] == len(exp[1]) and host2[out_off2[i]:out_off2[i] + olen2[i]].tobytes() == exp[1], ("batch damaged out", i, st2[i], olen2[i], len(exp[1]))
        n_batch += 1
    if n_cases % 20 == 0:
        print("cases %d batches %d" % (n_cases, n_batch), flush=True)
print("fuzz ok: %d cases, %d batches" % (n_cases, n_batch), flush=True)
