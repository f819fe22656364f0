"""Parity tests proper: the HIP path, called through the C-ABI (include/zes.h), against the CPU
oracle on the same seeded inputs, against the committed golden fixtures, and — at
BASELINE.json's full 64 MiB size — through sha256 pins and the inflate(deflate(x)) == x
round trip.  Bit-exact everywhere: this path is integer/byte work."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from test_oracle_golden import make_input, sha

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


# ---------------------------------------------------------------------------------------------
# stage level
# ---------------------------------------------------------------------------------------------
def test_adler32_wavefront_reduction(z, oracle, gpu):
    for kind, n in [("xorshift", 1), ("xorshift", 15), ("itext", 4097), ("lowent4k", 65536), ("xorshift", 65537),
                    ("itext", (1 << 22) + 13)]:
        a = z.gen(kind, 77, n)
        assert z.adler32_tensor(dev(a, gpu)) == oracle.adler32(a)
    ff = np.full(1 << 20, 255, dtype=np.uint8)  # largest sums before the modulo
    assert z.adler32_tensor(dev(ff, gpu)) == oracle.adler32(ff)
    assert z.adler32(b"") == 1


def test_package_merge_lengths_kernel(z, oracle, gpu):
    for e in golden("huffman.json"):
        assert list(z.stage_huff_lengths(e["hist"], e["maxlen"])) == e["lens"]
    rng = np.random.default_rng(5)
    for t in range(40):
        nsym, L = [(286, 15), (30, 15), (19, 7)][t % 3]
        hist = (rng.integers(0, 4, nsym) * rng.integers(0, 2000, nsym) ** (t % 3)).astype(np.uint32)
        assert (z.stage_huff_lengths(hist, L) == oracle.huff_lengths(hist, L)).all()


def test_lz77_tokens_kernels(z, oracle, gpu):
    for e in golden("lz77.json"):
        a = make_input(z, e)
        tk = z.stage_lz77_tensor(dev(a, gpu), e["start"], e["len"])
        assert len(tk) == e["ntokens"] and sha(tk) == e["tokens_sha256"], e["kind"]
    for kind, seed, n, start, ln in [("itext", 5, 131072 * 2 + 9, 131072, 131072), ("xorshift", 6, 131072 + 2, 131072, 2),
                                     ("lowent4k", 7, 200000, 131072, 68928), ("itext", 8, 40000, 0, 40000)]:
        a = z.gen(kind, seed, n)
        got = z.stage_lz77_tensor(dev(a, gpu), start, ln)
        want = oracle.lz77_block(a, start, ln)
        assert len(got) == len(want) and (got == want).all()


# ---------------------------------------------------------------------------------------------
# deflate
# ---------------------------------------------------------------------------------------------
def test_deflate_small_vectors_host_api(z, gpu):
    kat = golden("kat.json")
    for name, v in kat["small"].items():
        out = z.deflate(bytes.fromhex(v["input"]))
        assert out.tobytes().hex() == v["deflate"], name


def test_deflate_throw_cases(z, gpu):
    for n in (0, 1, 131073, 262145):
        with pytest.raises(z.ZlibEsError, match="Data is corrupted"):
            z.deflate(z.gen("xorshift", 1, n))


def test_deflate_manifest(z, gpu):
    m = golden("manifest.json")
    for e in m["cases"]:
        if "error" in e:
            continue
        a = make_input(z, e)
        out = z.deflate_tensor(dev(a, gpu)).cpu().numpy()
        assert len(out) == e["deflate_len"] and sha(out) == e["deflate_sha256"], e
    for zc in m["zeros"]:
        out = z.deflate_tensor(dev(np.zeros(zc["n"], dtype=np.uint8), gpu)).cpu().numpy()
        assert len(out) == zc["deflate_len"] and sha(out) == zc["deflate_sha256"], zc


def test_deflate_vs_oracle_seeded(z, oracle, gpu):
    rng = np.random.default_rng(11)
    sizes = [2, 3, 4, 5, 17, 258, 259, 260, 300, 1000, 4095, 32768, 32769, 65535, 131071, 131072, 131074, 131075, 200001]
    for i, n in enumerate(sizes):
        kind = ["xorshift", "lowent4k", "itext"][i % 3]
        a = z.gen(kind, int(rng.integers(1, 2**31)), n)
        if i % 5 == 0 and n > 64:
            a = np.resize(a[: max(1, n // 40)], n).copy()  # short period: many candidates, long matches
        got = z.deflate_tensor(dev(a, gpu)).cpu().numpy()
        want = oracle.deflate(a)
        assert got.shape == want.shape and (got == want).all(), (kind, n)
    for val, n in [(0, 70000), (255, 131072 + 5000), (65, 258 * 3 + 7)]:  # runs: the 16/128-candidate early exit
        a = np.full(n, val, dtype=np.uint8)
        got = z.deflate_tensor(dev(a, gpu)).cpu().numpy()
        assert (got == oracle.deflate(a)).all()


def test_deflate_reference_binary_fixture(z, gpu):
    rd = golden("ref_data.json")
    raw = np.frombuffer(open(os.path.join(GOLDEN, "ref_data", "raw.bin"), "rb").read(), dtype=np.uint8)
    out = z.deflate(raw)
    assert len(out) == rd["deflate_of_raw_len"] and sha(out) == rd["deflate_of_raw_sha256"]


def test_deflate_batch_api(z, oracle, gpu):
    import torch

    specs = [("itext", 1, 1 << 20), ("xorshift", 2, 131074), ("lowent4k", 3, 300000), ("itext", 4, 2), ("xorshift", 5, 131073),
             ("itext", 6, 65536)]
    bufs = [z.gen(k, s, n) for k, s, n in specs]
    in_off, out_off, caps = [], [], []
    pos = opos = 0
    for b in bufs:
        in_off.append(pos)
        pos += (len(b) + 15) // 16 * 16
        caps.append(z.deflate_bound(len(b)))
        out_off.append(opos)
        opos += (caps[-1] + 15) // 16 * 16
    big = np.zeros(pos, dtype=np.uint8)
    for b, o in zip(bufs, in_off):
        big[o:o + len(b)] = b
    d_in = dev(big, gpu)
    d_out = torch.zeros(opos, dtype=torch.uint8, device=gpu)
    cnt = len(bufs)
    arr = lambda v: (C.c_uint64 * cnt)(*v)
    out_len = (C.c_uint64 * cnt)()
    status = (C.c_int32 * cnt)()
    rc = z.lib().zes_deflate_batch_dev(d_in.data_ptr(), arr(in_off), arr([len(b) for b in bufs]), d_out.data_ptr(), arr(out_off),
                                       arr(caps), out_len, status, cnt)
    assert rc == 0
    host = d_out.cpu().numpy()
    for i, b in enumerate(bufs):
        if len(b) % 131072 == 1:
            assert status[i] == -3
            continue
        assert status[i] == 0
        want = oracle.deflate(b)
        assert out_len[i] == len(want) and (host[out_off[i]:out_off[i] + len(want)] == want).all(), specs[i]


def test_inflate_batch_api(z, oracle, gpu):
    import torch

    import zlib as pz

    specs = [("itext", 61, 300000), ("xorshift", 62, 131072), ("lowent4k", 63, 500000), ("itext", 64, 2),
             ("itext", 65, 1 << 20), ("itext", 66, 700001)]
    raws = [z.gen(k, s, n) for k, s, n in specs]
    comps = [oracle.deflate(r) for r in raws[:4]]
    # two streams of another encoder in the same call: the segment-parallel tier next to the block-parallel one
    comps += [np.frombuffer(pz.compress(r.tobytes(), lvl), dtype=np.uint8) for r, lvl in zip(raws[4:], (6, 1))]
    comps.append(np.frombuffer(bytes([0x77, 0x9C, 1, 2, 3]), dtype=np.uint8))  # not deflate
    comps.append(np.frombuffer(bytes([0x78, 0x9C, 7, 0, 0, 0]), dtype=np.uint8))  # BTYPE 3
    in_off, out_off, caps = [], [], []
    pos = opos = 0
    for i, cdat in enumerate(comps):
        in_off.append(pos)
        pos += (len(cdat) + 15) // 16 * 16
        cap = len(raws[i]) if i < len(raws) else 64
        caps.append(cap)
        out_off.append(opos)
        opos += (cap + 15) // 16 * 16
    big = np.zeros(pos, dtype=np.uint8)
    for cdat, o in zip(comps, in_off):
        big[o:o + len(cdat)] = cdat
    d_in = dev(big, gpu)
    d_out = torch.zeros(opos, dtype=torch.uint8, device=gpu)
    cnt = len(comps)
    arr = lambda v: (C.c_uint64 * cnt)(*v)
    out_len = (C.c_uint64 * cnt)()
    status = (C.c_int32 * cnt)()
    rc = z.lib().zes_inflate_batch_dev(d_in.data_ptr(), arr(in_off), arr([len(x) for x in comps]), d_out.data_ptr(), arr(out_off),
                                       arr(caps), out_len, status, cnt, 0)
    assert rc == 0
    host = d_out.cpu().numpy()
    for i, r in enumerate(raws):
        assert status[i] == 0 and out_len[i] == len(r)
        assert (host[out_off[i]:out_off[i] + len(r)] == r).all(), specs[i]
    assert status[len(raws)] == -1 and status[len(raws) + 1] == -2


def test_inflate_batch_of_many_foreign_streams(z, oracle, gpu):
    """A batch of another encoder's streams: the left-over streams of a call are decoded side by side (one serial
    wavefront each); a malformed one among them still ends as the reference does."""
    import torch
    import zlib as pz

    raws, comps = [], []
    for i in range(40):
        r = z.gen(("itext", "lowent4k", "xorshift")[i % 3], 900 + i, 5000 + 37111 * (i % 7) + 1000 * i)
        raws.append(r)
        comps.append(np.frombuffer(pz.compress(r.tobytes(), (1, 6, 9)[i % 3]), dtype=np.uint8))
    raws.append(z.gen("itext", 990, 400000))
    comps.append(oracle.deflate(raws[-1]))  # one the reference wrote, among them
    bad = bytearray(comps[1].tobytes())
    bad[len(bad) // 2] ^= 0x55  # damaged in the middle: error or other bytes, whatever the reference makes of it
    try:
        exp_bad = ("out", oracle.inflate(np.frombuffer(bytes(bad), dtype=np.uint8)).tobytes())
    except oracle.OracleError as ex:
        exp_bad = ("err", ex.code)
    comps.append(np.frombuffer(bytes(bad), dtype=np.uint8))
    cnt = len(comps)
    in_off, out_off, caps = [], [], []
    pos = opos = 0
    for i, cdat in enumerate(comps):
        in_off.append(pos)
        pos += (len(cdat) + 15) // 16 * 16
        cap = len(raws[i]) if i < len(raws) else 1 << 20
        caps.append(cap)
        out_off.append(opos)
        opos += (cap + 15) // 16 * 16
    big = np.zeros(pos, dtype=np.uint8)
    for cdat, o in zip(comps, in_off):
        big[o:o + len(cdat)] = cdat
    d_out = torch.zeros(opos, dtype=torch.uint8, device=gpu)
    olen, st = z.inflate_batch_tensor(dev(big, gpu), in_off, [len(x) for x in comps], d_out, out_off, caps)
    host = d_out.cpu().numpy()
    for i, r in enumerate(raws):
        assert st[i] == 0 and olen[i] == len(r), i
        assert (host[out_off[i]:out_off[i] + len(r)] == r).all(), i
    if exp_bad[0] == "err":
        assert st[cnt - 1] == exp_bad[1]
    else:
        assert st[cnt - 1] == 0 and host[out_off[cnt - 1]:out_off[cnt - 1] + olen[cnt - 1]].tobytes() == exp_bad[1]


def test_raw_deflate_and_offset_inflate_entry_points(z, oracle, gpu):
    """src/deflate.ts:14 and src/inflate.ts:16: the raw forms the zlib wrapper of src/zlib.ts encloses."""
    for kind, seed, n in (("itext", 71, 200000), ("xorshift", 72, 131073 + 5), ("lowent4k", 73, 300001), ("itext", 74, 2)):
        a = z.gen(kind, seed, n)
        raw = z.deflate_raw(a)
        assert raw.tobytes() == oracle.deflate_raw(a).tobytes() == oracle.deflate(a)[2:-4].tobytes()
        assert z.inflate_raw(raw).tobytes() == a.tobytes()
        # embedded in a container: 7 foreign bytes in front, 9 behind (stay readable, like the zlib trailer)
        boxed = np.concatenate([np.arange(7, dtype=np.uint8), raw, np.full(9, 0xEE, dtype=np.uint8)])
        assert z.inflate_raw(boxed, 7).tobytes() == oracle.inflate_raw(boxed, 7).tobytes() == a.tobytes()
    # a raw stream of another encoder (history across blocks), at an odd offset
    import zlib as pz

    a = z.gen("itext", 75, 900000)
    co = pz.compressobj(6, pz.DEFLATED, -15)
    raw = np.frombuffer(co.compress(a.tobytes()) + co.flush(), dtype=np.uint8)
    boxed = np.concatenate([np.arange(3, dtype=np.uint8), raw])
    assert z.inflate_raw(boxed, 3).tobytes() == a.tobytes() and z.last_inflate_tier() == 2
    with pytest.raises(z.ZlibEsError, match="Data is corrupted"):
        z.deflate_raw(np.zeros(1, dtype=np.uint8))
    # the raw path has no CM-nibble check: garbage is reported by the decoder itself, as by the reference
    junk = np.frombuffer(bytes([0x07, 0, 0, 0]), dtype=np.uint8)  # BFINAL=1, BTYPE=3
    with pytest.raises(z.ZlibEsError, match="Not supported BTYPE : 3"):
        z.inflate_raw(junk)
    with pytest.raises(oracle.OracleError):
        oracle.inflate_raw(junk)


def test_deflate_capacity_is_checked(z, gpu):
    import torch

    a = dev(z.gen("itext", 1, 100000), gpu)
    out = torch.empty(1000, dtype=torch.uint8, device=gpu)
    n = C.c_uint64()
    assert z.lib().zes_deflate_dev(a.data_ptr(), a.numel(), out.data_ptr(), out.numel(), C.byref(n)) == z.ZES_E_NOSPACE
    assert n.value == z.deflate_bound(100000)


# ---------------------------------------------------------------------------------------------
# inflate
# ---------------------------------------------------------------------------------------------
def test_inflate_reference_suite_vectors(z, gpu):
    kat = golden("kat.json")
    for k in ("UNCOMPRESSED", "FIXED", "DYNAMIC"):  # test/index.js:16-35
        assert z.inflate(bytes.fromhex(kat["kat"][k])).tobytes().hex() == kat["kat"]["RAW"]
    raw = open(os.path.join(GOLDEN, "ref_data", "raw.bin"), "rb").read()
    comp = open(os.path.join(GOLDEN, "ref_data", "compressed.bin"), "rb").read()
    assert z.inflate(comp).tobytes() == raw  # test/index.js:37-42 (foreign stream, cross-block history)


def test_inflate_malformed_streams_raise_the_reference_error(z, gpu):
    cases = golden("inflate_cases.json")
    for e in cases:
        try:
            got = ("out", z.inflate(bytes.fromhex(e["input"])).tobytes().hex())
        except z.ZlibEsError as ex:
            got = ("err", str(ex))
        exp = ("err", e["error"]) if "error" in e else ("out", e["output"])
        assert got == exp, e["name"]


def test_inflate_truncated_stream_the_reference_never_finishes(z, gpu):
    """tests/test_oracle_golden.py::test_truncated_stream_the_reference_never_finishes, through the product path:
    the decode must end (with 'Lack of data length'), whatever capacity the caller offers."""
    import torch

    data = open(os.path.join(GOLDEN, "truncated_runaway.zz"), "rb").read()
    with pytest.raises(z.ZlibEsError, match="Lack of data length"):
        z.inflate(data)
    with pytest.raises(z.ZlibEsError, match="Lack of data length"):
        z.inflate_tensor(dev(np.frombuffer(data, dtype=np.uint8), gpu), torch.empty(1 << 20, dtype=torch.uint8, device=gpu))


def test_inflate_truncated_inside_the_last_code_with_live_bytes_behind(z, oracle, gpu):
    """The device entry point gets a length, not a zero-padded copy: what lies behind the stream in memory must not
    count.  Here the bytes behind a truncated stream are its own continuation, so a decoder that reads on finds the
    end-of-block code it is missing (found by tests/gpu_fuzz.py: the block decoder tested 'end of block' before
    'behind the data').  Every length around the end of the stream must end as the reference does."""
    import torch

    for kind, n in (("lowent4k", 65536), ("itext", 131072), ("xorshift", 40000), ("itext", 300000)):
        a = z.gen(kind, 77, n)
        full = oracle.deflate(a)
        d = dev(full, gpu)
        for c in range(len(full) - 12, len(full) + 1):
            try:
                exp = ("out", oracle.inflate(full[:c]).tobytes())
            except oracle.OracleError as ex:
                exp = ("err", ex.code)
            out = torch.empty(n + 4096, dtype=torch.uint8, device=gpu)  # (a cut stream can come out a few bytes longer)
            try:
                got = ("out", z.inflate_tensor(d[:c], out).cpu().numpy().tobytes())
            except z.ZlibEsError as ex:
                got = ("err", ex.code)
            assert got == exp, (kind, n, c, len(full))


def test_inflate_foreign_streams(z, gpu):
    for f in golden("foreign.json"):
        comp = open(os.path.join(GOLDEN, f["file"]), "rb").read()
        assert sha(z.inflate(comp)) == f["output_sha256"], f["file"]


def test_inflate_all_tiers_agree(z, oracle, gpu):
    import torch

    for kind, seed, n in [("itext", 21, 2), ("itext", 22, 131072), ("xorshift", 23, 131072 * 3 + 1234), ("lowent4k", 24, 1 << 20),
                          ("itext", 25, (1 << 21) + 77)]:
        a = z.gen(kind, seed, n)
        comp = dev(oracle.deflate(a), gpu)
        for flags in (0, z.ZES_F_NO_FASTPATH):
            out = torch.empty(n, dtype=torch.uint8, device=gpu)
            back = z.inflate_tensor(comp, out, flags)
            assert back.numel() == n and (back.cpu().numpy() == a).all(), (kind, n, flags)


def test_reference_made_streams_take_the_block_parallel_tier(z, oracle, gpu):
    """Speed regressions hide behind the tiers (a T1 bug silently becomes a slow T2 run): pin the tier."""
    import torch

    for kind, seed, n in [("itext", 51, 131072 * 5 + 999), ("xorshift", 52, 131072 * 3), ("lowent4k", 53, 700000),
                          ("itext", 54, 3000), ("itext", 55, (4 << 20) + 17)]:
        a = z.gen(kind, seed, n)
        comp = dev(oracle.deflate(a), gpu)
        out = torch.empty(n, dtype=torch.uint8, device=gpu)
        back = z.inflate_tensor(comp, out)
        assert back.numel() == n and (back.cpu().numpy() == a).all()
        assert z.last_inflate_tier() == 1, (kind, n)
    raw = open(os.path.join(GOLDEN, "ref_data", "compressed.bin"), "rb").read()
    z.inflate(raw)  # foreign stream with cross-block history
    assert z.last_inflate_tier() == 2
    z.inflate(raw, z.ZES_F_NO_FASTPATH)
    assert z.last_inflate_tier() == 3
    z.inflate(bytes.fromhex(golden("kat.json")["kat"]["FIXED"]))
    assert z.last_inflate_tier() in (3, 4)


def _foreign_cases(z):
    """Streams of another encoder (CPython's zlib module): history across blocks, stored and fixed blocks,
    sync-flush markers, blocks far longer than 131072 bytes.  -> (name, stream, plain, expected tier or None)"""
    import zlib as pz

    text = z.gen("itext", 61, 6 << 20).tobytes()
    rnd = z.gen("xorshift", 62, 3 << 20).tobytes()
    low = z.gen("lowent4k", 63, 24 << 20).tobytes()
    cases = [("text level 6", pz.compress(text, 6), text, 2), ("text level 1", pz.compress(text, 1), text, 2),
             ("text level 9", pz.compress(text, 9), text, 2),
             ("stored blocks", pz.compress(rnd, 6), rnd, 2),  # nothing to decode: header walk + parallel copy
             ("stored blocks, level 0", pz.compress(text[:1 << 20], 0), text[:1 << 20], 2),
             ("long blocks", pz.compress(low, 6), low, None)]
    co = pz.compressobj(6)
    parts, plain = [], []
    for i in range(40):  # dynamic / stored / empty stored (sync flush) / history reset (full flush), interleaved
        chunk = (text[i * 150000:(i + 1) * 150000], rnd[i * 70001:(i + 1) * 70001], text[i * 333:i * 333 + 1500])[i % 3]
        plain.append(chunk)
        parts.append(co.compress(chunk))
        parts.append(co.flush(pz.Z_FULL_FLUSH if i % 7 == 3 else pz.Z_SYNC_FLUSH if i % 2 else pz.Z_NO_FLUSH))
    parts.append(co.flush())
    cases.append(("mixed block types", b"".join(parts), b"".join(plain), 2))
    co = pz.compressobj(6, pz.DEFLATED, 15, 8, pz.Z_FIXED)
    fixed = co.compress(text[:1 << 20]) + co.flush()
    cases.append(("fixed blocks only", fixed, text[:1 << 20], None))
    return cases


def test_inflate_foreign_streams_in_parallel_segments(z, gpu):
    import torch

    for name, comp, plain, tier in _foreign_cases(z):
        out = z.inflate(comp)
        assert bytes(out) == plain, name
        if tier is not None:
            assert z.last_inflate_tier() == tier, (name, z.last_inflate_tier())
        # the device entry point, at an exact-size and at a too-small capacity
        d = dev(np.frombuffer(comp, dtype=np.uint8), gpu)
        o = torch.empty(len(plain), dtype=torch.uint8, device=gpu)
        back = z.inflate_tensor(d, o)
        assert back.numel() == len(plain) and bytes(back.cpu().numpy()) == plain, name
        with pytest.raises(z.ZlibEsError, match="need %d bytes" % len(plain)):
            z.inflate_tensor(d, torch.empty(len(plain) - 1, dtype=torch.uint8, device=gpu))
        assert bytes(z.inflate(comp, z.ZES_F_NO_FASTPATH)) == plain, name


def test_inflate_foreign_stream_sweep(z, gpu):
    """Encoder settings that change the block structure: strategies, window and memory levels (memLevel 1 =
    blocks of a few hundred symbols: thousands of segments), dictionary-less restarts.  Checked against the plain
    input; every stream must come out of the segment-parallel tier."""
    import zlib as pz

    rng = np.random.default_rng(2024)
    text = z.gen("itext", 81, 3 << 20)
    low = z.gen("lowent4k", 82, 3 << 20)
    skew = (rng.integers(0, 256, 3 << 20, dtype=np.uint8) & rng.integers(0, 256, 3 << 20, dtype=np.uint8)
            & rng.integers(0, 256, 3 << 20, dtype=np.uint8))  # compressible only by its symbol statistics
    mixed = np.concatenate([text[:700000], skew[:500000], low[:300000], text[700000:1400000]])
    cases = []
    for name, data in (("text", text), ("pattern", low), ("skewed bytes", skew), ("mixed", mixed)):
        for level, wbits, mem, strat in ((6, 15, 8, pz.Z_DEFAULT_STRATEGY), (1, 15, 9, pz.Z_DEFAULT_STRATEGY),
                                         (9, 15, 1, pz.Z_DEFAULT_STRATEGY), (6, 9, 8, pz.Z_DEFAULT_STRATEGY),
                                         (6, 12, 4, pz.Z_FILTERED), (6, 15, 8, pz.Z_HUFFMAN_ONLY), (6, 15, 2, pz.Z_RLE)):
            co = pz.compressobj(level, pz.DEFLATED, wbits, mem, strat)
            cases.append(("%s level %d wbits %d mem %d strategy %d" % (name, level, wbits, mem, strat), co.compress(data.tobytes()) + co.flush(), data))
    slow = []
    for name, comp, data in cases:
        out = z.inflate(comp)  # (wbits 9 makes the first byte 0x18: still CM 8, accepted like any other, src/zlib.ts:13-16)
        assert out.tobytes() == data.tobytes(), name
        # short streams, and streams of stored blocks only (nothing to cut at), go to the serial wavefront by design
        if len(comp) >= 32768 and len(comp) < 0.98 * data.size and z.last_inflate_tier() != 2:
            slow.append(name)
    assert not slow, slow


def test_segment_tier_block_decoder_and_wave_decoder_agree(z, gpu):
    """The segment-parallel tier sends every work item to the block decoder first (k_inf_seg_block_par) and the items
    it declines — stored and fixed blocks, blocks behind a start that is not on the list, blocks of more than
    128 KiB — to the wave decoder; ZES_NO_SEG_PAR=1 sends everything to the wave decoder.  Same bytes either way,
    and the input back."""
    import zlib as pz

    text = z.gen("itext", 91, 5 << 20)
    low = z.gen("lowent4k", 92, 6 << 20)
    rnd = z.gen("xorshift", 93, 1 << 20)
    mixed = np.concatenate([text[:1500000], low[:2000000], rnd[:300000], text[1500000:3000000], low[2000000:2300000]])
    cases = []
    for name, data in (("text", text), ("pattern", low), ("mixed", mixed)):
        for level, mem, strat, flush in ((6, 8, pz.Z_DEFAULT_STRATEGY, None), (9, 9, pz.Z_DEFAULT_STRATEGY, pz.Z_SYNC_FLUSH),
                                         (1, 8, pz.Z_DEFAULT_STRATEGY, pz.Z_FULL_FLUSH), (6, 8, pz.Z_FIXED, None),
                                         (4, 3, pz.Z_DEFAULT_STRATEGY, pz.Z_SYNC_FLUSH)):
            co = pz.compressobj(level, pz.DEFLATED, 15, mem, strat)
            parts, b, step = [], data.tobytes(), len(data) // 5 + 1
            for o in range(0, len(b), step):
                parts.append(co.compress(b[o:o + step]))
                if flush is not None:
                    parts.append(co.flush(flush))
            parts.append(co.flush())
            cases.append(("%s level %d mem %d strategy %d flush %s" % (name, level, mem, strat, flush), np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), data))
    for name, comp, data in cases:
        a = z.inflate(comp)
        ta = z.last_inflate_tier()
        os.environ["ZES_NO_SEG_PAR"] = "1"
        try:
            b = z.inflate(comp)
        finally:
            del os.environ["ZES_NO_SEG_PAR"]
        assert a.tobytes() == data.tobytes() and b.tobytes() == data.tobytes(), name
        assert ta == z.last_inflate_tier(), (name, ta, z.last_inflate_tier())
        # damaged in the middle: the same result (bytes or the reference's error) both ways
        bad = comp.copy()
        bad[len(bad) // 2] ^= 0x10

        def run():
            try:
                return ("out", z.inflate(bad).tobytes())
            except z.ZlibEsError as e:
                return ("err", str(e))

        ra = run()
        os.environ["ZES_NO_SEG_PAR"] = "1"
        try:
            rb = run()
        finally:
            del os.environ["ZES_NO_SEG_PAR"]
        assert ra == rb, name


def test_inflate_reports_needed_size(z, oracle, gpu):
    import torch

    a = z.gen("itext", 31, 500000)
    comp = dev(oracle.deflate(a), gpu)
    out = torch.empty(1024, dtype=torch.uint8, device=gpu)
    n = C.c_uint64()
    for flags in (0, z.ZES_F_NO_FASTPATH):
        rc = z.lib().zes_inflate_dev(comp.data_ptr(), comp.numel(), out.data_ptr(), out.numel(), C.byref(n), flags)
        assert rc == z.ZES_E_NOSPACE and n.value == 500000
    assert len(z.inflate(comp.cpu().numpy())) == 500000  # host API sizes, allocates, decodes


def test_inflate_false_positive_block_headers_are_survivable(z, oracle, gpu):
    """A literal-only payload that *contains* a valid dynamic block header: the candidate finder
    reports a start that is not on the chain; the chain walk must still produce the right bytes."""
    import torch

    inner = oracle.deflate(z.gen("itext", 40, 3000))[2:-4]  # raw deflate bytes: begin with a clean header
    a = np.concatenate([z.gen("xorshift", 41, 70000), inner, z.gen("xorshift", 42, 200000), inner, z.gen("xorshift", 43, 5000)])
    comp = dev(oracle.deflate(a), gpu)
    out = torch.empty(len(a), dtype=torch.uint8, device=gpu)
    back = z.inflate_tensor(comp, out)
    assert back.numel() == len(a) and (back.cpu().numpy() == a).all()


@pytest.mark.parametrize("repeats,length", [(40, 3), (300, 4), (700, 3), (1500, 5), (6000, 3), (3000, 40)])
def test_deflate_match_poor_blocks_with_listed_matches(z, oracle, gpu, repeats, length):
    """Incompressible data with a number of short repeats copied in: blocks whose matches k_lz_match only lists
    (no more sorted slots than the list holds: the result words are not cleared), blocks with a full list over
    cleared words, and blocks whose list overflows (the parse searches the result words) — bit-exact either way."""
    n = 3 * 131072 + 777
    a = z.gen("xorshift", 4000 + repeats, n).copy()
    rng = np.random.default_rng(repeats * 7 + length)
    for _ in range(repeats):
        src = int(rng.integers(0, n - 2 * length - 40000))
        dst = src + length + int(rng.integers(0, 30000))
        a[dst:dst + length] = a[src:src + length]
    exp = oracle.deflate(a)
    got = z.deflate(a)
    assert got.tobytes() == exp.tobytes()
    assert z.inflate(got).tobytes() == a.tobytes()


def test_inflate_damaged_block_that_runs_past_both_end_estimates(z, oracle, gpu):
    """tools/gpu_fuzz.py seed 910, case 1709: another encoder's stream with one bit flipped inside a block.  The block
    misses its end-of-block code and decodes on through the next blocks' bits to a BTYPE 3 header: the reference throws
    'Not supported BTYPE'.  The block decoder's segments (64 bits at least) reached beyond the bytes staged for the
    two end estimates, its chain walked stale bytes of the staging area, and the serial tail behind the last segment met
    a real block end: 134226 bytes of garbage accepted.  Now a chain that leaves the staged bytes fails."""
    data = np.fromfile(os.path.join(GOLDEN, "fuzz", "damaged_zlib_stream_seed910_case1709.bin"), dtype=np.uint8)
    with pytest.raises(oracle.OracleError) as ex:
        oracle.inflate(data)
    with pytest.raises(z.ZlibEsError) as got:
        z.inflate(data)
    assert got.value.code == ex.value.code
    with pytest.raises(z.ZlibEsError) as got:
        z.inflate(data, z.ZES_F_NO_FASTPATH)
    assert got.value.code == ex.value.code


def test_inflate_periodic_stream_with_more_scan_survivors_than_the_list_holds(z, oracle, gpu):
    """The compressed form of periodic data is periodic itself: the bits of one repeated match pass the block-start
    scan at every repetition, thousands of times in an 8 KiB chunk (found by tools/gpu_fuzz.py, seed 77: the scan used
    to give such a chunk up with its reserved survivor slots unwritten, and the header test then read through stale
    entries: a GPU memory fault).  tools/gpu_scan_overflow_probe.py lists the inputs: zlib -6 of a 259-byte period has
    1549 survivors in 2099 bytes (n = 400000) and 11626 in 13435 (n = 3000000)."""
    import zlib as pz

    for n in (400000, 3000000):
        a = np.resize(z.gen("xorshift", 266, n)[:259], n).copy()
        fz = np.frombuffer(pz.compress(a.tobytes(), 6), dtype=np.uint8)
        assert z.inflate(fz).tobytes() == a.tobytes()
        own = z.deflate(a)
        assert z.inflate(own).tobytes() == a.tobytes()
        for src in (fz, own):  # cut short: the same answer as the oracle, whichever tier ends up with it
            cut = src[: len(src) * 2 // 3]
            try:
                exp = ("out", oracle.inflate(cut).tobytes())
            except oracle.OracleError as ex:
                exp = ("err", ex.code)
            try:
                got = ("out", z.inflate(cut).tobytes())
            except z.ZlibEsError as ex:
                got = ("err", ex.code)
            assert got == exp


@pytest.mark.parametrize("kind,seed", [("itext", 1093), ("xorshift", 1025)])
def test_inflate_false_candidate_inside_a_block_stays_block_parallel(z, oracle, gpu, kind, seed):
    """These two 1 MiB inputs compress to streams whose *bits* contain a spurious, fully valid dynamic
    block header in the middle of a block (found on the GPU box: 9 candidates for 8 blocks).  The
    decoder of the enclosing block must extend its end estimate past the false candidate, the chain
    walk must drop it, and the blocks behind it must end up in their own slots — all in T1."""
    import torch

    a = z.gen(kind, seed, 1 << 20)
    comp = dev(oracle.deflate(a), gpu)
    out = torch.empty(len(a), dtype=torch.uint8, device=gpu)
    z.set_profiling(True)
    try:
        # the block-start search normally also applies the reference's run-length-coding rules, which these two
        # spurious headers break: without them (testing flag) they reach the block decoder
        back = z.inflate_tensor(comp, out, z.ZES_F_LOOSE_CANDIDATES)
        launches = {k: n for k, _, n in z.last_kernel_times()}
        tier = z.last_inflate_tier()
        back2 = z.inflate_tensor(comp, torch.empty_like(out))
        launches2 = {k: n for k, _, n in z.last_kernel_times()}
    finally:
        z.set_profiling(False)
    assert back.numel() == len(a) and (back.cpu().numpy() == a).all()
    assert tier == 1
    # the false candidate was really there: blocks behind it are moved into place (k_inf_move_slots) and the one
    # whose slot the capacity cut off is decoded again (second k_inf_block_par launch)
    assert launches.get("k_inf_block_par", 0) + launches.get("k_inf_block_par2", 0) == 2  # (par2: the variant for compressible data)
    assert back2.numel() == len(a) and (back2.cpu().numpy() == a).all()
    assert z.last_inflate_tier() == 1 and launches2.get("k_inf_block_par", 0) + launches2.get("k_inf_block_par2", 0) == 1 and "k_inf_move_slots" not in launches2


def test_inflate_mostly_8bit_codes_with_other_tokens_mixed_in(z, oracle, gpu):
    """The block decoder's fast paths for 8-bit literal codes must hand over to the
    generic construction when a segment holds more other tokens than its list takes, and must agree
    with it when matches and shorter codes are sprinkled in."""
    import torch

    rnd = z.gen("xorshift", 81, 600000).copy()
    a = rnd.copy()
    a[::7] = 0                                   # one byte value gets a short code: many non-8-bit tokens per segment
    b = rnd.copy()
    for k in range(40, len(b) - 200, 997):       # copies of earlier data: matches between long literal runs
        b[k:k + 37] = b[k - 31:k + 6]
    c = rnd.copy()
    c[300000:] = z.gen("itext", 82, 300000)      # a block of each kind, and one that changes in the middle
    for data in (a, b, c):
        comp = dev(oracle.deflate(data), gpu)
        out = torch.empty(len(data), dtype=torch.uint8, device=gpu)
        back = z.inflate_tensor(comp, out)
        assert back.numel() == len(data) and (back.cpu().numpy() == data).all()
        assert z.last_inflate_tier() == 1


# ---------------------------------------------------------------------------------------------
# BASELINE.json sizes: 64 MiB, pinned by sha256 of the reference's own output + round trip
# ---------------------------------------------------------------------------------------------
def test_64mib_false_candidate_moves_hundreds_of_blocks(z, gpu):
    """xorshift seed 12347 at 64 MiB: a spurious header (visible only with the loose search) sits in the stream, so
    every block behind it is decoded one slot too far right and has to be moved; same bytes either way."""
    import torch

    n = 64 << 20
    t = dev(z.gen("xorshift", 12347, n), gpu)
    comp = z.deflate_tensor(t).clone()
    out = torch.empty(n, dtype=torch.uint8, device=gpu)
    z.set_profiling(True)
    try:
        back = z.inflate_tensor(comp, out, z.ZES_F_LOOSE_CANDIDATES)
        launches = {k: c for k, _, c in z.last_kernel_times()}
    finally:
        z.set_profiling(False)
    assert back.numel() == n and bool((back == t).all()) and z.last_inflate_tier() == 1
    assert launches.get("k_inf_move_slots") == 1 and launches.get("k_inf_block_par", 0) + launches.get("k_inf_block_par2", 0) == 2
    back = z.inflate_tensor(comp, out)
    assert back.numel() == n and bool((back == t).all()) and z.last_inflate_tier() == 1


@pytest.mark.parametrize("kind", ["xorshift", "itext", "lowent4k"])
def test_64mib_bit_exact_and_round_trip(z, gpu, kind):
    import torch

    e = [x for x in golden("manifest.json")["big"] if x["kind"] == kind][0]
    a = z.gen(kind, e["seed"], e["n"])
    assert hashlib.sha256(a.tobytes()).hexdigest() == e["input_sha256"]
    t = dev(a, gpu)
    comp = z.deflate_tensor(t)
    assert comp.numel() == e["deflate_len"]
    assert hashlib.sha256(comp.cpu().numpy().tobytes()).hexdigest() == e["deflate_sha256"]
    out = torch.empty(e["n"], dtype=torch.uint8, device=gpu)
    back = z.inflate_tensor(comp.clone(), out)
    assert back.numel() == e["n"] and bool((back == t).all())
    assert z.last_inflate_tier() == 1
    # size-independent property: Adler-32 trailer written by the device == Adler-32 of the input
    tail = comp[-4:].cpu().numpy()
    assert int.from_bytes(tail.tobytes(), "big") == z.adler32_tensor(t)
