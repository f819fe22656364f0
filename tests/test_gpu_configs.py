"""BASELINE.json's larger configurations and the entry points around them, on the GPU, through the C-ABI:

  configs[3]  this GPU's share of the 1024 x 1 MiB batch (128 buffers, generator i % 3, seed 12345 + i) through the
              batch entry points, every buffer against the reference's own output (tests/golden/batch1m.json)
  configs[4]  one 256 MiB low-entropy buffer against the reference's own output, round trip, block-parallel tier
  multi-GPU   the gather of shard.py on the "nccl" backend (world size 1: all a one-GPU box allows), one buffer
              split into block ranges and joined again
  boundary    host-pointer batch forms, calls from other threads, pinned host memory, unaligned device views
"""
import ctypes as C
import hashlib
import os
import threading

import numpy as np
import pytest

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu
MIX = ("xorshift", "itext", "lowent4k")


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_shard():
    import importlib.util

    spec = importlib.util.spec_from_file_location("zlibes_amd_shard", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    return shard


@pytest.fixture(scope="module")
def nccl_group(gpu):
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29533"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    yield dist.group.WORLD
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# configs[3]: this GPU's 128 of the 1024 x 1 MiB buffers
# ---------------------------------------------------------------------------------------------
def test_config3_batch_share_bit_exact_vs_reference(z, gpu):
    import torch

    entries = golden("batch1m.json")[:128]  # rank 0's share of the job (bench.py: buffer i of rank r = r * 128 + i)
    n1 = 1 << 20
    cnt = len(entries)
    host = np.empty(cnt * n1, dtype=np.uint8)
    for k, e in enumerate(entries):
        assert e["kind"] == MIX[e["i"] % 3] and e["seed"] == 12345 + e["i"] and e["n"] == n1
        host[k * n1:(k + 1) * n1] = z.gen(e["kind"], e["seed"], n1)
        if k % 16 == 0:
            assert sha(host[k * n1:(k + 1) * n1]) == e["input_sha256"]
    d_in = dev(host, gpu)
    bound = (z.deflate_bound(n1) + 15) // 16 * 16
    d_comp = torch.empty(bound * cnt, dtype=torch.uint8, device=gpu)
    in_off = [k * n1 for k in range(cnt)]
    c_off = [k * bound for k in range(cnt)]
    clen, st = z.deflate_batch_tensor(d_in, in_off, [n1] * cnt, d_comp, c_off, [bound] * cnt)
    assert not any(st)
    comp_h = d_comp.cpu().numpy()
    for k, e in enumerate(entries):
        assert clen[k] == e["deflate_len"], (k, clen[k], e["deflate_len"])
        assert sha(comp_h[c_off[k]:c_off[k] + clen[k]]) == e["deflate_sha256"], k
    d_back = torch.empty(cnt * n1, dtype=torch.uint8, device=gpu)
    olen, st = z.inflate_batch_tensor(d_comp, c_off, clen, d_back, in_off, [n1] * cnt)
    assert not any(st) and all(o == n1 for o in olen)
    assert bool((d_back == d_in).all()) and z.last_inflate_tier() == 1


# ---------------------------------------------------------------------------------------------
# configs[4]: one 256 MiB low-entropy buffer
# ---------------------------------------------------------------------------------------------
def test_config4_256mib_lowent_bit_exact_vs_reference(z, gpu):
    import torch

    e = [x for x in golden("manifest.json")["big"] if x["kind"] == "lowent4k" and x["n"] == 256 << 20][0]
    a = z.gen("lowent4k", e["seed"], e["n"])
    assert sha(a) == e["input_sha256"]
    t = dev(a, gpu)
    comp = z.deflate_tensor(t)
    assert comp.numel() == e["deflate_len"]
    assert sha(comp.cpu().numpy()) == e["deflate_sha256"]
    out = torch.empty(e["n"], dtype=torch.uint8, device=gpu)
    back = z.inflate_tensor(comp.clone(), out)
    assert back.numel() == e["n"] and bool((back == t).all())
    assert z.last_inflate_tier() == 1
    assert int.from_bytes(comp[-4:].cpu().numpy().tobytes(), "big") == z.adler32_tensor(t)


# ---------------------------------------------------------------------------------------------
# multi-GPU code paths on the one GPU there is
# ---------------------------------------------------------------------------------------------
def test_run_sharded_over_nccl(z, oracle, gpu, nccl_group):
    shard = load_shard()
    specs = [("itext", 1, 50000), ("xorshift", 2, 131074), ("lowent4k", 3, 9000), ("itext", 4, 1), ("xorshift", 5, 300), ("itext", 6, 262144)]
    bufs = [z.gen(k, s, n) for k, s, n in specs]
    run_deflate, run_inflate = shard.gpu_engines(z, gpu)
    res = shard.run_sharded(bufs, run_deflate, group=nccl_group, device=gpu)
    comps = []
    for b, (st, data) in zip(bufs, res):
        if len(b) == 1:
            assert st == -3 and len(data) == 0  # 'Data is corrupted' (SURVEY A.7), carried through the gather
            continue
        assert st == 0 and data.tobytes() == oracle.deflate(b).tobytes()
        comps.append((b, data))
    back = shard.run_sharded([c for _, c in comps], run_inflate, group=nccl_group, device=gpu)
    for (b, _), (st, data) in zip(comps, back):
        assert st == 0 and data.tobytes() == b.tobytes()


def test_block_ranges_join_to_the_whole_stream(z, oracle, gpu):
    """zes_deflate_range_dev x 3 + zes_deflate_join_dev == zes_deflate_dev of the whole buffer == the oracle."""
    for kind, seed, n, cuts in (("itext", 31, 5 * 131072 + 4321, (0, 2, 3, 6)), ("lowent4k", 32, 3 * 131072, (0, 1, 2, 3)),
                                ("xorshift", 33, 2 * 131072 + 2, (0, 1, 1, 3))):
        a = z.gen(kind, seed, n)
        t = dev(a, gpu)
        want = z.deflate_tensor(t).cpu().numpy()
        assert want.tobytes() == oracle.deflate(a).tobytes()
        pieces, bits, adlers, lens = [], [], [], []
        for k in range(len(cuts) - 1):
            lo, hi = cuts[k] * 131072, min(n, cuts[k + 1] * 131072)
            if hi <= lo:
                continue
            p, nb, ad = z.deflate_range_tensor(t, lo, hi, final=(hi == n))
            want_p, want_bits = oracle.deflate_range(a, lo, hi - lo, hi == n)
            assert nb == want_bits and p.cpu().numpy().tobytes() == want_p.tobytes(), (kind, k)
            assert ad == oracle.adler32(a[lo:hi])
            pieces.append(p.clone())
            bits.append(nb)
            adlers.append(ad)
            lens.append(hi - lo)
        got = z.deflate_join_tensors(pieces, bits, adlers, lens).cpu().numpy()
        assert got.tobytes() == want.tobytes(), kind


def test_deflate_split_over_nccl(z, oracle, gpu, nccl_group):
    shard = load_shard()
    n = 3 * 131072 + 777
    a = z.gen("itext", 55, n)
    t = dev(a, gpu)
    res = shard.deflate_split(n, lambda lo, hi, final: z.deflate_range_tensor(t, lo, hi, final), join=z.deflate_join_tensors,
                              group=nccl_group, device=gpu)
    assert res.cpu().numpy().tobytes() == oracle.deflate(a).tobytes()


def test_stream_ranges_decode_to_the_whole(z, oracle, gpu):
    """zes_inflate_range_dev over 2..4 bit ranges of one stream: the block starts it reports are the oracle's map of
    the stream, the ranges fit (shard.check_chain) and the outputs concatenate to the input."""
    import torch

    shard = load_shard()
    for kind, seed, n, world in (("itext", 61, 9 * 131072 + 999, 4), ("xorshift", 62, 6 * 131072, 3), ("lowent4k", 63, 20 * 131072 + 2, 4),
                                 ("itext", 64, 131072 * 2, 2), ("itext", 65, 50000, 2)):
        a = z.gen(kind, seed, n)
        comp = oracle.deflate(a)
        starts, ends = oracle.inflate_blocks(comp)
        t = dev(comp, gpu)
        table, outs = [], []
        for r, (lo, own) in enumerate(shard.split_bits(len(comp), world)):
            out = torch.zeros(n + 131072, dtype=torch.uint8, device=gpu)
            res = z.inflate_range_tensor(t, lo, own, r == 0, out)
            assert res is not None, (kind, r)
            ln, fb, eb, nb, fin = res
            ks = [k for k, s in enumerate(starts) if max(16, lo) <= s < own]
            assert nb == len(ks), (kind, r, nb, ks)
            if ks:
                assert fb == starts[ks[0]] and fin == (ks[-1] == len(starts) - 1)
                if not fin:
                    assert eb == starts[ks[-1] + 1]
                assert ln == ends[ks[-1]] - (ends[ks[0] - 1] if ks[0] else 0)
            table.append([1, fb, eb, nb, ln, int(fin)])
            outs.append(out[:ln])
        assert shard.check_chain(table), (kind, table)
        assert torch.cat(outs).cpu().numpy().tobytes() == a.tobytes(), kind
    # a range of another encoder's stream is refused, not mis-decoded
    import zlib as pyzlib

    other = np.frombuffer(pyzlib.compress(z.gen("itext", 66, 600000).tobytes(), 6), dtype=np.uint8).copy()
    t = dev(other, gpu)
    out = torch.zeros(700000, dtype=torch.uint8, device=gpu)
    assert z.inflate_range_tensor(t, 16, 8 * len(other), True, out) is None


def test_inflate_split_over_nccl(z, oracle, gpu, nccl_group):
    import torch

    shard = load_shard()
    n = 7 * 131072 + 31
    a = z.gen("itext", 71, n)
    t = dev(oracle.deflate(a), gpu)
    out = torch.zeros(n + 131072, dtype=torch.uint8, device=gpu)

    def rng(lo, own, exact):
        res = z.inflate_range_tensor(t, lo, own, exact, out)
        return None if res is None else (out,) + res

    def whole():
        return z.inflate_tensor(t, torch.empty(n, dtype=torch.uint8, device=gpu))

    res = shard.inflate_split(t.numel(), rng, whole, group=nccl_group, device=gpu)
    assert res.cpu().numpy().tobytes() == a.tobytes()
    # and the fallback leg: a stream that is not reference-made
    import zlib as pyzlib

    t2 = dev(np.frombuffer(pyzlib.compress(a.tobytes(), 6), dtype=np.uint8).copy(), gpu)
    res = shard.inflate_split(t2.numel(), lambda lo, own, ex: (lambda r: None if r is None else (out,) + r)(z.inflate_range_tensor(t2, lo, own, ex, out)),
                              lambda: z.inflate_tensor(t2, torch.empty(n, dtype=torch.uint8, device=gpu)), group=nccl_group, device=gpu)
    assert res.cpu().numpy().tobytes() == a.tobytes()


def test_long_streams_piece_by_piece(z, oracle, gpu):
    """Streams of 512 MiB and more are decoded by the block-parallel tier in pieces (256 MiB); ZES_F_PIECES runs the
    same loop with 1 MiB pieces on streams of ordinary length: same bytes, same tier, same errors."""
    import torch

    for kind, seed, n in (("xorshift", 81, 5 * (1 << 20) + 12345), ("itext", 82, 6 << 20), ("lowent4k", 83, (4 << 20) + 2), ("itext", 84, 3000)):
        a = z.gen(kind, seed, n)
        comp = oracle.deflate(a)
        t = dev(comp, gpu)
        out = torch.zeros(n, dtype=torch.uint8, device=gpu)
        got = z.inflate_tensor(t, out, flags=z.ZES_F_PIECES)
        assert z.last_inflate_tier() == 1, kind
        assert got.cpu().numpy().tobytes() == a.tobytes(), kind
        # capacity one byte short: the size needed comes back
        small = torch.zeros(n - 1 - ((n - 1) % 16), dtype=torch.uint8, device=gpu)
        with pytest.raises(z.ZlibEsError) as e:
            z.inflate_tensor(t, small, flags=z.ZES_F_PIECES)
        assert e.value.code == -16 and e.value.need == n
    # another encoder's stream falls through to the general tiers with the same bytes
    import zlib as pyzlib

    a = z.gen("itext", 85, 3 << 20)
    t = dev(np.frombuffer(pyzlib.compress(a.tobytes(), 6), dtype=np.uint8).copy(), gpu)
    got = z.inflate_tensor(t, torch.zeros(len(a), dtype=torch.uint8, device=gpu), flags=z.ZES_F_PIECES)
    assert z.last_inflate_tier() != 1 and got.cpu().numpy().tobytes() == a.tobytes()
    # a corrupted block in the middle: the reference's error, whatever tier finds it
    a = z.gen("itext", 86, 4 << 20)
    comp = oracle.deflate(a).copy()
    comp[len(comp) // 2: len(comp) // 2 + 64] ^= 0x5A
    try:
        want = ("ok", oracle.inflate(comp).tobytes())
    except oracle.OracleError as e:
        want = ("err", str(e))
    try:
        got = ("ok", z.inflate_tensor(dev(comp, gpu), torch.zeros(len(a) + (1 << 20), dtype=torch.uint8, device=gpu), flags=z.ZES_F_PIECES).cpu().numpy().tobytes())
    except z.ZlibEsError as e:
        got = ("err", str(e))
    assert got == want


def test_stream_of_more_than_512_mib(z, gpu):
    """A compressed stream of 2^29 bytes and more (bit positions beyond 32 bits) stays on the block-parallel tier:
    three pieces here.  The stream is made by the engine's own deflate (bit-exact with the reference by the tests
    above); the round trip must give the input back."""
    import torch

    n = 650 << 20
    a = z.gen("xorshift", 91, n)
    t = torch.from_numpy(a).to(gpu)
    comp = z.deflate_tensor(t)
    assert comp.numel() >= 1 << 29
    comp = comp.clone()
    back = torch.empty(n, dtype=torch.uint8, device=gpu)
    got = z.inflate_tensor(comp, back)
    assert z.last_inflate_tier() == 1
    assert got.numel() == n and torch.equal(got, t)


# ---------------------------------------------------------------------------------------------
# boundary
# ---------------------------------------------------------------------------------------------
def test_host_batch_forms(z, oracle, gpu):
    specs = [("itext", 1, 300000), ("xorshift", 2, 131074), ("lowent4k", 3, 70000), ("itext", 4, 2), ("xorshift", 5, 131073), ("itext", 6, 1)]
    bufs = [z.gen(k, s, n) for k, s, n in specs]
    outs = z.deflate_batch(bufs)
    comps = []
    for b, o in zip(bufs, outs):
        if len(b) % 131072 == 1:
            assert isinstance(o, z.ZlibEsError) and str(o) == "Data is corrupted"
            continue
        assert o.tobytes() == oracle.deflate(b).tobytes()
        comps.append((b, o))
    streams = [c for _, c in comps] + [np.frombuffer(bytes([0x78, 0x9C, 7, 0, 0, 0]), dtype=np.uint8), np.zeros(0, dtype=np.uint8),
                                       oracle.deflate(np.zeros(3 << 20, dtype=np.uint8))]  # (the last one inflates far beyond the first guess)
    back = z.inflate_batch(streams)
    for (b, _), o in zip(comps, back):
        assert o.tobytes() == b.tobytes()
    assert isinstance(back[len(comps)], z.ZlibEsError) and str(back[len(comps)]) == "Not supported BTYPE : 3"
    assert isinstance(back[len(comps) + 1], z.ZlibEsError) and str(back[len(comps) + 1]) == "Not compressed by deflate"
    assert len(back[-1]) == 3 << 20 and not back[-1].any()


def test_calls_from_other_threads(z, oracle, gpu):
    """Every entry point makes the bound device current on its calling thread (HIP's current device is per thread)
    and holds one lock: host calls from four threads at once give the single-thread results."""
    bufs = [z.gen(MIX[i % 3], 700 + i, 200000 + 1111 * i) for i in range(8)]
    want = [oracle.deflate(b) for b in bufs]
    got = [None] * len(bufs)
    errs = []

    def work(i):
        try:
            c = z.deflate(bufs[i])
            r = z.inflate(c)
            a = z.adler32(bufs[i])
            got[i] = (c.tobytes(), r.tobytes(), a)
        except Exception as e:  # noqa: BLE001 - reported below
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(bufs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i, b in enumerate(bufs):
        assert got[i] == (want[i].tobytes(), b.tobytes(), oracle.adler32(b))


def test_pinned_host_memory_and_large_host_calls(z, oracle, gpu):
    """The host forms stage pageable memory through a pinned ring (chunks of 4 MiB) and hand pinned memory to the
    DMA engine as it is: both give the bytes of the device form."""
    n = (9 << 20) + 12345  # three ring chunks, the last one short
    a = z.gen("itext", 808, n)
    want = oracle.deflate(a)
    got = z.deflate(a)
    assert got.tobytes() == want.tobytes()
    assert z.inflate(got).tobytes() == a.tobytes()
    pin = z.host_alloc(n)
    try:
        pin[:] = a
        got2 = z.deflate(pin)
        assert got2.tobytes() == want.tobytes()
        assert z.adler32(pin) == oracle.adler32(a)
    finally:
        z.host_free(pin)


def test_pipelined_host_calls(z, oracle, gpu):
    """Host calls on large buffers run in pieces (upload, kernels and download side by side): the same bytes as the
    one-pass path (ZES_NO_PIPELINE=1), which the tests above hold against the oracle and the reference's goldens."""
    import zlib as pyzlib

    L = z.lib()
    for kind, n, pinned in (("xorshift", (70 << 20) + 12345, False), ("itext", (96 << 20) + 131073 + 5, True), ("lowent4k", 64 << 20, False)):
        src = z.gen(kind, 4242, n)
        cap = z.deflate_bound(n)
        if pinned:
            a, comp, comp1, back = z.host_alloc(n), z.host_alloc(cap), z.host_alloc(cap), z.host_alloc(n)
            a[:] = src
        else:
            a, comp, comp1, back = src, np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        clen, clen1, blen = C.c_uint64(), C.c_uint64(), C.c_uint64()
        assert L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, cap, C.byref(clen)) == 0
        os.environ["ZES_NO_PIPELINE"] = "1"
        try:
            assert L.zes_deflate(a.ctypes.data, n, comp1.ctypes.data, cap, C.byref(clen1)) == 0
        finally:
            del os.environ["ZES_NO_PIPELINE"]
        assert clen.value == clen1.value and sha(comp[: clen.value]) == sha(comp1[: clen1.value]), kind
        assert pyzlib.adler32(src.tobytes()) == int.from_bytes(comp[clen.value - 4: clen.value].tobytes(), "big")
        # a capacity that holds the result but not the bound: still the same bytes; one byte short: the size needed
        tight = np.zeros(clen.value, dtype=np.uint8)
        t = C.c_uint64()
        assert L.zes_deflate(a.ctypes.data, n, tight.ctypes.data, tight.size, C.byref(t)) == 0 and sha(tight) == sha(comp[: clen.value])
        assert L.zes_deflate(a.ctypes.data, n, tight.ctypes.data, tight.size - 1, C.byref(t)) == -16 and t.value == clen.value
        # and back, in pieces
        assert L.zes_inflate(comp.ctypes.data, clen.value, back.ctypes.data, n, C.byref(blen), 0) == 0
        assert blen.value == n and z.last_inflate_tier() == 1 and sha(back) == sha(src), kind
        assert L.zes_inflate(comp.ctypes.data, clen.value, back.ctypes.data, n - 1, C.byref(blen), 0) == -16 and blen.value == n
        got = z.inflate(comp[: clen.value])  # (the form that allocates once the size is known)
        assert sha(got) == sha(src)
        # bytes behind the end of the stream are ignored (src/inflate.ts:22-37 stops at BFINAL)
        longer = np.concatenate([comp[: clen.value], np.frombuffer(os.urandom(3 << 20), dtype=np.uint8)])
        assert sha(z.inflate(longer)) == sha(src)
        if pinned:
            for x in (a, comp, comp1, back):
                z.host_free(x)
    # another encoder's long stream: the pieces do not chain, the one-pass path decodes it
    src = z.gen("xorshift", 77, 24 << 20)
    other = np.frombuffer(pyzlib.compress(src.tobytes(), 1), dtype=np.uint8).copy()
    assert len(other) >= 8 << 20
    assert sha(z.inflate(other)) == sha(src) and z.last_inflate_tier() != 1
    # ... and one that begins like the reference's (a dynamic block): its first piece fails on the history the block
    # decoder does not have, with the next piece enqueued already
    src = z.gen("itext", 78, 40 << 20)
    other = np.frombuffer(pyzlib.compress(src.tobytes(), 6), dtype=np.uint8).copy()
    assert len(other) >= 8 << 20
    assert sha(z.inflate(other)) == sha(src) and z.last_inflate_tier() == 2
    back = np.zeros(len(src), dtype=np.uint8)
    blen = C.c_uint64()
    assert L.zes_inflate(other.ctypes.data, len(other), back.ctypes.data, len(src), C.byref(blen), 0) == 0 and sha(back) == sha(src)


def test_unaligned_device_views_are_staged(z, oracle, gpu):
    import torch

    a = z.gen("itext", 4141, 300003)
    t = dev(a, gpu)
    v = t[3:]  # data_ptr() % 16 == 3
    assert v.data_ptr() % 16 != 0
    comp = z.deflate_tensor(v)
    assert comp.cpu().numpy().tobytes() == oracle.deflate(a[3:]).tobytes()
    big = torch.zeros(comp.numel() + 5, dtype=torch.uint8, device=gpu)
    big[5:] = comp
    out = torch.empty(300000, dtype=torch.uint8, device=gpu)
    back = z.inflate_tensor(big[5:], out)
    assert back.cpu().numpy().tobytes() == a[3:].tobytes()
    with pytest.raises(AssertionError):
        z.inflate_tensor(comp, torch.empty(300016, dtype=torch.uint8, device=gpu)[1:])


def test_inflate_size_keeps_nothing(z, oracle, gpu):
    a = z.gen("lowent4k", 9, 500000)
    c = oracle.deflate(a)
    need = C.c_uint64()
    assert z.lib().zes_inflate_size(c.ctypes.data, c.size, C.byref(need), 0) == 0 and need.value == a.size
    assert not hasattr(z.lib(), "zes_inflate_fetch")


def test_trim_returns_the_scratch_and_calls_go_on(z, oracle, gpu):
    import torch

    a = z.gen("itext", 77, 6 << 20)
    ref = oracle.deflate(a)
    assert np.array_equal(z.deflate(a), ref)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    z.trim()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 - free0 >= 40 << 20  # (the deflate scratch of a 6 MiB input alone is ~14 bytes per byte)
    # the pools grow again on demand: same results, both directions, device and host entry points
    assert np.array_equal(z.deflate(a), ref)
    assert np.array_equal(z.inflate(ref), a)
    t = dev(a, gpu)
    out = torch.empty(z.deflate_bound(a.size), dtype=torch.uint8, device=gpu)
    assert np.array_equal(z.deflate_tensor(t, out).cpu().numpy(), ref)


def test_stored_streams_find_their_blocks_in_parallel(z, gpu):
    """Another encoder's stored blocks (incompressible input): from 2 MiB on the headers are found by a parallel search
    (k_inf_stored_find / k_inf_stored_rank); a stream that turns to dynamic blocks half way is left to the other tiers."""
    import zlib as pz

    import torch

    rnd = z.gen("xorshift", 31, 5 << 20)
    txt = z.gen("itext", 32, 3 << 20)
    z.set_profiling(True)
    try:
        for level, data, want_rank in ((0, rnd, True), (6, rnd, True), (6, np.concatenate([rnd, txt]), False), (0, rnd[: 1 << 20], False)):
            comp = np.frombuffer(pz.compress(data.tobytes(), level), dtype=np.uint8)
            out = torch.empty(data.size, dtype=torch.uint8, device=gpu)
            got = z.inflate_tensor(dev(comp, gpu), out)
            assert got.numel() == data.size and np.array_equal(got.cpu().numpy(), data)
            assert z.last_inflate_tier() == 2
            names = [k for k, ms, n in z.last_kernel_times()]
            assert ("k_inf_stored_copy" in names) == (want_rank or data.size == 1 << 20)  # (the mixed stream is not decoded by the stored path)
            if want_rank:
                assert "k_inf_stored_rank" in names and "k_inf_stored_walk" not in names
    finally:
        z.set_profiling(False)


def test_allocator_asked_early_for_an_upper_estimate(z, gpu):
    """zes_inflate_alloc with ZES_F_ALLOC_BOUND (include/zes.h; what the N-API addon passes): the allocator is asked after the
    first piece for an upper estimate, the result is a prefix of what it returned — or, when the stream's first piece is far
    more compressible than the rest and the estimate falls short, it is asked a second time for the exact size and the last
    pointer holds the result.  Same bytes as the exact-size protocol either way; a foreign stream takes the one-pass path."""
    import zlib as pyzlib

    L = z.lib()
    n = 40 << 20
    mixed = np.concatenate([z.gen("xorshift", 6, 12 << 20), z.gen("lowent4k", 5, n - (12 << 20))])  # the estimate from piece 0 (random bytes) falls short
    for name, src in (("itext", z.gen("itext", 99, n)), ("xorshift", z.gen("xorshift", 98, n)), ("mixed", mixed)):
        comp = z.deflate(src)
        calls, bufs = [], []

        def alloc(_user, _index, need):
            calls.append(int(need))
            bufs.append(np.zeros(max(int(need), 1), dtype=np.uint8))
            return bufs[-1].ctypes.data

        cb = z.ALLOC_FN(alloc)
        blen = C.c_uint64()
        assert L.zes_inflate_alloc(comp.ctypes.data, comp.size, cb, None, C.byref(blen), z.ZES_F_ALLOC_BOUND) == 0
        assert blen.value == n and 1 <= len(calls) <= 2 and calls[-1] >= n, (name, calls)
        assert sha(bufs[-1][:n]) == sha(src), name
        if name == "mixed":
            assert len(calls) == 2 and calls[0] < n and calls[1] == n, calls  # short estimate, then the exact size
        else:
            assert len(calls) == 1 and n <= calls[0] <= n + n // 8 + (1 << 20), (name, calls)  # (5 % and two blocks on top of the first piece's ratio)
    # an allocator that declines the early request ("not now") is asked once, for the exact size
    src = z.gen("itext", 99, n)
    comp = z.deflate(src)
    calls, bufs = [], []

    def picky(_user, index, need):
        calls.append((int(index), int(need)))
        if index & z.ZES_ALLOC_EARLY:
            return None
        bufs.append(np.zeros(int(need), dtype=np.uint8))
        return bufs[-1].ctypes.data

    cb = z.ALLOC_FN(picky)
    blen = C.c_uint64()
    assert L.zes_inflate_alloc(comp.ctypes.data, comp.size, cb, None, C.byref(blen), z.ZES_F_ALLOC_BOUND) == 0
    assert blen.value == n and len(calls) == 2 and calls[0][0] & z.ZES_ALLOC_EARLY and calls[1] == (0, n) and sha(bufs[-1]) == sha(src), calls
    # another encoder's stream: the pieces do not chain, the call starts over on the one-pass path and asks once, exactly
    src = z.gen("itext", 7, 24 << 20)
    foreign = np.frombuffer(pyzlib.compress(src.tobytes(), 6), dtype=np.uint8)
    calls, bufs = [], []
    cb = z.ALLOC_FN(lambda u, i, need: (calls.append(int(need)), bufs.append(np.zeros(int(need), dtype=np.uint8)), bufs[-1].ctypes.data)[2])
    blen = C.c_uint64()
    assert L.zes_inflate_alloc(foreign.ctypes.data, foreign.size, cb, None, C.byref(blen), z.ZES_F_ALLOC_BOUND) == 0
    assert blen.value == src.size and calls[-1] == src.size and sha(bufs[-1][: src.size]) == sha(src)
