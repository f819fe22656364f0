"""N>1 path on CPU: world_size-2 gloo run of the sharding/gather layer (zlib.es_amd/shard.py).
The per-buffer engine is injected; here it is the CPU oracle (test infrastructure), so the test
checks partitioning, ordering, error propagation and the gather — not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import __graft_entry__ as ge
    import _oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = ge.load()
    import importlib.util

    spec = importlib.util.spec_from_file_location("zlibes_amd.shard", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)

    specs = [("itext", 1, 50000), ("xorshift", 2, 131074), ("lowent4k", 3, 9000), ("itext", 4, 1), ("xorshift", 5, 300),
             ("itext", 6, 262144), ("lowent4k", 7, 2)]
    bufs = [z.gen(k, s, n) for k, s, n in specs]

    def engine(b):
        try:
            return 0, _oracle.deflate(b)
        except _oracle.OracleError as e:
            return e.code, np.zeros(0, dtype=np.uint8)

    res = shard.run_sharded(bufs, engine)
    # the asynchronous form bench.py uses under --gpus N (the transfers of one step in flight during the next): same bytes
    import torch

    owned = shard.partition([len(b) for b in bufs], world)
    mine = owned[rank]
    outs = [engine(bufs[i]) for i in mine]
    local = torch.cat([torch.from_numpy(d) for _, d in outs]) if outs else torch.empty(0, dtype=torch.uint8)
    lens, stats = [len(d) for _, d in outs], [st for st, _ in outs]
    pend = shard.gather_results(local, mine, lens, stats, owned, len(bufs), async_op=True)
    got = pend.finish()

    def same(g):
        if rank == 0:
            for i, (st, data) in enumerate(res):
                assert g.status[i] == st and g.result(i).numpy().tobytes() == data.tobytes()
        else:
            assert g is None

    same(got)
    # the hinted form (no wait for the size exchange before the transfers are posted): the step before as the hint,
    # then hints that are too long (padded message), a little too short (second round) and far too short (the arena
    # segment overflows: every rank repeats the gather with exact lengths)
    true_len = pend.lengths
    assert true_len == [len(d) for _, d in res] if rank == 0 else True
    for hint in (true_len, [x + 777 for x in true_len], [max(x - 100, 0) for x in true_len], [x // 3 for x in true_len],
                 [0] * len(true_len)):
        ph = shard.gather_results(local, mine, lens, stats, owned, len(bufs), async_op=True, hint=hint)
        same(ph.finish())
        assert ph.lengths == true_len
    if rank == 0:
        ok = True
        for b, (st, data) in zip(bufs, res):
            try:
                want = (0, _oracle.deflate(b).tobytes())
            except _oracle.OracleError as e:
                want = (e.code, b"")
            ok = ok and (st, data.tobytes()) == want
        q.put(ok)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def _load_shard():
    import importlib.util

    spec = importlib.util.spec_from_file_location("shard_mod", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    return shard


def _split_worker(rank, world, port, q):
    """One buffer over two ranks: block ranges, the (bits, adler, length) exchange, the gather of the
    pieces and the shift-merge at a seam that is not on a byte boundary — against the oracle's deflate of the whole."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import __graft_entry__ as ge
    import _oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = ge.load()
    shard = _load_shard()
    ok = True
    seams = []
    for kind, seed, n in (("itext", 77, 4 * 131072 + 50001), ("lowent4k", 5, 2 * 131072), ("xorshift", 9, 131072 + 2), ("itext", 3, 70000)):
        data = z.gen(kind, seed, n)  # 5 blocks -> 3 + 2; 2 -> 1 + 1; 2 -> 1 + 1 (a 2-byte last block); 1 -> 1 + 0

        def rng(lo, hi, final):
            piece, bits = _oracle.deflate_range(data, lo, hi - lo, final)
            seams.append(bits & 7)
            return piece, bits, _oracle.adler32(data[lo:hi])

        res = shard.deflate_split(n, rng)
        if rank == 0:
            ok = ok and res.tobytes() == _oracle.deflate(data).tobytes()
        else:
            assert res is None
    if rank == 0:
        q.put(ok and any(seams))  # at least one piece ended inside a byte
    dist.barrier()
    dist.destroy_process_group()


def test_split_one_buffer_over_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


def _oracle_range(_oracle, comp, want):
    """What zes_inflate_range_dev does, restated with the oracle's map of the stream (test infrastructure)."""
    starts, ends = _oracle.inflate_blocks(comp)

    def rng(lo_bit, own_bit, exact):
        ks = [k for k, s in enumerate(starts) if lo_bit <= s < own_bit]
        if not ks:
            return np.zeros(0, dtype=np.uint8), 0, lo_bit, lo_bit, 0, False
        if exact and starts[ks[0]] != max(16, lo_bit):
            return None
        a, b = ks[0], ks[-1]
        o0 = ends[a - 1] if a else 0
        eb = starts[b + 1] if b + 1 < len(starts) else 0  # (behind the last block: unused by the chain check)
        return want[o0: ends[b]].copy(), ends[b] - o0, starts[a], eb, len(ks), b == len(starts) - 1

    return rng


def _inflate_split_worker(rank, world, port, q):
    """One stream over the ranks: bit ranges, the six-number exchange, the chain check, the gather —
    and the fallback when the ranges do not fit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    import _oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = ge.load()
    shard = _load_shard()
    ok = True
    for kind, seed, n in (("itext", 21, 5 * 131072 + 777), ("xorshift", 4, 3 * 131072), ("lowent4k", 8, 131072 * 9 + 5), ("itext", 2, 40000)):
        data = z.gen(kind, seed, n)
        comp = _oracle.deflate(data)
        fell = []

        def fallback():
            fell.append(1)
            return torch.from_numpy(_oracle.inflate(comp))

        res = shard.inflate_split(len(comp), _oracle_range(_oracle, comp, data), fallback)
        if rank == 0:
            ok = ok and not fell and res.numpy().tobytes() == data.tobytes()
        else:
            assert res is None
        # a rank whose range is not a clean chain: every rank sees it in the table, rank 0 decodes the whole stream
        inner = _oracle_range(_oracle, comp, data)
        res = shard.inflate_split(len(comp), (lambda lo, own, ex: None if rank == world - 1 else inner(lo, own, ex)), fallback)
        if rank == 0:
            ok = ok and len(fell) == 1 and res.numpy().tobytes() == data.tobytes()
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_inflate_split_one_stream_over_ranks_gloo(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inflate_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


def test_chain_check():
    shard = _load_shard()
    B = 131072
    good = [[1, 16, 5000, 2, 2 * B, 0], [1, 5000, 5000, 0, 0, 0], [1, 5000, 9000, 1, 70, 1]]
    assert shard.check_chain(good)
    for r, f, v in ((0, 1, 24), (0, 2, 5001), (0, 4, 2 * B - 1), (0, 5, 1), (2, 5, 0), (2, 4, B + 1), (1, 0, 0)):
        bad = [list(m) for m in good]
        bad[r][f] = v
        assert not shard.check_chain(bad), (r, f, v)
    assert not shard.check_chain([[1, 16, 16, 0, 0, 0]])
    assert shard.split_bits(1000, 4) == [(16, 2000), (2000, 4000), (4000, 6000), (6000, 8000)]
    assert shard.split_bits(3, 4)[0] == (16, 16)


def test_adler_combine_and_block_split():
    import zlib as pyzlib

    shard = _load_shard()
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 300001, dtype=np.uint8).tobytes()
    cuts = [0, 1, 70000, 70000, 131072, 300001]
    parts = [(pyzlib.adler32(a[cuts[i]:cuts[i + 1]]), cuts[i + 1] - cuts[i]) for i in range(len(cuts) - 1)]
    assert shard.adler_combine(parts) == pyzlib.adler32(a)
    assert shard.split_blocks(5 * 131072 + 7, 4) == [(0, 2), (2, 4), (4, 5), (5, 6)]
    assert shard.split_blocks(131072, 8)[0] == (0, 1) and shard.split_blocks(131072, 8)[7] == (1, 1)


def test_partition_is_balanced_and_deterministic():
    import importlib.util

    spec = importlib.util.spec_from_file_location("shard_mod", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    sizes = [1 << 20] * 1024
    owned = shard.partition(sizes, 8)
    assert sorted(sum(owned, [])) == list(range(1024)) and all(len(o) == 128 for o in owned)
    sizes = [5, 1, 9, 3, 7, 2]
    a, b = shard.partition(sizes, 2)
    assert sorted(a + b) == list(range(6)) and abs(sum(sizes[i] for i in a) - sum(sizes[i] for i in b)) <= 2
    assert shard.partition(sizes, 2) == [a, b]


def test_sharded_batch_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True
