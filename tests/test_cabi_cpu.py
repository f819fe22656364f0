"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/zes.h declares, maps statuses to the reference's messages, and refuses to compute
without a device (no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zes.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zes_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(z):
    L = z.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libzes_hip.so does not export %s" % n


def test_status_messages_are_the_reference_strings(z):
    # src/zlib.ts:15, src/inflate.ts:32,35,50, src/utils/BitReadStream.ts:15
    assert z.strerror(-1) == "Not compressed by deflate"
    assert z.strerror(-2) == "Not supported BTYPE : 3"
    assert z.strerror(-3) == "Data is corrupted"
    assert z.strerror(-4) == "Data length is insufficient"
    assert z.strerror(-5) == "Lack of data length"


def test_deflate_bound_matches_reference_heap(z):
    # src/deflate.ts:16 (+6 for the wrapper, src/zlib.ts:42)
    assert z.deflate_bound(0) == 131072 + 6
    assert z.deflate_bound(65535) == 131072 + 6
    assert z.deflate_bound(65536) == 131072 + 6
    assert z.deflate_bound(1 << 20) == (2 << 20) + 6


def test_no_cpu_fallback_without_gpu(z):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    a = np.arange(100, dtype=np.uint8)
    with pytest.raises(z.ZlibEsError) as ei:
        z.deflate(a)
    assert ei.value.code == z.ZES_E_DEVICE
    with pytest.raises(z.ZlibEsError) as ei:
        z.inflate(bytes([0x78, 0x9C, 3, 0]))
    assert ei.value.code == z.ZES_E_DEVICE


def test_throw_cases_are_decided_before_the_device(z):
    # n = 0, 1, 131073 throw 'Data is corrupted' in the reference; the library
    # answers from the host, so this holds even without a GPU
    L = z.lib()
    out = np.zeros(1 << 19, dtype=np.uint8)
    n = C.c_uint64()
    for ln in (0, 1, 131073):
        a = np.zeros(max(ln, 1), dtype=np.uint8)
        assert L.zes_deflate(a.ctypes.data, ln, out.ctypes.data, out.size, C.byref(n)) == -3


def test_generators_are_deterministic(z):
    a = z.gen("xorshift", 12345, 64)
    assert a[:4].tolist() == z.gen("xorshift", 12345, 4).tolist()
    t = z.gen("itext", 7, 200).tobytes()
    assert t == z.gen("itext", 7, 200).tobytes() and b" " in t
    lo = z.gen("lowent4k", 3, 9000)
    assert (lo[:4096] == lo[4096:8192]).all()


def test_partition_of_a_host_batch_is_the_rule_of_shard_partition(z):
    """zes_partition (what zes_deflate_batch / zes_inflate_batch_alloc cut a batch with when the library drives several
    GPUs from one process) against zlib.es_amd/shard.py's partition(): the same owners, so that a Node batch and a
    torch.distributed job split the same work the same way.  Pure host code: no GPU needed."""
    import importlib.util
    import random

    spec = importlib.util.spec_from_file_location("shard_mod", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    rnd = random.Random(7)
    cases = [[], [5], [1 << 20] * 1024, [3, 3, 3, 3, 3], [10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
             [rnd.randrange(0, 1 << 22) for _ in range(300)], [rnd.choice((0, 1, 2, 131072, 131073, 1 << 20)) for _ in range(77)]]
    for sizes in cases:
        for parts in (1, 2, 3, 8, 16):
            own = z.partition(sizes, parts)
            want = [None] * len(sizes)
            for r, ids in enumerate(shard.partition(sizes, parts)):
                for i in ids:
                    want[i] = r
            assert own == want, (sizes[:8], parts)
            load = [sum(s for s, o in zip(sizes, own) if o == r) for r in range(parts)]
            if sizes:
                assert max(load) - min(load) <= max(sizes)  # greedy longest-first: no part is ahead by more than one buffer
