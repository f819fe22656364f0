"""zlib.es_amd — Python host mirror of zlib.es's API over the MI355X DEFLATE engine.

The reference's public surface is two functions (``/root/reference/src/zlib.ts:11,25``):

    deflate(input: Uint8Array): Uint8Array
    inflate(input: Uint8Array): Uint8Array

both synchronous, both throwing plain ``Error`` with fixed messages.  This module keeps the
same names, argument meaning and error strings (``ZlibEsError.args[0]`` is the reference's
message) and routes every call through the C-ABI of ``include/zes.h`` (``libzes_hip.so``:
hand-written gfx950 kernels).  There is no CPU implementation here: if the library is missing
or no GPU is usable the call raises, it never falls back.

The directory name contains a dot, so it cannot be imported with a plain ``import``; load it
with ``importlib`` (see ``load()`` in ``__graft_entry__.py``) under the module name
``zlibes_amd``.

Host code in the reference's own language (TypeScript on Node over N-API) lives in
``zlib.es_amd/host/``; this Python mirror drives the same C-ABI for pytest, bench.py and
torch.distributed sharding.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ZES_LIB") or os.path.join(_HERE, "libzes_hip.so")  # (ZES_LIB: development builds of the same library)
_lib = None

BLOCK_MAX_BUFFER_LEN = 131072  # src/const.ts:7

ZES_OK = 0
ZES_E_NOSPACE = -16
ZES_E_DEVICE = -17
ZES_E_ARG = -18
ZES_F_NO_FASTPATH = 1
ZES_F_LOOSE_CANDIDATES = 2
ZES_F_PIECES = 4
ZES_F_ALLOC_BOUND = 8
ZES_ALLOC_EARLY = 0x80000000  # in the allocator's index argument: an early request for an upper estimate (may be declined with NULL)
ZES_E_NOTRANGE = -19

GEN_KINDS = {"xorshift": 0, "lowent4k": 1, "itext": 2}


class ZlibEsError(Exception):
    """Mirror of the reference's ``throw new Error(message)``; ``.code`` is the zes_status."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class ZesKTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_uint32)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64)  # zes_alloc_fn of include/zes.h


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcdir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", srcdir, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", srcdir])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "zlib.es_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % _LIB_PATH)
        L = C.CDLL(_LIB_PATH)
        u64p, u32p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.zes_strerror.restype = C.c_char_p
        L.zes_strerror.argtypes = [C.c_int]
        L.zes_init.argtypes = [C.c_int]
        L.zes_init_devices.argtypes = [C.c_int]
        L.zes_partition.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.zes_device_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), u64p]
        L.zes_deflate_bound.argtypes = [C.c_uint64, u64p]
        for name in ("zes_deflate", "zes_deflate_dev"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        for name in ("zes_inflate", "zes_inflate_dev"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.c_uint32]
        L.zes_inflate_size.argtypes = [C.c_void_p, C.c_uint64, u64p, C.c_uint32]
        L.zes_inflate_alloc.argtypes = [C.c_void_p, C.c_uint64, ALLOC_FN, C.c_void_p, u64p, C.c_uint32]
        L.zes_deflate_batch.argtypes = [C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p), u64p, u64p, i32p, C.c_uint32]
        L.zes_inflate_batch_alloc.argtypes = [C.POINTER(C.c_void_p), u64p, ALLOC_FN, C.c_void_p, u64p, i32p, C.c_uint32, C.c_uint32]
        L.zes_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        L.zes_host_free.argtypes = [C.c_void_p]
        for name in ("zes_deflate_raw", "zes_deflate_raw_dev"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        for name in ("zes_inflate_raw", "zes_inflate_raw_dev"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.c_uint32]
        L.zes_adler32.argtypes = [C.c_void_p, C.c_uint64, u32p]
        L.zes_adler32_dev.argtypes = [C.c_void_p, C.c_uint64, u32p]
        L.zes_deflate_batch_dev.argtypes = [C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, u64p, i32p, C.c_uint32]
        L.zes_inflate_batch_dev.argtypes = [C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, u64p, i32p, C.c_uint32,
                                            C.c_uint32]
        L.zes_deflate_range_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, u64p, u32p]
        L.zes_inflate_range_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, u64p, u64p, u64p,
                                            u32p, i32p]
        L.zes_deflate_join_dev.argtypes = [C.POINTER(C.c_void_p), u64p, u32p, u64p, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
        L.zes_stage_lz77_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, u32p]
        L.zes_stage_huff_lengths_dev.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zes_selftest_lds_order.argtypes = [C.c_uint32, C.c_uint32, u64p, u64p]
        L.zes_last_kernel_times.argtypes = [C.POINTER(ZesKTime), C.c_int]
        L.zes_set_profiling.argtypes = [C.c_int]
        L.zes_pool_bytes.restype = C.c_uint64
        L.zes_pool_bytes.argtypes = []
        L.zes_gen.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def strerror(code):
    return lib().zes_strerror(code).decode()


def _raise(code):
    raise ZlibEsError(code, strerror(code))


def init(device=0):
    rc = lib().zes_init(int(device))
    if rc:
        _raise(rc)


def trim():
    """zes_trim: the pooled device scratch of every context goes back to the driver (the next call allocates again)."""
    rc = lib().zes_trim()
    if rc:
        _raise(rc)


def pool_bytes():
    """zes_pool_bytes: pooled device scratch held right now, all contexts."""
    return int(lib().zes_pool_bytes())


def init_devices(n=0):
    """One process, n GPUs (n <= 0: every visible one): the host batch forms then spread over all of them
    (zes_init_devices).  Returns the number of devices in use."""
    rc = lib().zes_init_devices(int(n))
    if rc:
        _raise(rc)
    return int(lib().zes_device_count())


def partition(sizes, parts):
    """owner[i] of buffer i among `parts` devices: the library's rule for host batches (zes_partition; no GPU needed)."""
    n = len(sizes)
    arr = (C.c_uint64 * max(n, 1))(*[int(x) for x in sizes])
    own = (C.c_uint32 * max(n, 1))()
    rc = lib().zes_partition(arr, n, int(parts), own)
    if rc:
        _raise(rc)
    return [int(own[i]) for i in range(n)]


def device_info():
    name = C.create_string_buffer(64)
    cus = C.c_int()
    hbm = C.c_uint64()
    rc = lib().zes_device_info(name, 64, C.byref(cus), C.byref(hbm))
    if rc:
        _raise(rc)
    return {"arch": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}


def gen(kind, seed, n):
    """Deterministic workload bytes (xorshift / lowent4k / itext), as a numpy uint8 array."""
    out = np.empty(n, dtype=np.uint8)
    rc = lib().zes_gen(out.ctypes.data, n, GEN_KINDS[kind] if isinstance(kind, str) else int(kind), int(seed) & 0xFFFFFFFF)
    if rc:
        _raise(rc)
    return out


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def deflate_bound(n):
    cap = C.c_uint64()
    lib().zes_deflate_bound(n, C.byref(cap))
    return cap.value


# ---- host-buffer API: the drop-in pair ------------------------------------------------------
def deflate(data):
    """``deflate(input)`` of src/zlib.ts:25 — zlib-wrapped, bit-exact; returns a fresh uint8 array."""
    a = _as_u8(data)
    cap = deflate_bound(a.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().zes_deflate(a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc)
    return out[: n.value].copy() if n.value < (1 << 20) else out[: n.value]  # large results: a view, not a second copy


def inflate(data, flags=0):
    """``inflate(input)`` of src/zlib.ts:11 — same accept-set and errors as the reference.

    One library call: it decodes on the device, then asks (callback) for the exact-size result array and copies
    into it — the growable Uint8WriteStream of src/inflate.ts:17 without a second decode or hidden state.
    """
    a = _as_u8(data)
    got = []

    def alloc(_user, _index, n):
        got.append(np.empty(max(int(n), 1), dtype=np.uint8))
        return got[0].ctypes.data

    n = C.c_uint64()
    rc = lib().zes_inflate_alloc(a.ctypes.data, a.size, ALLOC_FN(alloc), None, C.byref(n), flags)
    if rc:
        _raise(rc)
    return got[0][: n.value]


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[x.ctypes.data for x in arrs])


def deflate_batch(buffers):
    """Host-pointer batch: a list of independent buffers in one call → list of uint8 arrays or ZlibEsError (not raised)."""
    arrs = [_as_u8(b) for b in buffers]
    cnt = len(arrs)
    outs = [np.empty(deflate_bound(x.size), dtype=np.uint8) for x in arrs]
    lens = (C.c_uint64 * cnt)(*[x.size for x in arrs])
    caps = (C.c_uint64 * cnt)(*[o.size for o in outs])
    out_len = (C.c_uint64 * cnt)()
    status = (C.c_int32 * cnt)()
    rc = lib().zes_deflate_batch(_ptr_array(arrs), lens, _ptr_array(outs), caps, out_len, status, cnt)
    if rc:
        _raise(rc)
    return [outs[i][: out_len[i]].copy() if status[i] == 0 else ZlibEsError(status[i], strerror(status[i])) for i in range(cnt)]


def inflate_batch(buffers, flags=0):
    arrs = [_as_u8(b) for b in buffers]
    cnt = len(arrs)
    got = {}

    def alloc(_user, index, n):
        got[index] = np.empty(max(int(n), 1), dtype=np.uint8)
        return got[index].ctypes.data

    lens = (C.c_uint64 * cnt)(*[x.size for x in arrs])
    out_len = (C.c_uint64 * cnt)()
    status = (C.c_int32 * cnt)()
    rc = lib().zes_inflate_batch_alloc(_ptr_array(arrs), lens, ALLOC_FN(alloc), None, out_len, status, cnt, flags)
    if rc:
        _raise(rc)
    return [got[i][: out_len[i]] if status[i] == 0 else ZlibEsError(status[i], strerror(status[i])) for i in range(cnt)]


def host_alloc(n):
    """A uint8 numpy array of n bytes in page-locked memory (zes_host_alloc); free with host_free(arr)."""
    p = C.c_void_p()
    rc = lib().zes_host_alloc(n, C.byref(p))
    if rc:
        _raise(rc)
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n]
    return arr


def host_free(arr):
    rc = lib().zes_host_free(arr.ctypes.data)
    if rc:
        _raise(rc)


def deflate_raw(data):
    """Raw DEFLATE: ``deflate(input)`` of src/deflate.ts:14 (what the zlib wrapper of src/zlib.ts:25 encloses)."""
    a = _as_u8(data)
    cap = deflate_bound(a.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().zes_deflate_raw(a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc)
    return out[: n.value].copy()


def inflate_raw(data, offset=0, flags=0):
    """Raw inflate from byte ``offset``: ``inflate(input, offset)`` of src/inflate.ts:16."""
    a = _as_u8(data)
    cap = max(a.size * 8, 1 << 16)
    for _ in range(8):
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_uint64()
        rc = lib().zes_inflate_raw(a.ctypes.data, a.size, offset, out.ctypes.data, cap, C.byref(n), flags)
        if rc == ZES_E_NOSPACE and n.value > cap:
            cap = n.value
            continue
        if rc:
            _raise(rc)
        return out[: n.value].copy()
    _raise(ZES_E_DEVICE)


def adler32(data):
    a = _as_u8(data)
    out = C.c_uint32()
    rc = lib().zes_adler32(a.ctypes.data, a.size, C.byref(out))
    if rc:
        _raise(rc)
    return out.value


# ---- HBM-resident API (torch uint8 CUDA tensors; torch is plumbing for device memory) --------
def _aligned(t):
    """The device forms need 16-byte aligned pointers (include/zes.h); a sliced view such as t[3:] is copied once."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def deflate_tensor(t, out=None):
    """Compress a 1-D uint8 CUDA tensor; returns a view of ``out`` (allocated if None)."""
    import torch

    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
    t = _aligned(t)
    cap = deflate_bound(t.numel())
    if out is None:
        out = torch.empty(cap, dtype=torch.uint8, device=t.device)
    assert out.numel() >= cap
    assert out.data_ptr() % 16 == 0, "deflate_tensor: `out` must start on a 16-byte boundary (include/zes.h: device forms)"
    torch.cuda.current_stream(t.device).synchronize()  # the library runs on its own stream
    n = C.c_uint64()
    rc = lib().zes_deflate_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), C.byref(n))
    if rc:
        _raise(rc)
    return out[: n.value]


def inflate_tensor(t, out, flags=0):
    """Decompress a 1-D uint8 CUDA tensor into ``out``; returns the filled view of ``out``."""
    import torch

    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
    t = _aligned(t)
    assert out.data_ptr() % 16 == 0, "inflate_tensor: `out` must start on a 16-byte boundary (include/zes.h: device forms)"
    torch.cuda.current_stream(t.device).synchronize()
    n = C.c_uint64()
    rc = lib().zes_inflate_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), C.byref(n), flags)
    if rc == ZES_E_NOSPACE:
        err = ZlibEsError(rc, "%s (need %d bytes)" % (strerror(rc), n.value))
        err.need = n.value
        raise err
    if rc:
        _raise(rc)
    return out[: n.value]


def adler32_tensor(t):
    import torch

    torch.cuda.current_stream(t.device).synchronize()
    out = C.c_uint32()
    rc = lib().zes_adler32_dev(t.data_ptr(), t.numel(), C.byref(out))
    if rc:
        _raise(rc)
    return out.value


def _batch_call(fn, d_in, in_off, in_len, d_out, out_off, out_cap, *extra):
    import torch

    cnt = len(in_off)
    arr = lambda v: (C.c_uint64 * cnt)(*[int(x) for x in v])
    out_len = (C.c_uint64 * cnt)()
    status = (C.c_int32 * cnt)()
    torch.cuda.current_stream(d_in.device).synchronize()
    rc = fn(d_in.data_ptr(), arr(in_off), arr(in_len), d_out.data_ptr(), arr(out_off), arr(out_cap), out_len, status, cnt, *extra)
    if rc:
        _raise(rc)
    return list(out_len), list(status)


def deflate_batch_tensor(d_in, in_off, in_len, d_out, out_off, out_cap):
    """Independent buffers inside one arena (offsets 16-byte aligned); returns (out_len[], status[])."""
    return _batch_call(lib().zes_deflate_batch_dev, d_in, in_off, in_len, d_out, out_off, out_cap)


def inflate_batch_tensor(d_in, in_off, in_len, d_out, out_off, out_cap, flags=0):
    return _batch_call(lib().zes_inflate_batch_dev, d_in, in_off, in_len, d_out, out_off, out_cap, flags)


def deflate_range_tensor(t, lo, hi, final, out=None):
    """Raw bit stream of the block range [lo, hi) of the CUDA tensor ``t`` (one buffer split over several GPUs,
    SURVEY §8e-ii) -> (bytes view, nbits, adler32 of the range).  The 258-byte halo behind ``hi`` is read from ``t``."""
    import torch

    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and lo % BLOCK_MAX_BUFFER_LEN == 0
    view = _aligned(t[lo:])  # (lo is a block boundary: aligned whenever t is)
    n = hi - lo
    cap = deflate_bound(n)
    if out is None:
        out = torch.empty(cap, dtype=torch.uint8, device=t.device)
    torch.cuda.current_stream(t.device).synchronize()
    bits, ad = C.c_uint64(), C.c_uint32()
    rc = lib().zes_deflate_range_dev(view.data_ptr(), n, min(view.numel(), n + 258), 1 if final else 0, out.data_ptr(), out.numel(),
                                     C.byref(bits), C.byref(ad))
    if rc:
        _raise(rc)
    return out[: (bits.value + 7) // 8], bits.value, ad.value


def deflate_join_tensors(pieces, bits, adlers, lens, out=None):
    """zlib stream of one buffer from the bit streams of its consecutive block ranges (CUDA tensors on this device)."""
    import torch

    cnt = len(pieces)
    total = 2 + (sum(bits) + 7) // 8 + 4
    if out is None:
        out = torch.empty((total + 15) // 16 * 16, dtype=torch.uint8, device=pieces[0].device)
    torch.cuda.current_stream(out.device).synchronize()
    n = C.c_uint64()
    rc = lib().zes_deflate_join_dev((C.c_void_p * cnt)(*[p.data_ptr() for p in pieces]), (C.c_uint64 * cnt)(*bits),
                                    (C.c_uint32 * cnt)(*adlers), (C.c_uint64 * cnt)(*lens), cnt, out.data_ptr(), out.numel(), C.byref(n))
    if rc:
        _raise(rc)
    return out[: n.value]


def inflate_range_tensor(t, lo_bit, own_bit, exact, out):
    """The blocks of a reference-made zlib stream (CUDA tensor ``t``, the whole stream) that start at bits
    lo_bit <= s < own_bit of it, decoded into ``out`` (block k of the range at k * 131072) -> (out_len, first_bit,
    end_bit, nblocks, final) with bit positions relative to the stream, or None when the range is not a clean chain
    of reference-made blocks (zes_inflate_range_dev; one stream split over several GPUs)."""
    import torch

    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and out.is_cuda
    lo_bit = max(16, int(lo_bit))
    byte0 = (lo_bit >> 3) & ~15
    if byte0 >= 16:
        byte0 -= 16  # (the search starts 16 bits into what it is given)
    end = min(t.numel(), (int(own_bit) + 7) // 8 + (1 << 20))  # the last block's bytes and the header behind it
    view = _aligned(t[byte0:end])
    torch.cuda.current_stream(t.device).synchronize()
    n, fb, eb, nb, fin = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_int32()
    rc = lib().zes_inflate_range_dev(view.data_ptr(), view.numel(), lo_bit - 8 * byte0, int(own_bit) - 8 * byte0, 1 if exact else 0,
                                     out.data_ptr(), out.numel(), C.byref(n), C.byref(fb), C.byref(eb), C.byref(nb), C.byref(fin))
    if rc == ZES_E_NOTRANGE:
        return None
    if rc:
        _raise(rc)
    return n.value, fb.value + 8 * byte0, eb.value + 8 * byte0, nb.value, bool(fin.value)


def last_inflate_tier():
    """1 block-parallel, 2 segment-parallel (any stream), 3 sequential wavefront, 4 exact restatement."""
    return int(lib().zes_last_inflate_tier())


def set_profiling(on):
    lib().zes_set_profiling(1 if on else 0)


def last_kernel_times():
    arr = (ZesKTime * 32)()
    n = lib().zes_last_kernel_times(arr, 32)
    return [(arr[i].name.decode(), float(arr[i].ms), int(arr[i].launches)) for i in range(n)]


# ---- stage-level entries used by the kernel parity tests -------------------------------------
def stage_lz77_tensor(t, start, length):
    tok = np.empty(length + 4, dtype=np.uint32)
    nt = C.c_uint32()
    rc = lib().zes_stage_lz77_dev(t.data_ptr(), t.numel(), start, length, tok.ctypes.data, C.byref(nt))
    if rc:
        _raise(rc)
    return tok[: nt.value].copy()


def selftest_lds_order(iters=200, seed=1):
    """zes_selftest_lds_order: (values out of lane order, values checked) of returning LDS adds on this device."""
    bad, n = C.c_uint64(0), C.c_uint64(0)
    rc = lib().zes_selftest_lds_order(iters, seed, C.byref(bad), C.byref(n))
    if rc:
        _raise(rc)
    return int(bad.value), int(n.value)


def stage_huff_lengths(hist, maxlen):
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    lens = np.zeros(h.size, dtype=np.uint8)
    rc = lib().zes_stage_huff_lengths_dev(h.ctypes.data, h.size, maxlen, lens.ctypes.data)
    if rc:
        _raise(rc)
    return lens
