"""ctypes binding of oracle/libzes_oracle.so — the CPU restatement of the reference.

Test infrastructure only (see oracle/zes_oracle.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

MESSAGES = {
    -1: "Not compressed by deflate",
    -2: "Not supported BTYPE : 3",
    -3: "Data is corrupted",
    -4: "Data length is insufficient",
    -5: "Lack of data length",
}


def build():
    so = os.path.join(ROOT, "oracle", "libzes_oracle.so")
    src = os.path.join(ROOT, "oracle", "zes_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.zor_adler32.argtypes = [C.c_void_p, C.c_uint64, u32p]
        L.zor_deflate_bound.argtypes = [C.c_uint64, u64p]
        L.zor_deflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        L.zor_inflate.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), u64p]
        L.zor_inflate_raw.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), u64p]
        L.zor_deflate_raw.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
        L.zor_deflate_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, u64p]
        L.zor_inflate_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, u32p]
        L.zor_lz77_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, u32p]
        L.zor_huff_lengths.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zor_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__(MESSAGES.get(code, "oracle error %d" % code))


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def adler32(data):
    a = _as_u8(data)
    out = C.c_uint32()
    lib().zor_adler32(a.ctypes.data, a.size, C.byref(out))
    return out.value


def deflate(data):
    a = _as_u8(data)
    cap = C.c_uint64()
    lib().zor_deflate_bound(a.size, C.byref(cap))
    out = np.empty(cap.value, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().zor_deflate(a.ctypes.data, a.size, out.ctypes.data, cap.value, C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[: n.value].copy()


def inflate(data):
    a = _as_u8(data)
    p = C.c_void_p()
    n = C.c_uint64()
    rc = lib().zor_inflate(a.ctypes.data, a.size, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[: n.value].copy()
    lib().zor_free(p)
    return out


def inflate_raw(data, offset=0):
    a = _as_u8(data)
    p = C.c_void_p()
    n = C.c_uint64()
    rc = lib().zor_inflate_raw(a.ctypes.data, a.size, offset, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[: n.value].copy()
    lib().zor_free(p)
    return out


def deflate_raw(data):
    a = _as_u8(data)
    cap = C.c_uint64()
    lib().zor_deflate_bound(a.size, C.byref(cap))
    out = np.empty(cap.value, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().zor_deflate_raw(a.ctypes.data, a.size, out.ctypes.data, cap.value, C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[: n.value].copy()


def lz77_block(data, start, length):
    a = _as_u8(data)
    tok = np.empty(length + 4, dtype=np.uint32)
    nt = C.c_uint32()
    rc = lib().zor_lz77_block(a.ctypes.data, a.size, start, length, tok.ctypes.data, C.byref(nt))
    if rc:
        raise OracleError(rc)
    return tok[: nt.value].copy()


def huff_lengths(hist, maxlen):
    h = np.ascontiguousarray(hist, dtype=np.uint32)
    lens = np.zeros(h.size, dtype=np.uint8)
    lib().zor_huff_lengths(h.ctypes.data, h.size, maxlen, lens.ctypes.data)
    return lens


def deflate_range(data, start, length, final):
    """Raw bit stream of the blocks [start, start + length) of `data` -> (bytes, nbits)."""
    a = _as_u8(data)
    cap = max(2 * length, 131072) + 64
    out = np.zeros(cap, dtype=np.uint8)
    bits = C.c_uint64()
    rc = lib().zor_deflate_range(a.ctypes.data, a.size, start, length, 1 if final else 0, out.ctypes.data, cap, C.byref(bits))
    if rc:
        raise OracleError(rc)
    return out[: (bits.value + 7) // 8].copy(), bits.value


def inflate_blocks(data, offset=2):
    """Map of a stream: (start bit of every block, output length behind every block)."""
    a = _as_u8(data)
    cap = a.size // 8 + 16
    sb, oe = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64)
    k = C.c_uint32()
    rc = lib().zor_inflate_blocks(a.ctypes.data, a.size, offset, sb.ctypes.data, oe.ctypes.data, cap, C.byref(k))
    if rc:
        raise OracleError(rc)
    return [int(x) for x in sb[: k.value]], [int(x) for x in oe[: k.value]]
