"""ctypes binding of oracle/libflate_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (flate_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libflate_oracle.so")
_PUFF = os.path.join(_ORACLE_DIR, "_ref", "libpuff.so")

RAW, GZIP, ZLIB = 0, 1, 2
STORE, HUFFMAN = 0, 1

STATUS = {
    0: "Ok", 1: "EndOfStream", 2: "BadGzipHeader", 3: "BadZlibHeader",
    4: "WrongGzipChecksum", 5: "WrongGzipSize", 6: "WrongZlibChecksum",
    7: "InvalidCode", 8: "OversubscribedHuffmanTree", 9: "IncompleteHuffmanTree",
    10: "MissingEndOfBlockCode", 11: "InvalidMatch", 12: "InvalidBlockType",
    13: "WrongStoredBlockNlen", 14: "InvalidDynamicBlockHeader", 100: "OutputTooSmall",
}


def build():
    """Compile the oracle (and oracle/_ref from the reference tree when present)."""
    src = os.path.join(_ORACLE_DIR, "flate_oracle.c")
    need = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if need or (os.path.isdir("/root/reference") and not os.path.exists(_PUFF)):
        subprocess.run(["make", "-C", _ORACLE_DIR], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u16p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)
        szp = C.POINTER(C.c_size_t)
        L.fo_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, szp]
        L.fo_compress_bound.restype = C.c_size_t
        L.fo_compress_bound.argtypes = [C.c_size_t]
        L.fo_tokenize.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, szp]
        L.fo_block_write.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                     C.c_int, C.c_void_p, C.c_size_t, szp]
        L.fo_huffman_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_fixed_literal_codes.argtypes = [C.c_void_p, C.c_void_p]
        L.fo_hash4.restype = C.c_uint32
        L.fo_hash4.argtypes = [C.c_void_p]
        L.fo_lookup_add_all.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_lookup_bulk_add.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.fo_window_match.restype = C.c_uint16
        L.fo_window_match.argtypes = [C.c_void_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_uint16]
        L.fo_length_code.restype = C.c_uint16
        L.fo_length_code.argtypes = [C.c_uint8]
        L.fo_distance_code.restype = C.c_uint8
        L.fo_distance_code.argtypes = [C.c_uint16]
        L.fo_length_extra_bits.restype = C.c_uint8
        L.fo_length_extra_bits.argtypes = [C.c_uint16]
        L.fo_distance_extra_bits.restype = C.c_uint8
        L.fo_distance_extra_bits.argtypes = [C.c_uint8]
        L.fo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, szp, szp]
        L.fo_crc32.restype = C.c_uint32
        L.fo_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.fo_adler32.restype = C.c_uint32
        L.fo_adler32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.fo_deflate_new.restype = C.c_void_p
        L.fo_deflate_new.argtypes = [C.c_int, C.c_int]
        L.fo_deflate_free.argtypes = [C.c_void_p]
        L.fo_deflate_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.fo_deflate_flush.argtypes = [C.c_void_p]
        L.fo_deflate_finish.argtypes = [C.c_void_p]
        L.fo_deflate_output.restype = C.c_void_p
        L.fo_deflate_output.argtypes = [C.c_void_p, szp]
        L.fo_deflate_log_tokens.argtypes = [C.c_void_p, C.c_int]
        L.fo_deflate_token_log.restype = C.c_void_p
        L.fo_deflate_token_log.argtypes = [C.c_void_p, szp]
        L.fo_deflate_block_log.restype = C.c_void_p
        L.fo_deflate_block_log.argtypes = [C.c_void_p, szp]
        _lib = L
    return _lib


def _buf(b):
    """bytes / numpy -> (pointer, length, keepalive)."""
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    return a.ctypes.data if a.size else None, a.size, a


def tok_lit(b):
    return (b & 0xFF) << 15


def tok_match(dist, length):
    return (1 << 23) | ((length - 3) << 15) | (dist - 1)


def tok_decode(t):
    t = int(t)
    if (t >> 23) & 1:
        return ("M", (t & 0x7FFF) + 1, ((t >> 15) & 0xFF) + 3)
    return ("L", (t >> 15) & 0xFF)


def compress(data, container=RAW, mode=6, repair_q1=False):
    """repair_q1: NOT the reference -- the reference with its window advanced before a full token block is flushed (the
    twin of FLATE_HIP_DEFLATE_REPAIR_Q1; oracle/flate_oracle.c, fo_set_q1_repair)."""
    if repair_q1:
        lib().fo_set_q1_repair(1)
        try:
            return compress(data, container, mode)
        finally:
            lib().fo_set_q1_repair(0)
    p, n, keep = _buf(data)
    cap = lib().fo_compress_bound(n)
    out = np.empty(cap, dtype=np.uint8)
    olen = C.c_size_t(0)
    rc = lib().fo_compress(p, n, container, mode, out.ctypes.data, cap, C.byref(olen))
    assert rc == 0, rc
    return out[: olen.value].tobytes()


def tokenize(data, level=6):
    p, n, keep = _buf(data)
    cap = n + 16
    out = np.empty(cap, dtype=np.uint32)
    cnt = C.c_size_t(0)
    rc = lib().fo_tokenize(p, n, level, out.ctypes.data, cap, C.byref(cnt))
    assert rc == 0, rc
    return out[: cnt.value].copy()


def block_write(fn, tokens, eof, input_bytes):
    """fn: 'wb' | 'dyn' | 'huff'.  input_bytes None == Zig null."""
    fnid = {"wb": 0, "dyn": 1, "huff": 2}[fn]
    toks = np.ascontiguousarray(tokens, dtype=np.uint32)
    if input_bytes is None:
        p, n, keep, has = None, 0, None, 0
    else:
        p, n, keep = _buf(input_bytes)
        has = 1
    cap = max(n, 4 * toks.size) + 4096
    out = np.empty(cap, dtype=np.uint8)
    olen = C.c_size_t(0)
    rc = lib().fo_block_write(fnid, toks.ctypes.data if toks.size else None, toks.size, int(eof), p, n, has,
                              out.ctypes.data, cap, C.byref(olen))
    assert rc == 0, rc
    return out[: olen.value].tobytes()


def huffman_generate(freq, max_bits):
    f = np.ascontiguousarray(freq, dtype=np.uint16)
    codes = np.zeros(f.size, dtype=np.uint16)
    lens = np.zeros(f.size, dtype=np.uint16)
    lib().fo_huffman_generate(f.ctypes.data, f.size, max_bits, codes.ctypes.data, lens.ctypes.data)
    return codes, lens


def decompress(data, container=RAW, flags=0, cap=None):
    """returns (status_name, output_bytes, consumed)."""
    p, n, keep = _buf(data)
    if cap is None:
        cap = max(1 << 16, n * 1100 + 1024)
    out = np.empty(cap, dtype=np.uint8)
    olen = C.c_size_t(0)
    used = C.c_size_t(0)
    rc = lib().fo_decompress(p, n, container, flags, out.ctypes.data, cap, C.byref(olen), C.byref(used))
    return STATUS[rc], out[: olen.value].tobytes(), used.value


def crc32(data, start=0):
    p, n, keep = _buf(data)
    return lib().fo_crc32(start, p, n)


def adler32(data, start=1):
    p, n, keep = _buf(data)
    return lib().fo_adler32(start, p, n)


class Deflate:
    """Streaming compressor object (Deflate / SimpleCompressor of deflate.zig)."""

    def __init__(self, container=RAW, mode=6, log_tokens=False):
        self._h = lib().fo_deflate_new(container, mode)
        if log_tokens:
            lib().fo_deflate_log_tokens(self._h, 1)

    def write(self, data):
        p, n, keep = _buf(data)
        lib().fo_deflate_write(self._h, p, n)
        return n

    def flush(self):
        lib().fo_deflate_flush(self._h)

    def finish(self):
        lib().fo_deflate_finish(self._h)

    def output(self):
        n = C.c_size_t(0)
        p = lib().fo_deflate_output(self._h, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def tokens(self):
        n = C.c_size_t(0)
        p = lib().fo_deflate_token_log(self._h, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()

    def blocks(self):
        """(tokens, final, has_input, slice_start, slice_len) of every flushTokens (needs log_tokens)."""
        n = C.c_size_t(0)
        p = lib().fo_deflate_block_log(self._h, C.byref(n))
        if not n.value:
            return []
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n.value, 5)).copy()
        return [tuple(int(x) for x in row) for row in a]

    def close(self):
        if self._h:
            lib().fo_deflate_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the reference's own differential inflater (bin/puff/puff.c), built into oracle/_ref ----
_puff = None


def puff_available():
    build()
    return os.path.exists(_PUFF)


def puff(data, cap=None):
    """returns (rc, output).  rc == 0 success; puff.c:793-840."""
    global _puff
    if _puff is None:
        _puff = C.CDLL(_PUFF)
        _puff.puff.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.POINTER(C.c_ulong)]
        _puff.puff.restype = C.c_int
    p, n, keep = _buf(data)
    if cap is None:
        cap = max(1 << 16, n * 1100 + 1024)
    out = np.empty(cap, dtype=np.uint8)
    dl = C.c_ulong(cap)
    sl = C.c_ulong(n)
    rc = _puff.puff(out.ctypes.data, C.byref(dl), p, C.byref(sl))
    return rc, out[: dl.value].tobytes()
