"""Host-side mirror of the reference's public interface (src/flate.zig:9-71,
src/gzip.zig:4-66, src/zlib.zig:4-66): the same names, argument meaning and error
behaviour, over the C ABI of libflate_hip.so.

  compress(reader, writer, options)     flate.zig:28-30   (one-shot: compressor + compress + finish)
  Compressor / compressor(writer, opt)  flate.zig:33-40   (write / compress / finish; flush: see below)
  decompress(reader, writer)            flate.zig:10-12
  Decompressor / decompressor(reader)   flate.zig:15-22   (decompress / next / read / reader / reset / set_reader)
  huffman.{compress,Compressor,compressor}   flate.zig:44-56
  store.{compress,Compressor,compressor}     flate.zig:59-71
  Options(level=Level.default), Level   deflate.zig:15-32

`reader` is anything with .read() (or a bytes-like), `writer` anything with .write().
Errors are FlateError subclasses named after the reference's error set
(inflate.zig:72-78, huffman_decoder.zig:35-40, container.zig:45-51, bit_reader.zig:29).

Scope of this round: one-shot streams of any length.  At levels 4..9 an
input longer than 65535 bytes is compressed as ONE stream by the whole-stream path
(SURVEY.md 8f-2), byte-identical to the reference's sliding-window compressor.
Compressor.flush (deflate.zig:335-337; 474-478 for the huffman-only / store-only
compressors) is a sync flush that keeps the LZ history.  It is incremental: the object
keeps only the tail of what was written that the reference's window, slide schedule and
pending block can still depend on (from a 32 KiB-aligned position at least 96 KiB before
the previous flush), compresses that tail with its flush points on the GPU and hands the
writer the bytes that follow what the previous flush produced; the container checksum is
folded piece by piece.
"""
import enum
import io

from . import _capi
from .engine import default_engine


class Level(enum.IntEnum):  # deflate.zig:23-32
    fast = 4
    level_4 = 4
    level_5 = 5
    default = 6
    level_6 = 6
    level_7 = 7
    level_8 = 8
    best = 9
    level_9 = 9


class Options:  # deflate.zig:15-17
    def __init__(self, level=Level.default, repair_q1=False):
        self.level = Level(level)
        self.repair_q1 = bool(repair_q1)  # not in the reference: FLATE_HIP_DEFLATE_REPAIR_Q1 (ReferenceQ1StreamWarning)


class FlateError(Exception):
    status = None


def _mk(name, code):
    return type(name, (FlateError,), {"status": code})


EndOfStream = _mk("EndOfStream", 1)
BadGzipHeader = _mk("BadGzipHeader", 2)
BadZlibHeader = _mk("BadZlibHeader", 3)
WrongGzipChecksum = _mk("WrongGzipChecksum", 4)
WrongGzipSize = _mk("WrongGzipSize", 5)
WrongZlibChecksum = _mk("WrongZlibChecksum", 6)
InvalidCode = _mk("InvalidCode", 7)
OversubscribedHuffmanTree = _mk("OversubscribedHuffmanTree", 8)
IncompleteHuffmanTree = _mk("IncompleteHuffmanTree", 9)
MissingEndOfBlockCode = _mk("MissingEndOfBlockCode", 10)
InvalidMatch = _mk("InvalidMatch", 11)
InvalidBlockType = _mk("InvalidBlockType", 12)
WrongStoredBlockNlen = _mk("WrongStoredBlockNlen", 13)
InvalidDynamicBlockHeader = _mk("InvalidDynamicBlockHeader", 14)
OutputTooSmall = _mk("OutputTooSmall", 100)
ChunkTooLarge = _mk("ChunkTooLarge", 101)
InvalidState = _mk("InvalidState", 103)  # inflate.zig:303 (a host-side state, no device status)


class ReferenceQ1StreamWarning(UserWarning):
    """The stream just written is byte for byte the reference's -- and does not inflate to its input: the reference flushes
    a full block of 32768 tokens before its window has advanced over the last token's match (deflate.zig:227-230 before
    :193), and when exactly one of the two blocks at that seam is stored the match's bytes are lost or written twice
    (include/flate_hip.h: FLATE_HIP_ST_REFERENCE_Q1_STREAM).  `Options(level, repair_q1=True)` writes a stream that
    inflates to the input instead."""

_ERRORS = {e.status: e for e in (
    EndOfStream, BadGzipHeader, BadZlibHeader, WrongGzipChecksum, WrongGzipSize, WrongZlibChecksum, InvalidCode,
    OversubscribedHuffmanTree, IncompleteHuffmanTree, MissingEndOfBlockCode, InvalidMatch, InvalidBlockType,
    WrongStoredBlockNlen, InvalidDynamicBlockHeader, OutputTooSmall, ChunkTooLarge)}


def raise_for_status(code):
    if code:
        raise _ERRORS.get(code, FlateError)(_capi.status_name(code))


def _compress_status(code):
    """Status of a compress call: 102 is the reference's own (broken) stream -- written as the reference writes it, with a
    warning; anything else non-zero raises."""
    if code == _capi.ST_REFERENCE_Q1_STREAM:
        import warnings
        warnings.warn(ReferenceQ1StreamWarning(ReferenceQ1StreamWarning.__doc__.split("\n")[0]), stacklevel=3)
        return
    raise_for_status(code)


def _read_all(reader):
    if isinstance(reader, (bytes, bytearray, memoryview)):
        return bytes(reader)
    return reader.read()


_HEADERS = {_capi.RAW: b"", _capi.GZIP: bytes([0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0x03]), _capi.ZLIB: bytes([0x78, 0x9C])}
_KEEP = 98304   # history a piece after a flush can depend on: 64 KiB window + one 32 KiB slide step
_STEP = 32768


class _Compressor:
    """Deflate (deflate.zig:121-373) / SimpleCompressor (:449-529) seen from the caller.

    One-shot use (write / compress, then finish) is one GPU call over the whole input.  With flush()
    (deflate.zig:335-337: pending tokens out, then an empty stored block; the LZ history stays) the
    object works INCREMENTALLY: what the reference emits for the bytes after a flush point F depends
    only on the stream from a 32 KiB-aligned position B <= F - 96 KiB on (its 64 KiB window has slid
    past everything older, and the slide schedule is periodic in 32 KiB), so each flush() / finish()
    runs flate_hip_compress_flush on the retained tail [B, now) only and hands the writer the bytes
    after the previous flush's marker.  Cost per flush: O(new bytes + 128 KiB), memory: the tail.
    The container header is written once, the footer's checksum is folded from per-piece checksums
    (flate_hip_checksum / _combine)."""

    def __init__(self, container, mode, writer, engine=None, repair_q1=False):
        self._container, self._mode, self._wrt = container, int(mode), writer
        self._eng = engine or default_engine()
        self._repair = bool(repair_q1)
        self._buf = bytearray()  # stream bytes from absolute position self._base on
        self._base = 0
        self._total = 0          # bytes written so far
        self._flushes = []       # absolute flush points >= self._base
        self._nflush = 0         # flush() calls so far
        self._rel_emitted = None  # bytes the tail [base, last flush) compresses to (cached while base stands)
        self._cks, self._cks_pos = None, 0
        self._done = False

    def _call(self, fn, *args):
        # (the flag is the handle's: set for this object's calls only)
        if not self._repair:
            return fn(*args)
        self._eng.set_flags(_capi.DEFLATE_REPAIR_Q1)
        try:
            return fn(*args)
        finally:
            self._eng.set_flags(0)

    def _live(self):
        # the reference's compressor has no such check (writing after finish() emits a broken stream);
        # the mirror refuses instead of handing the writer bytes that do not extend the finished stream
        if self._done:
            raise InvalidState("compressor used after finish()")

    def write(self, data):  # deflate.zig:363-367
        self._live()
        self._buf += data
        self._total += len(data)
        return len(data)

    def compress(self, reader):  # deflate.zig:304-321
        self._live()
        data = _read_all(reader)
        self._buf += data
        self._total += len(data)

    def writer(self):  # deflate.zig:369-371
        return self

    def _fold_checksum(self):
        """checksum of everything written so far (gzip / zlib footer), piece by piece"""
        if self._container == _capi.RAW:
            return
        new = bytes(self._buf[self._cks_pos - self._base:])
        v = self._eng.checksum(new, self._container)
        self._cks = v if self._cks is None else self._eng.checksum_combine(self._container, self._cks, v, len(new))
        self._cks_pos = self._total

    def _run_tail(self, finish):
        """compress the retained tail with its flush points (raw container); returns the new bytes"""
        rel = [f - self._base for f in self._flushes]
        tail = bytes(self._buf)
        if self._rel_emitted is None:
            # how much of the tail's output was handed over already: the tail up to the previous flush
            prev = rel[-2] if not finish else rel[-1]
            done, st = self._call(self._eng.compress_flush, tail[:prev], rel[:-1] if not finish else rel, False, _capi.RAW, self._mode)
            _compress_status(st)
            self._rel_emitted = len(done)
        out, st = self._call(self._eng.compress_flush, tail, rel, finish, _capi.RAW, self._mode)
        _compress_status(st)
        new = out[self._rel_emitted:]
        self._rel_emitted = len(out)
        return new

    def _drop_history(self):
        last = self._flushes[-1]
        nb = ((last - _KEEP) // _STEP) * _STEP
        if nb > self._base:
            del self._buf[: nb - self._base]
            self._flushes = [f for f in self._flushes if f >= nb]
            self._base = nb
            self._rel_emitted = None

    def flush(self):  # deflate.zig:335-337: pending tokens out, then an empty stored block; history stays
        self._live()
        self._fold_checksum()
        first = self._nflush == 0
        self._flushes.append(self._total)
        self._nflush += 1
        if first:
            self._rel_emitted = 0
            self._wrt.write(_HEADERS[self._container])
        self._wrt.write(self._run_tail(False))
        self._drop_history()

    def set_writer(self, new_writer):  # deflate.zig:351-354
        self._wrt = new_writer

    def finish(self):  # deflate.zig:344-347
        if self._done:
            return
        if self._nflush:
            self._fold_checksum()
            self._wrt.write(self._run_tail(True))
            if self._container == _capi.GZIP:  # container.zig:92-96
                self._wrt.write(self._cks.to_bytes(4, "little") + (self._total & 0xFFFFFFFF).to_bytes(4, "little"))
            elif self._container == _capi.ZLIB:  # container.zig:104
                self._wrt.write(self._cks.to_bytes(4, "big"))
        else:
            outs, st = self._call(self._eng.compress_many, [bytes(self._buf)], self._container, self._mode)
            _compress_status(st[0])
            self._wrt.write(outs[0])
        self._done = True


class _Decompressor:
    """Inflate (inflate.zig:43-355) seen from the caller.  The reader is consumed as far as the current stream
    needs it, in steps that double: a decode that runs out of input (EndOfStream) while the reader still has
    bytes is repeated with twice as much -- the input read past the end of a stream stays buffered for reset()
    (inflate.zig:301-309), and the GPU work is at most twice that of the final decode."""

    CHUNK = 65536  # the reference hands out at most its 64 KiB ring per next() (inflate.zig:322-336)
    FIRST_READ = 1 << 16

    def __init__(self, container, reader, engine=None, flags=0):
        self._container, self._flags = container, flags
        self._eng = engine or default_engine()
        self._set_input(reader)
        self._out = None   # decoded bytes of the current stream
        self._rp = 0
        self._ended = False

    def _set_input(self, reader):
        if isinstance(reader, (bytes, bytearray, memoryview)):
            reader = io.BytesIO(bytes(reader))
        self._rd = reader
        self._in = bytearray()  # bytes read from the reader and not yet given up
        self._pos = 0           # start of the current stream in _in
        self._eof = False

    def _fill(self, want):
        """Have `want` bytes of the current stream buffered, or the reader at its end."""
        while not self._eof and len(self._in) - self._pos < want:
            got = self._rd.read(want - (len(self._in) - self._pos))
            if not got:
                self._eof = True
                break
            self._in += got

    def _decode(self):
        if self._out is not None:
            return
        if self._pos > (1 << 20):  # drop what earlier streams consumed
            del self._in[:self._pos]
            self._pos = 0
        want = self.FIRST_READ
        while True:
            self._fill(want)
            data = bytes(self._in[self._pos:])
            cap = max(1 << 16, len(data) * 64)
            while True:
                outs, st, used = self._eng.decompress_many([data], self._container, self._flags, caps=[cap])
                if st[0] == 100 and cap < (1 << 34):
                    cap *= 8
                    continue
                break
            if st[0] == 1 and not self._eof:  # EndOfStream with input still to come: not an error yet
                want = 2 * max(want, len(data))
                continue
            break
        raise_for_status(st[0])
        self._out, self._used = outs[0], used[0]

    def get(self, limit=0):  # inflate.zig:326-336
        self._decode()
        n = len(self._out) - self._rp
        n = min(n, limit if limit else self.CHUNK)
        buf = self._out[self._rp:self._rp + n]
        self._rp += n
        if n == 0:
            self._ended = True
        return buf

    def next(self):  # inflate.zig:315-319
        buf = self.get(0)
        return buf if buf else None

    def read(self, n=-1):  # inflate.zig:343-347 (n < 0: read to the end, Python convention)
        if n is None or n < 0:
            self._decode()
            buf = self._out[self._rp:]
            self._rp = len(self._out)
            self._ended = True
            return buf
        return self.get(n) if n else b""

    def reader(self):  # inflate.zig:349-351
        return self

    def decompress(self, writer):  # inflate.zig:292-296
        while True:
            buf = self.next()
            if buf is None:
                break
            writer.write(buf)

    def reset(self):  # inflate.zig:301-309: next stream of the same reader
        if not self._ended:
            raise InvalidState("reset() before the end of the stream")
        self._pos += self._used
        self._out, self._rp, self._ended = None, 0, False

    def more_input(self):
        """True when input is left after the stream just decoded (a further concatenated member)."""
        self._decode()
        if self._pos + self._used >= len(self._in):
            self._fill(self._used + 1)
        return self._pos + self._used < len(self._in)

    def set_reader(self, new_reader):  # inflate.zig:283-288
        self._set_input(new_reader)
        self._out, self._rp, self._ended = None, 0, False


class _Simple:
    """huffman / store namespaces (flate.zig:44-71)."""

    def __init__(self, container, mode):
        self._container, self._mode = container, mode

    def compress(self, reader, writer, engine=None):
        c = self.compressor(writer, engine)
        c.compress(reader)
        c.finish()

    def compressor(self, writer, engine=None):
        return _Compressor(self._container, self._mode, writer, engine)

    Compressor = compressor


class ContainerModule:
    """One of flate (raw) / gzip / zlib: identical function set (readme.md:104-124)."""

    def __init__(self, container):
        self._container = container
        self.huffman = _Simple(container, _capi.MODE_HUFFMAN)
        self.store = _Simple(container, _capi.MODE_STORE)
        self.Options, self.Level = Options, Level

    def compress(self, reader, writer, options=None, engine=None):
        c = self.compressor(writer, options, engine)
        c.compress(reader)
        c.finish()

    def compressor(self, writer, options=None, engine=None):
        options = options or Options()
        return _Compressor(self._container, int(options.level), writer, engine, repair_q1=options.repair_q1)

    Compressor = compressor

    def decompress(self, reader, writer, engine=None):
        self.decompressor(reader, engine).decompress(writer)

    def decompressor(self, reader, engine=None):
        return _Decompressor(self._container, reader, engine)

    Decompressor = decompressor


def compress_bytes(data, container=_capi.RAW, mode=6, engine=None):
    w = io.BytesIO()
    c = _Compressor(container, mode, w, engine)
    c.compress(data)
    c.finish()
    return w.getvalue()
