"""GPU parity tests of the compress path: HIP output == CPU oracle output, byte for
byte (integer/byte work: the bar is bit-exact), through the C ABI."""
import io
import json
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN, golden
from gpu_util import engine

pytestmark = pytest.mark.gpu

WBITS = {0: -15, 1: 31, 2: 15}


def _rng_cases():
    rng = np.random.default_rng(2024)
    from flate_amd import synth
    cases = {
        "empty": b"",
        "one": b"a",
        "three": b"abc",
        "four": b"abcd",
        "blah": b"Blah blah blah blah blah!",
        "abcde": b"ABCDEABCD ABCDEABCD",
        "zeros262": bytes(262),
        "zeros263": bytes(263),
        "zeros65535": bytes(65535),
        "ff1000": b"\xff" * 1000,              # one repeated byte: answered without a sort (k_lz_match)
        "a66": b"a" * 66,
        "zeros_then_x": bytes(65534) + b"x",  # ... and almost: the ordinary path
        "x_then_zeros": b"x" + bytes(65534),
        "zeros_x_zeros": bytes(40000) + b"x" + bytes(25534),
        "rfc": golden("rfc1951.txt"),
        "text64k": synth.text(synth.SEED_TEXT, 65535).tobytes(),
        "text_odd": synth.text(synth.SEED_TEXT + 1, 40001).tobytes(),
        "rand1k": rng.integers(0, 256, 1000, dtype=np.uint8).tobytes(),
        "rand65535": rng.integers(0, 256, 65535, dtype=np.uint8).tobytes(),
        "noise4": rng.integers(97, 101, 65535, dtype=np.uint8).tobytes(),
        "period16": (bytes(range(16)) * 5000)[:65535],
        "sparse": bytes(b if (i % 97 == 0) else 0 for i, b in enumerate(rng.integers(0, 256, 50000, dtype=np.uint8))),
        "pi": golden("block_writer", "huffman-pi.input"),
        "shifts": golden("block_writer", "huffman-shifts.input"),
    }
    return cases


CASES = _rng_cases()


def test_tokenizer_matches_oracle_tokens():
    eng = engine()
    names = list(CASES)
    for level in (4, 5, 6, 7, 8, 9):
        outs, st = eng.compress_many([CASES[n] for n in names], O.RAW, level)
        assert st == [0] * len(names)
        for i, n in enumerate(names):
            want = O.tokenize(CASES[n], level)
            got = eng.debug_tokens(i)
            assert len(got) == len(want), (n, level, len(got), len(want))
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (n, level, int(bad[0]), O.tok_decode(got[bad[0]]), O.tok_decode(want[bad[0]]))


@pytest.mark.parametrize("container", [0, 1, 2])
@pytest.mark.parametrize("mode", [0, 1, 4, 6, 9])
def test_bytes_match_oracle(container, mode):
    eng = engine()
    names = list(CASES)
    outs, st = eng.compress_many([CASES[n] for n in names], container, mode)
    assert st == [0] * len(names)
    for n, got in zip(names, outs):
        want = O.compress(CASES[n], container, mode)
        assert got == want, (n, container, mode, len(got), len(want))
        assert pyzlib.decompress(got, WBITS[container]) == CASES[n]


@pytest.mark.parametrize("mode", [0, 1])
def test_simple_modes_multi_block_streams(mode):
    # huffman-only / store-only streams longer than one 65535-byte block, incl. the exact
    # multiples that end with an empty final block (Q5, deflate.zig:498-511 + 480-484)
    eng = engine()
    from flate_amd import synth
    base = synth.silesia_like(synth.SEED_SILESIA, 3 * 65535 + 777).tobytes()
    datas = [base, base[:65535], base[:2 * 65535], base[:65536], base[100:100 + 200000]]
    for container in (0, 1, 2):
        outs, st = eng.compress_many(datas, container, mode)
        assert st == [0] * len(datas)
        for d, got in zip(datas, outs):
            assert got == O.compress(d, container, mode)


def test_reference_goldens_through_the_gpu():
    # block_writer.zig:599-706: a chunk whose tokens / bytes are the golden case must yield the
    # golden block (BFINAL set, the stream being a single final block)
    eng = engine()
    with open(os.path.join(GOLDEN, "block_writer_tokens.json")) as f:
        cases = json.load(f)
    names = [c["input"] for c in cases if c["input"]] + ["huffman-rand-max.input"]
    datas = [golden("block_writer", n) for n in names]
    outs, st = eng.compress_many(datas, 0, 1)
    assert st == [0] * len(datas)
    for n, d, got in zip(names, datas, outs):
        want = bytearray(golden("block_writer", n.replace(".input", ".huff.expect")))
        if len(d) % 65535 == 0:
            # a full 65535-byte buffer is written as a non-final block and followed by an
            # empty final block (Q5, deflate.zig:498-511 + 480-484)
            assert got[:len(want) - 1] == bytes(want[:-1]), n
            assert got == O.compress(d, 0, 1), n
        else:
            want[0] |= 1
            assert got == bytes(want), n


def test_known_answer_sizes_and_config1_vector(rfc1951):
    # flate.zig:101-124 (sizes) through the GPU; rfc1951.txt is 36944 bytes: one chunk
    eng = engine()
    sizes = {4: 11513, 5: 11217, 6: 11139, 7: 11126, 8: 11122, 9: 11119}
    for level, gz in sizes.items():
        outs, st = eng.compress_many([rfc1951], 1, level)
        assert st == [0] and len(outs[0]) == gz
    outs, _ = eng.compress_many([rfc1951], 1, 1)
    assert len(outs[0]) == 20287
    outs, _ = eng.compress_many([rfc1951], 1, 0)
    assert len(outs[0]) == 36967
    # deflate.zig:721-748
    outs, _ = eng.compress_many([b"Hello world!", b"Hello world!"], 0, 0)
    assert outs[0] == bytes([1, 0xC, 0, 0xF3, 0xFF]) + b"Hello world!"
    outs, _ = eng.compress_many([b"Hello world!"], 0, 1)
    assert outs[0] == bytes([1, 0xC, 0, 0xF3, 0xFF]) + b"Hello world!"


def test_q1_token_flush_input_slice():
    # 32768th token is a match -> the stored/huffman decision of the two
    # blocks sees the reference's (shifted) input slices.  Whatever the reference does, we do.
    eng = engine()
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 32767, dtype=np.uint8).tobytes()
    rep = a[100:140]
    tail = (b"compressible tail " * 2000)[:30000]
    datas = [a + rep + tail, a + a[:20] + tail, a[:32766] + rep + tail, a[:32768] + tail]
    for level in (4, 6, 9):
        outs, st = eng.compress_many(datas, 0, level)
        assert st == [0] * len(datas)
        for d, got in zip(datas, outs):
            assert got == O.compress(d, 0, level)


def _q1_inputs(level):
    """Inputs whose 32768th token is a match next to a stored block (quirk Q1), made with the oracle's
    tokenizer: `lost` -- 32767 tokens of noise (the block goes out stored), a 40-byte copy, text (Huffman): the reference's
    stream loses the copy's bytes; `twice` -- 32767 tokens over 64 symbols (Huffman), a 16-byte copy, 600 bytes of noise
    (stored): the copy's bytes come out twice."""
    def covered(data, ntok):
        pos = 0
        for t in O.tokenize(data, level)[:ntok]:
            t = int(t)
            pos += ((t >> 15) & 0xFF) + 3 if (t >> 23) & 1 else 1
        return pos

    out = {}
    for seed in (0, 1):
        a = np.random.default_rng(seed).integers(0, 256, 33000, dtype=np.uint8).tobytes()
        out["lost%d" % seed] = a[:covered(a, 32767)] + a[100:140] + (b"compressible tail " * 2000)[:30000]
        h = np.random.default_rng(seed).integers(0, 64, 40000, dtype=np.uint8).tobytes()
        out["twice%d" % seed] = (h[:covered(h, 32767)] + h[200:216] +
                                 np.random.default_rng(1000 + seed).integers(0, 256, 600, dtype=np.uint8).tobytes())
    # ... and the same seam in a stream that slides its window (the whole-stream path: k_st_emit / k_st_blocks)
    a = np.random.default_rng(7).integers(0, 256, 33000, dtype=np.uint8).tobytes()
    out["lost_long"] = a[:covered(a, 32767)] + a[100:140] + (b"compressible tail, and a long one " * 6000)[:150000]
    return out


@pytest.mark.parametrize("level", [4, 6, 9])
def test_q1_streams_are_reported_and_repairable(level):
    # the reference's Q1 stream is replicated byte for byte (status 102, informational) and
    # FLATE_HIP_DEFLATE_REPAIR_Q1 writes a stream that inflates to the input -- under the oracle, puff and zlib.
    # deflate.zig:193, 227-230, 268-288; SlidingWindow.zig:119-123.
    import zlib
    from flate_amd import _capi
    eng = engine()
    inputs = _q1_inputs(level)
    names, datas = list(inputs), list(inputs.values())
    n102 = 0
    for container in (0, 1):
        outs, st = eng.compress_many(datas, container, level)
        for nm, d, got, s in zip(names, datas, outs, st):
            assert got == O.compress(d, container, level), (nm, "default bytes are the reference's")
            name, back = O.decompress(got, container, 0, cap=len(d) + 600)[:2]
            broken = not (name == "Ok" and back == d)
            assert s == (_capi.ST_REFERENCE_Q1_STREAM if broken else 0), (nm, s, name, len(back), len(d))
            n102 += broken
    assert n102 >= 4, "the constructions above no longer reach the seam"
    assert _capi.status_name(102) == "ReferenceQ1Stream"
    eng.set_flags(_capi.DEFLATE_REPAIR_Q1)
    try:
        for container in (0, 1, 2):
            outs, st = eng.compress_many(datas, container, level)
            assert st == [0] * len(datas)
            for nm, d, got in zip(names, datas, outs):
                assert got == O.compress(d, container, level, repair_q1=True), (nm, "the oracle's twin of the repair")
                assert O.decompress(got, container, 0, cap=len(d) + 600)[:2] == ("Ok", d), nm
                assert zlib.decompress(got, {0: -15, 1: 31, 2: 15}[container]) == d, nm
                if container == 0:
                    assert O.puff(got) == (0, d), nm
        # flush points around the seam (flate_hip_compress_flush): the piece that holds it behaves the same way
        d = inputs["lost0"]
        fl, s = eng.compress_flush(d, [100, len(d) - 50], True, 0, level)
        assert s == 0 and zlib.decompress(fl, -15) == d
        back, st2, _ = eng.decompress_many(outs, 2, caps=[len(x) + 8 for x in datas])
        assert st2 == [0] * len(datas) and back == datas
    finally:
        eng.set_flags(0)
    # the flag is off again: the reference's bytes
    outs, st = eng.compress_many(datas[:1], 0, level)
    assert outs[0] == O.compress(datas[0], 0, level)
    # the mirror: the reference's bytes with a warning, or Options(repair_q1=True)
    import io
    from flate_amd import api, flate
    if level == 6:
        d = inputs["lost0"]
        w = io.BytesIO()
        with pytest.warns(api.ReferenceQ1StreamWarning):
            flate.compress(d, w, engine=eng)
        assert w.getvalue() == O.compress(d, 0, 6)
        w = io.BytesIO()
        flate.compress(d, w, api.Options(6, repair_q1=True), engine=eng)
        assert zlib.decompress(w.getvalue(), -15) == d


def test_long_chunk_takes_the_whole_stream_path():
    # levels 4..9: inputs of more than 65535 bytes are no longer refused (tests/test_gpu_stream.py)
    eng = engine()
    outs, st = eng.compress_many([bytes(65536), b"ok"], 0, 6)
    assert st == [0, 0]
    assert outs[0] == O.compress(bytes(65536), 0, 6)
    assert outs[1] == O.compress(b"ok", 0, 6)


def test_output_too_small_status():
    import ctypes as C
    eng = engine()
    from flate_amd import _capi
    data = np.frombuffer(os.urandom(5000), dtype=np.uint8)
    in_off = np.array([0, 5000], dtype=np.uint64)
    out_off = np.array([0, 100], dtype=np.uint64)
    out = np.zeros(128, dtype=np.uint8)
    out_len = np.zeros(1, dtype=np.uint64)
    status = np.zeros(1, dtype=np.int32)
    rc = _capi.lib().flate_hip_compress_batch(eng._h, data.ctypes.data, in_off.ctypes.data, 1, 0, 6, out.ctypes.data,
                                              out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, 0)
    assert rc == 0 and status[0] == 100 and out_len[0] == 0


def test_api_mirror_roundtrip():
    # flate.zig:356-481 "public interface": the same calls, through the Python mirror
    engine()
    from flate_amd import flate, gzip, zlib
    plain = b"Hello world\n"
    block = bytes([0x01, 0x0C, 0x00, 0xF3, 0xFF]) + plain
    gz = bytes([0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0x03]) + block + bytes([0xD5, 0xE0, 0x39, 0xB7, 0x0C, 0, 0, 0])
    zl = bytes([0x78, 0x9C]) + block + bytes([0x1C, 0xF2, 0x04, 0x47])
    for pkg, blob in ((gzip, gz), (zlib, zl), (flate, block)):
        w = io.BytesIO()
        pkg.decompress(io.BytesIO(blob), w)
        assert w.getvalue() == plain
        w = io.BytesIO()
        pkg.store.compress(io.BytesIO(plain), w)
        assert w.getvalue() == blob
        for comp in (lambda r, w_: pkg.compress(r, w_, pkg.Options()), pkg.huffman.compress, pkg.store.compress):
            c = io.BytesIO()
            comp(io.BytesIO(plain), c)
            w = io.BytesIO()
            pkg.decompress(io.BytesIO(c.getvalue()), w)
            assert w.getvalue() == plain
        c = io.BytesIO()
        cmp = pkg.compressor(c, pkg.Options(level=pkg.Level.best))
        cmp.write(plain[:5])
        cmp.write(plain[5:])
        cmp.finish()
        d = pkg.decompressor(io.BytesIO(c.getvalue()))
        assert d.reader().read() == plain


def test_decompressor_takes_from_the_reader_what_the_stream_needs():
    """inflate.zig:283-353: a decompressor reads its reader as it goes.  Three gzip members followed by a lot of
    unrelated bytes: every member decodes, reset() moves to the next one, and the reader is not drained."""
    from flate_amd import gzip, synth

    class CountingReader:
        def __init__(self, data):
            self.data, self.pos = data, 0

        def read(self, n=-1):
            if n is None or n < 0:
                n = len(self.data) - self.pos
            out = self.data[self.pos:self.pos + n]
            self.pos += len(out)
            return out

    parts = [synth.text(synth.SEED_TEXT + 40 + i, n).tobytes() for i, n in enumerate((300000, 10, 70000))]
    members = []
    for part in parts:
        c = io.BytesIO()
        gzip.compress(io.BytesIO(part), c, gzip.Options())
        members.append(c.getvalue())
    blob = b"".join(members) + bytes(8 << 20)
    rd = CountingReader(blob)
    d = gzip.decompressor(rd)
    got = []
    for i in range(3):
        pieces = []
        while True:
            buf = d.next()
            if buf is None:
                break
            assert len(buf) <= 65536
            pieces.append(buf)
        got.append(b"".join(pieces))
        if i < 2:
            assert d.more_input()
            d.reset()
    assert got == parts
    assert rd.pos < len(b"".join(members)) + (2 << 20)  # far from the 8 MiB that follow
    # a truncated stream still is EndOfStream once the reader has nothing more
    from flate_amd.api import EndOfStream
    with pytest.raises(EndOfStream):
        gzip.decompressor(CountingReader(members[0][:len(members[0]) // 2])).next()


def test_chunk_passes_are_split_and_mixed_with_streams():
    # FLATE_HIP_MAX_PASS_CHUNKS bounds one chunk-path pass; short and long inputs alternate, so the
    # call becomes many passes of both kinds: same bytes as in one go
    import os
    eng = engine()
    from flate_amd import synth
    text = synth.text(synth.SEED_TEXT + 3, 1 << 20).tobytes()
    rng = np.random.default_rng(8)
    datas = []
    for i in range(40):
        n = int(rng.integers(0, 65536)) if i % 5 else int(rng.integers(65536, 200000))
        o = int(rng.integers(0, len(text) - n))
        datas.append(text[o:o + n])
    a, st = eng.compress_many(datas, 1, 6)
    os.environ["FLATE_HIP_MAX_PASS_CHUNKS"] = "3"
    try:
        b, st2 = eng.compress_many(datas, 1, 6)
    finally:
        del os.environ["FLATE_HIP_MAX_PASS_CHUNKS"]
    assert st == [0] * 40 and st2 == [0] * 40 and a == b
    for d, got in zip(datas[:12], a):
        assert got == O.compress(d, 1, 6)


# ---- BASELINE.json configs[2]: gzip level 9 on the TAR-like workload ----
def test_config3_gzip_l9_tar_like_chunks():
    """Every 65535-byte chunk of 4 MiB of the TAR-like buffer, gzip level 9 == oracle."""
    from flate_amd import synth
    eng = engine()
    data = synth.tar_like(synth.SEED_TAR, 4 << 20).tobytes()
    chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
    outs, st = eng.compress_many(chunks, O.GZIP, 9)
    assert st == [0] * len(chunks)
    for i, (c, got) in enumerate(zip(chunks, outs)):
        assert got == O.compress(c, O.GZIP, 9), "chunk %d" % i
        assert pyzlib.decompress(got, 31) == c


def test_config3_gzip_l9_tar_like_one_stream():
    """The same 4 MiB + a ragged tail as ONE gzip level 9 stream (whole-stream path) == oracle, every byte."""
    from flate_amd import synth
    eng = engine()
    data = synth.tar_like(synth.SEED_TAR, (4 << 20) + 12345).tobytes()
    outs, st = eng.compress_many([data], O.GZIP, 9)
    assert st == [0]
    assert outs[0] == O.compress(data, O.GZIP, 9)
    assert pyzlib.decompress(outs[0], 31) == data


# ---- the reference's block-writer goldens through the DEVICE planner / bit packer ----
def _bw_cases():
    with open(os.path.join(GOLDEN, "block_writer_tokens.json")) as f:
        cases = json.load(f)
    for c in cases:
        c["tok"] = np.array([O.tok_lit(t[0]) if len(t) == 1 else O.tok_match(t[0], t[1])
                             for t in c["tokens"]], dtype=np.uint32)
    return cases


@pytest.mark.parametrize("fn", ["wb", "dyn"])
def test_block_writer_goldens_through_the_device_planner(fn):
    """block_writer.zig:599-706: token lists straight into BlockWriter.write (wb) / dynamicBlock (dyn),
    with and without the raw input, eof 0/1 -- 17 golden files each, run by k_plan / k_offsets / k_encode."""
    eng = engine()
    checked = 0
    for c in _bw_cases():
        variants = []
        if c["input"] and c["want"]:
            variants.append((golden("block_writer", c["input"]), c["want"]))
        variants.append((None, c["want_no_input"]))
        for inp, name in variants:
            want = golden("block_writer", name.replace("{s}", fn))
            got = eng.debug_write_block(c["tok"], inp, False, fn == "dyn")
            assert got == want, name
            got = bytearray(eng.debug_write_block(c["tok"], inp, True, fn == "dyn"))
            assert got[0] & 1 == 1
            got[0] &= 0xFE
            assert bytes(got) == want, name + " (eof)"
            checked += 1
    assert checked == 17


def test_device_planner_matches_oracle_on_random_token_blocks():
    eng = engine()
    rng = np.random.default_rng(12)
    for trial in range(24):
        n = int(rng.integers(0, 5000))
        kind = trial % 4
        if kind == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = rng.integers(97, 101, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = (b"the quick brown fox jumps over the lazy dog " * (n // 40 + 1))[:n]
        else:
            data = bytes(n)
        toks = O.tokenize(data, 6)
        for eof in (False, True):
            for fn in ("wb", "dyn"):
                for inp in (data, None):
                    assert eng.debug_write_block(toks, inp, eof, fn == "dyn") == O.block_write(fn, toks, eof, inp)


def test_sharded_entry_points_with_a_one_rank_rccl_communicator():
    """flate_hip_{compress,decompress}_batch_sharded with a real RCCL communicator of one rank (what a
    single-GPU box can run): local compress + pack + size all-gather; the slice holds the oracle's
    streams back to back.  The multi-rank exchange itself is exercised by the gloo protocol tests."""
    import ctypes as C
    import torch
    eng = engine()
    import torch.distributed  # loads the RCCL that torch ships
    from flate_amd import _capi, synth
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        data = synth.text(synth.SEED_TEXT + 3, 5 * 65535 + 99).tobytes()
        chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
        n = len(chunks)
        dev = torch.device("cuda", 0)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        in_off = torch.tensor(np.r_[0, np.cumsum([len(c) for c in chunks])], dtype=torch.int64, device=dev)
        caps = [(eng.compress_bound(len(c), O.GZIP, 6) + 7) & ~7 for c in chunks]
        out_off = torch.tensor(np.r_[0, np.cumsum(caps)], dtype=torch.int64, device=dev)
        out = torch.zeros(sum(caps) + 8, dtype=torch.uint8, device=dev)
        out_len = torch.zeros(n, dtype=torch.int64, device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        slice_bytes = (sum(caps) + 15) & ~15
        gathered = torch.zeros(slice_bytes, dtype=torch.uint8, device=dev)
        sizes = torch.zeros(1, dtype=torch.int64, device=dev)
        dst_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L = _capi.lib()
        rc = L.flate_hip_compress_batch_sharded(eng._h, comm, 0, 1, d_in.data_ptr(), in_off.data_ptr(), n, O.GZIP, 6,
                                                out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(),
                                                gathered.data_ptr(), slice_bytes, sizes.data_ptr(), dst_off.data_ptr())
        assert rc == 0, eng._L.flate_hip_last_error(eng._h)
        torch.cuda.synchronize()
        want = b"".join(O.compress(c, O.GZIP, 6) for c in chunks)
        assert int(sizes[0]) == len(want) and int(status.abs().sum()) == 0
        assert gathered[:len(want)].cpu().numpy().tobytes() == want
        # and back: the packed members inflate into the slice
        back = torch.zeros(len(data) + 16, dtype=torch.uint8, device=dev)
        dlen = torch.zeros(n, dtype=torch.int64, device=dev)
        rc = L.flate_hip_decompress_batch_sharded(eng._h, comm, 0, 1, gathered.data_ptr(), dst_off.data_ptr(), n, O.GZIP, 0,
                                                  back.data_ptr(), len(data) + 16, in_off.data_ptr(), dlen.data_ptr(),
                                                  status.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0 and back[:len(data)].cpu().numpy().tobytes() == data
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_planned_batches_only_enqueue():
    """flate_hip_plan_compress / flate_hip_compress_planned: four batches enqueued back to back with
    set_sync(0) return to the host while the GPU is still busy (event not reached, host time a small
    fraction of the GPU time), and every output equals the synchronous path's."""
    import time
    import torch
    from flate_amd import Engine, synth
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng.set_stream(stream.cuda_stream)
        n_bytes = 64 << 20
        datas = [synth.text_torch(synth.SEED_TEXT + 11 * k, n_bytes, device=dev) for k in range(4)]
        off = synth.split_offsets(n_bytes, 65535)
        n = len(off) - 1
        caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), O.GZIP, 6) + 7) & ~7 for i in range(n)], dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        outs = [torch.empty(int(out_off[-1]) + 8, dtype=torch.uint8, device=dev) for _ in range(4)]
        lens = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(4)]
        sts = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
        plan = eng.plan_compress(off, out_off, O.GZIP, 6)
        eng.set_sync(False)
        eng.compress_planned(plan, datas[0].data_ptr(), outs[0].data_ptr(), lens[0].data_ptr(), sts[0].data_ptr())  # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        t0 = time.perf_counter()
        for k in range(4):
            eng.compress_planned(plan, datas[k].data_ptr(), outs[k].data_ptr(), lens[k].data_ptr(), sts[k].data_ptr())
        host_s = time.perf_counter() - t0
        e1.record(stream)
        still_running = not e1.query()
        torch.cuda.synchronize()
        gpu_s = e0.elapsed_time(e1) * 1e-3
        assert still_running, "the host waited for the GPU"
        assert host_s < 0.25 * gpu_s, (host_s, gpu_s)
        # same bytes as the synchronous entry point / the oracle
        in_off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
        out_off_t = torch.from_numpy(out_off.astype(np.int64)).to(dev)
        ref_out = torch.empty_like(outs[0])
        ref_len = torch.zeros(n, dtype=torch.int64, device=dev)
        ref_st = torch.zeros(n, dtype=torch.int32, device=dev)
        eng.set_sync(True)
        for k in range(4):
            assert int(sts[k].abs().sum()) == 0
            eng.compress_device(datas[k].data_ptr(), in_off_t.data_ptr(), n, O.GZIP, 6, ref_out.data_ptr(),
                                out_off_t.data_ptr(), ref_len.data_ptr(), ref_st.data_ptr())
            assert torch.equal(ref_len, lens[k])
            for i in (0, n // 2, n - 1):
                a, l = int(out_off[i]), int(ref_len[i])
                assert torch.equal(ref_out[a:a + l], outs[k][a:a + l])
            i = 3 + k
            src = datas[k][int(off[i]):int(off[i + 1])].cpu().numpy().tobytes()
            got = outs[k][int(out_off[i]):int(out_off[i]) + int(lens[k][i])].cpu().numpy().tobytes()
            assert got == O.compress(src, O.GZIP, 6)
        eng.plan_destroy(plan)
        # a batch with an input beyond 65535 bytes at level 6 is not plannable
        with pytest.raises(Exception):
            eng.plan_compress(np.array([0, 70000], dtype=np.uint64), np.array([0, 80000], dtype=np.uint64), O.RAW, 6)
    eng.close()


@pytest.mark.parametrize("mode,container", [(6, 1), (9, 0), (4, 2)])
def test_planned_batches_of_several_passes_alternate_between_two_streams(mode, container, monkeypatch):
    """Round 6: a batch of more than one pass (here: FLATE_HIP_MAX_PASS_CHUNKS=5) runs its passes alternately on the caller's
    stream and a second one, each over its slice of a batch-wide workspace -- planned device batches and pinned host batches
    alike.  Same bytes as the oracle, pass after pass, call after call; FLATE_HIP_ONE_COMPUTE_STREAM=1 is round 5's way."""
    import torch
    from flate_amd import Engine, synth
    for one in (None, "1"):
        monkeypatch.setenv("FLATE_HIP_MAX_PASS_CHUNKS", "5")
        if one:
            monkeypatch.setenv("FLATE_HIP_ONE_COMPUTE_STREAM", one)
        else:
            monkeypatch.delenv("FLATE_HIP_ONE_COMPUTE_STREAM", raising=False)
        eng = Engine(0)
        dev = torch.device("cuda", 0)
        rng = np.random.default_rng(31)
        text = synth.text(synth.SEED_TEXT + 5, 23 * 40000).tobytes()
        chunks = [text[i * 40000:(i + 1) * 40000][:int(rng.integers(1, 40001))] for i in range(23)]
        chunks[7] = bytes(30000)                       # (a constant chunk, an empty one, noise: every tokenizer path)
        chunks[11] = b""
        chunks[13] = rng.integers(0, 256, 33000, dtype=np.uint8).tobytes()
        blob = b"".join(chunks)
        off = np.zeros(len(chunks) + 1, dtype=np.uint64)
        np.cumsum([len(c) for c in chunks], out=off[1:])
        caps = np.array([(eng.compress_bound(len(c), container, mode) + 7) & ~7 for c in chunks], dtype=np.uint64)
        out_off = np.zeros(len(chunks) + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        d_in = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
        plan = eng.plan_compress(off, out_off, container, mode)
        want = [O.compress(c, container, mode) for c in chunks]
        for rep in range(3):
            d_out = torch.full((int(out_off[-1]) + 8,), 0xA5, dtype=torch.uint8, device=dev)
            d_len = torch.zeros(len(chunks), dtype=torch.int64, device=dev)
            d_st = torch.full((len(chunks),), 77, dtype=torch.int32, device=dev)
            eng.compress_planned(plan, d_in.data_ptr(), d_out.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
            assert d_st.cpu().tolist() == [0] * len(chunks)
            out = d_out.cpu().numpy()
            for i, w in enumerate(want):
                a, l = int(out_off[i]), int(d_len[i])
                assert out[a:a + l].tobytes() == w, (one, rep, i)
        eng.plan_destroy(plan)
        # ... and host buffers (pageable here: pinned mirrors, sub-batches of 5 chunks)
        monkeypatch.setenv("FLATE_HIP_HOST_PASS_CHUNKS", "5")
        bigtext = synth.text(synth.SEED_TEXT + 9, 65535 * 160).tobytes()
        big = [bigtext[i * 65535:(i + 1) * 65535] for i in range(160)]
        outs, st = eng.compress_many(big, container, mode)
        assert st == [0] * len(big)
        for i in (0, 4, 5, 77, 159):
            assert outs[i] == O.compress(big[i], container, mode), (one, i)
        back, st2, _ = eng.decompress_many(outs, container, caps=[65536] * len(big))
        assert st2 == [0] * len(big) and back == big
        eng.close()


@pytest.mark.parametrize("rect", [None, "1"])
def test_pinned_output_slots_of_one_size_rectangle_copy_and_the_rest(rect, monkeypatch):
    """Pinned (and pageable: pinned mirrors) output with slots of one size: the produced bytes of every slot come home by the
    copy kernel (the default since round 5), or -- FLATE_HIP_RECT=1 -- the first half of every slot by the DMA engine's
    rectangle copy and what a chunk produced beyond it by the copy kernel.  Chunks that compress well, chunks that do
    not (the second half of the slot is needed), empty ones; three sub-batches; every stream against the oracle."""
    import torch
    from flate_amd import _capi, synth
    eng = engine()
    L = _capi.lib()
    if rect:
        monkeypatch.setenv("FLATE_HIP_RECT", rect)
    eng._sync_env()
    rng = np.random.default_rng(99)
    csz, n = 16384, 2500
    text = synth.text(synth.SEED_TEXT + 9, n * csz)
    data = text.copy()
    kinds = rng.integers(0, 4, n)
    for i in range(n):
        if kinds[i] == 1:
            data[i * csz:(i + 1) * csz] = rng.integers(0, 256, csz, dtype=np.uint8)      # incompressible
        elif kinds[i] == 2:
            data[i * csz:(i + 1) * csz] = 0
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(csz)).astype(np.uint64)
    for container, mode in ((O.RAW, 6), (O.GZIP, 1)):
        cap = (eng.compress_bound(csz, container, mode) + 7) & ~7
        out_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(cap)).astype(np.uint64)
        res = []
        for pinned in (True, False):
            mk = (lambda k: torch.full((k,), 0xAA, dtype=torch.uint8).pin_memory()) if pinned else (lambda k: torch.full((k,), 0xAA, dtype=torch.uint8))
            h_in = mk(len(data) + 8)
            h_in[: len(data)] = torch.from_numpy(data)
            h_out = mk(int(out_off[-1]) + 8)
            out_len = np.zeros(n, dtype=np.uint64)
            status = np.zeros(n, dtype=np.int32)
            rc = L.flate_hip_compress_batch(eng._h, h_in.data_ptr(), off.ctypes.data, n, container, mode, h_out.data_ptr(),
                                            out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
            assert rc == 0 and not status.any()
            o = h_out.numpy()
            res.append([o[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)])
            assert int(o[int(out_off[-1]):].min()) == 0xAA  # nothing behind the last slot
            for i in range(0, n, 131):  # beyond out_len: zeros or the caller's bytes
                rest = o[int(out_off[i]) + int(out_len[i]): int(out_off[i + 1])]
                assert set(np.unique(rest).tolist()) <= {0x00, 0xAA}
        assert res[0] == res[1]
        assert max(len(x) for x in res[0]) > cap // 2 > min(len(x) for x in res[0])
        for i in list(range(0, n, 61)) + [n - 1]:
            assert res[0][i] == O.compress(data[i * csz:(i + 1) * csz].tobytes(), container, mode), i


def test_pinned_host_buffers_take_the_overlapped_path():
    """MEM_HOST with pinned buffers runs in sub-batches (H2D / kernels / D2H on three streams): every stream
    equals what the pageable call produces, chunk sizes ragged, more chunks than one sub-batch holds."""
    import torch
    from flate_amd import _capi, synth
    eng = engine()
    L = _capi.lib()
    rng = np.random.default_rng(77)
    text = synth.text(synth.SEED_TEXT + 3, 40 << 20)
    sizes = rng.integers(0, 20000, 2600)
    sizes[::97] = 65535
    sizes[5] = 0
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    np.cumsum(sizes, out=off[1:].view(np.int64))
    n = len(sizes)
    data = text[: int(off[-1])]
    for container, mode in ((O.GZIP, 6), (O.RAW, 1), (O.ZLIB, 4)):
        caps = np.array([(eng.compress_bound(int(x), container, mode) + 7) & ~7 for x in sizes], dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        res = []
        for pinned in (False, True):
            mk = (lambda k, dt: torch.zeros(k, dtype=dt).pin_memory()) if pinned else (lambda k, dt: torch.zeros(k, dtype=dt))
            h_in = mk(len(data) + 8, torch.uint8)
            h_in[: len(data)] = torch.from_numpy(data)
            h_out = mk(int(out_off[-1]) + 8, torch.uint8)
            out_len = np.zeros(n, dtype=np.uint64)
            status = np.zeros(n, dtype=np.int32)
            rc = L.flate_hip_compress_batch(eng._h, h_in.data_ptr(), off.ctypes.data, n, container, mode, h_out.data_ptr(),
                                            out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
            assert rc == 0 and not status.any()
            o = h_out.numpy()
            res.append([o[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)])
        assert res[0] == res[1]
        for i in (0, 5, 97, n - 1):
            assert res[1][i] == O.compress(data[int(off[i]): int(off[i + 1])].tobytes(), container, mode)
        # and back: the streams packed back to back in pinned memory, outputs into pinned slots of the original sizes
        lens = np.array([len(x) for x in res[1]], dtype=np.int64)
        c_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lens, out=c_off[1:].view(np.int64))
        p_in = torch.zeros(int(c_off[-1]) + 8, dtype=torch.uint8).pin_memory()
        p_in[: int(c_off[-1])] = torch.from_numpy(np.frombuffer(b"".join(res[1]), dtype=np.uint8).copy())
        p_out = torch.zeros(len(data) + 8, dtype=torch.uint8).pin_memory()
        dlen = np.zeros(n, dtype=np.uint64)
        status = np.ones(n, dtype=np.int32)
        cons = np.zeros(n, dtype=np.uint64)
        os.environ["FLATE_HIP_HOST_PASS_CHUNKS"] = "256"  # (inflate overlaps in sub-batches of 4 x this many streams)
        eng._sync_env()  # (the library reads its knobs once per handle: this call goes past the Engine's methods)
        try:
            rc = L.flate_hip_decompress_batch(eng._h, p_in.data_ptr(), c_off.ctypes.data, n, container, 0,
                                              p_out.data_ptr(), off.ctypes.data, dlen.ctypes.data, status.ctypes.data,
                                              cons.ctypes.data, _capi.MEM_HOST)
        finally:
            del os.environ["FLATE_HIP_HOST_PASS_CHUNKS"]
            eng._sync_env()
        assert rc == 0 and not status.any()
        assert np.array_equal(dlen.astype(np.int64), sizes) and np.array_equal(cons.astype(np.int64), lens)
        assert np.array_equal(p_out.numpy()[: len(data)], data)


@pytest.mark.parametrize("mode", [6, 9])
def test_pinned_batches_that_mix_long_and_short_inputs(mode):
    """on the pinned path the passes are enqueued without a host wait in between, and a whole-stream
    pass (an input longer than 65535 bytes) keeps its block table at the START of the table buffer, where the chunk
    passes' slices lie.  A 200 KB stream has 7 blocks, a 1 MiB stream 33: the next chunk pass's slice used to be copied
    over them while the stream pass's kernels were only enqueued.  Batches like [long, short, short, ..., long, short ...]
    from pinned buffers and from pageable ones of more than 8 MiB (the pinned mirrors): every stream == the oracle's."""
    import torch
    from flate_amd import _capi, synth
    eng = engine()
    L = _capi.lib()
    rng = np.random.default_rng(1234 + mode)
    text = synth.text(synth.SEED_TEXT + 9, 24 << 20)
    sizes = [200_000] + [int(x) for x in rng.integers(1, 3000, 300)] + [1 << 20] + [int(x) for x in rng.integers(1000, 65535, 40)] + \
            [70_000] + [int(x) for x in rng.integers(1, 500, 200)] + [300_000, 5, 65535, 65536]
    if mode == 6:  # (level 6 also as a pageable batch of more than 8 MiB: the pinned mirrors)
        sizes = sizes + [65535] * 90 + [150_000] + [int(x) for x in rng.integers(1, 9000, 150)]
    sizes = np.array(sizes, dtype=np.int64)
    n = len(sizes)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(sizes, out=off[1:].view(np.int64))
    assert mode != 6 or int(off[-1]) > (8 << 20)
    pad = 1 if mode == 6 else 0
    data = text[: int(off[-1])]
    caps = np.array([(eng.compress_bound(int(x), O.GZIP, mode) + 7) & ~7 for x in sizes], dtype=np.uint64)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    want = [O.compress(data[int(off[i]): int(off[i + 1])].tobytes(), O.GZIP, mode) for i in range(n)]
    os.environ["FLATE_HIP_HOST_PASS_CHUNKS"] = "128"  # (several chunk passes between and behind the streams)
    eng._sync_env()
    try:
        for pinned in ((True, False) if pad else (True,)):
            mk = (lambda k: torch.zeros(k, dtype=torch.uint8).pin_memory()) if pinned else (lambda k: torch.zeros(k, dtype=torch.uint8))
            h_in = mk(len(data) + 8)
            h_in[: len(data)] = torch.from_numpy(data)
            h_out = mk(int(out_off[-1]) + 8)
            for rep in range(2):  # (twice: the second call finds the buffers of the first)
                out_len = np.zeros(n, dtype=np.uint64)
                status = np.ones(n, dtype=np.int32)
                rc = L.flate_hip_compress_batch(eng._h, h_in.data_ptr(), off.ctypes.data, n, O.GZIP, mode, h_out.data_ptr(),
                                                out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
                assert rc == 0 and not status.any()
                o = h_out.numpy()
                for i in range(n):
                    got = o[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes()
                    assert got == want[i], (pinned, rep, i, int(sizes[i]))
    finally:
        del os.environ["FLATE_HIP_HOST_PASS_CHUNKS"]
        eng._sync_env()


def test_runny_windows_take_their_variant_and_match_the_oracle():
    """Windows with runs, padding, short periods and repeated records (k_lz_sort marks them, k_lz_match<.., RJ>
    filters groups of candidates on one byte and serves lone lanes with the whole wave): tokens == oracle at
    every level, chunk path and whole-stream path."""
    from flate_amd import synth
    eng = engine()
    rng = np.random.default_rng(4321)

    def sparse(n, every, lo=0, hi=256):
        z = np.zeros(n, dtype=np.uint8)
        k = max(1, n // every)
        z[rng.integers(0, n, k)] = rng.integers(lo, hi, k, dtype=np.uint8)
        return z.tobytes()

    def records(n):
        rec = bytearray(rng.integers(0, 256, 48, dtype=np.uint8).tobytes())
        out = bytearray()
        i = 0
        while len(out) < n:
            r = bytearray(rec)
            r[0:4] = int(i).to_bytes(4, "little")
            r[20 + int(rng.integers(0, 8))] = int(rng.integers(0, 256))
            out += r
            i += 1
        return bytes(out[:n])

    text = synth.text(synth.SEED_TEXT + 17, 70000).tobytes()
    cases = {
        "sparse97": sparse(65535, 97), "sparse9": sparse(50000, 9), "sparse400": sparse(65535, 400),
        "sparse_two_values": sparse(65535, 30, 1, 3),
        "ab": (b"ab" * 33000)[:65535], "abc": (b"abc" * 22000)[:65535], "period5": (b"hello" * 13200)[:65535],
        "period255": (bytes(range(255)) * 300)[:65535], "period300": (rng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 230)[:65535],
        "records48": records(65535),
        "runs_of_runs": b"".join(bytes([int(v)]) * int(l) for v, l in zip(rng.integers(0, 4, 3000), rng.integers(1, 70, 3000)))[:65535],
        "text_then_zeros_then_text": text[:20000] + bytes(25000) + text[20000:40535],
        "tar_like": synth.tar_like(synth.SEED_TAR, 1 << 20)[300000:365535].tobytes(),
    }
    names = list(cases)
    for level in (4, 5, 6, 7, 8, 9):
        outs, st = eng.compress_many([cases[n] for n in names], O.RAW, level)
        assert st == [0] * len(names)
        for i, n in enumerate(names):
            want = O.tokenize(cases[n], level)
            got = eng.debug_tokens(i)
            assert len(got) == len(want) and not (got != want).any(), (n, level)
            assert outs[i] == O.compress(cases[n], O.RAW, level), (n, level)
    # the same kinds of data as ONE stream each (tiles of the whole-stream path)
    long_cases = [sparse(300000, 97), (b"abcdefg" * 40000), records(200000), bytes(70000) + sparse(150000, 50) + text]
    for level in (4, 6, 9):
        outs, st = eng.compress_many(long_cases, O.GZIP, level)
        assert st == [0] * len(long_cases)
        for d, o in zip(long_cases, outs):
            assert o == O.compress(d, O.GZIP, level), level


def test_batches_of_many_blocks_take_the_wave_per_block_encoder(monkeypatch):
    """From 8192 plan slots up (4096 chunks of the chunk path in one pass) the bit packer runs one wave per block
    (k_encode_wave): every chunk of such a batch against the oracle -- text, incompressible (stored blocks), empty,
    one byte, all-zero, and a few of full size whose first block holds 32768 tokens.  (Host buffers staged in one
    piece: the pinned mirrors would cut the batch into sub-batches of 1024 chunks.)"""
    from flate_amd import synth
    monkeypatch.setenv("FLATE_HIP_NO_PIN_MIRROR", "1")
    eng = engine()
    rng = np.random.default_rng(31337)
    text = synth.text(synth.SEED_TEXT + 77, 12 << 20).tobytes()
    chunks, pos = [], 0
    for i in range(4300):
        kind = i % 43
        if kind == 0:
            c = b""
        elif kind == 1:
            c = b"x"
        elif kind == 2:
            c = rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes()
        elif kind == 3:
            c = bytes(int(rng.integers(1, 9000)))
        elif kind == 4 and i < 400:
            c = text[pos:pos + 65535]
        else:
            c = text[pos:pos + int(rng.integers(20, 4000))]
        pos = (pos + len(c)) % (len(text) - 70000)
        chunks.append(c)
    for container, mode in ((0, 6), (1, 4)):
        outs, st = eng.compress_many(chunks, container, mode)
        assert st == [0] * len(chunks)
        for i, (c, got) in enumerate(zip(chunks, outs)):
            assert got == O.compress(c, container, mode), (i, len(c), container, mode)


def test_pageable_host_buffers_go_through_pinned_mirrors(monkeypatch):
    """Pageable host buffers of 8 MiB or more (numpy arrays: what compress_many hands over) are copied into pinned
    mirrors by host threads and compressed in overlapped sub-batches; the bytes are those of the one-piece staging
    (FLATE_HIP_NO_PIN_MIRROR) and of the oracle."""
    from flate_amd import synth
    eng = engine()
    data = synth.text(synth.SEED_TEXT + 55, 20 * 1024 * 1024 + 333).tobytes()
    chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
    chunks[7] = b""
    chunks[11] = np.random.default_rng(3).integers(0, 256, 40000, dtype=np.uint8).tobytes()
    outs, st = eng.compress_many(chunks, 1, 6)
    assert st == [0] * len(chunks)
    monkeypatch.setenv("FLATE_HIP_NO_PIN_MIRROR", "1")
    outs2, st2 = eng.compress_many(chunks, 1, 6)
    assert st2 == st and outs2 == outs
    for i in (0, 7, 11, 100, len(chunks) - 1):
        assert outs[i] == O.compress(chunks[i], 1, 6), i
    back, st3, _ = eng.decompress_many(outs, 1, caps=[65536] * len(chunks))
    assert st3 == [0] * len(chunks) and back == chunks


def test_link_kernels_fallback_path_matches_oracle():
    # k_lz_links relies on the LDS serving a wave's exchanges in lane / program order and checks it: a lane that was
    # overtaken sends the chunk to a one-position-at-a-time path, which gfx950 has never taken.  The build with
    # -DFL_CHAIN_FORCE_SLOW always takes it; the token lists must be the oracle's all the same (its own process:
    # the library is chosen at import).
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "flate_amd", "lib", "var", "libflate_hip_slowchain.so")
    assert os.path.exists(lib), "build() makes it (`make variants` in flate_amd/csrc)"
    env = dict(os.environ, FLATE_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__) +
                        "::test_tokenizer_matches_oracle_tokens"], env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_sparse_chain_kernels_at_every_level():
    # Levels 8 and 9 take k_lz_links / k_lz_walk (the match finder over sparser chains, runs of one byte in one step,
    # offset mode); the build with -DFL_BULK_MIN_CHAIN=1 sends levels 4-7 through them as well: same token lists, same
    # bytes as the oracle (its own process: the library is chosen at import).
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "flate_amd", "lib", "var", "libflate_hip_walkall.so")
    assert os.path.exists(lib), "build() makes it (`make variants` in flate_amd/csrc)"
    env = dict(os.environ, FLATE_HIP_LIB=lib)
    me = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", me + "::test_tokenizer_matches_oracle_tokens",
                        me + "::test_bytes_match_oracle", me + "::test_runny_inputs_match_oracle"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_runny_inputs_match_oracle():
    # runs of one byte of every length, junk between them, repeated blocks, sparse zeros: what the run logic of
    # k_lz_walk (levels 8, 9) skips must be exactly what cannot change the result
    eng = engine()
    rng = np.random.default_rng(77)

    def mk(seed, n, maxrun, alphabet, junkmax):
        r = np.random.default_rng(seed)
        parts, k = [], 0
        while k < n:
            a = bytes([int(r.integers(0, alphabet))]) * int(r.integers(1, maxrun))
            b = r.integers(0, alphabet + 2, int(r.integers(0, junkmax)), dtype=np.uint8).tobytes()
            parts += [a, b]
            k += len(a) + len(b)
        return b"".join(parts)[:n]

    blk = mk(4, 3000, 100, 2, 6)
    sparse = np.zeros(65535, dtype=np.uint8)
    sparse[rng.integers(0, 65535, 700)] = rng.integers(0, 256, 700, dtype=np.uint8)
    datas = [mk(1, 65535, 40, 2, 4), mk(2, 65535, 700, 1, 3), mk(3, 60000, 300, 3, 12), (blk * 30)[:65535], mk(5, 65535, 1200, 1, 2),
             sparse.tobytes(), b"x" + bytes(65534), bytes(30000) + b"abcdefgh" * 100 + bytes(30000), mk(6, 777, 50, 2, 3)]
    for level in (4, 6, 8, 9):
        outs, st = eng.compress_many(datas, O.RAW, level)
        assert st == [0] * len(datas)
        for i, d in enumerate(datas):
            assert outs[i] == O.compress(d, O.RAW, level), (i, level)


def test_inputs_on_the_match_finders_thresholds():
    """tests/_adversarial.py: candidates exactly at, before and behind the chain budget and its quarter (deflate.zig:241-245), matches
    of exactly good / lazy / nice bytes and one less / more, longer ones behind them, better ones at the next positions -- what
    random data does not hold.  As chunks and as whole streams (the call in a window's interior, at its edge, in the next window):
    bytes == oracle at every level.  tools/threshold_sweep.py runs hundreds of these through every path."""
    import _adversarial as A
    rng = np.random.default_rng(2024)
    eng = engine()
    for i in range(6):
        for level in (4, 5, 6, 7, 8, 9):
            d = A.threshold_input(rng, level, int(rng.choice([20000, 50000, 65535])))[-65535:]
            s1 = A.junk(rng, int(rng.integers(0, 400))) + d + A.junk(rng, int(rng.integers(70000, 90000)))
            edge = 65274 + int(rng.integers(-300, 20)) - (len(d) - 200)
            s2 = (A.junk(rng, max(0, edge)) + d + A.junk(rng, 80000)) if edge > 0 else s1
            datas = [d, s1, s2, A.junk(rng, 32768 + int(rng.integers(0, 3000))) + s2]
            c = int(rng.integers(0, 3))
            outs, st = eng.compress_many(datas, c, level)
            assert st == [0] * 4, (i, level, st)
            for j, (x, o) in enumerate(zip(datas, outs)):
                assert o == O.compress(x, c, level), (i, level, j, len(x))


def test_symbol_frequencies_that_need_the_length_limits():
    """tests/_adversarial.py: literal, match-length / distance and code-length frequencies that grow like Fibonacci numbers or powers
    of two: Huffman trees deeper than 15 (7) bits, which the builder has to flatten the way the reference does
    (huffman_encoder.zig).  Modes 1 and 4-9, chunks and streams: bytes == oracle, and back through the GPU inflater."""
    import _adversarial as A
    rng = np.random.default_rng(77)
    eng = engine()
    for i in range(8):
        growth = float(rng.choice([1.618, 2.0, 1.5, 3.0]))
        nsym = int(rng.integers(12, 60))
        n = int(rng.choice([3000, 20000, 65535, 150000]))
        datas = [A.skewed(rng, n, growth, nsym), A.match_skew(rng, min(n, 100000), float(rng.choice([1.2, 1.618, 2.0]))),
                 A.skewed(rng, n // 2, growth, nsym) + A.match_skew(rng, min(n, 60000) // 2 + 4100, 1.618)]
        for mode in (1, 4, 6, 9):
            c = int(rng.integers(0, 3))
            outs, st = eng.compress_many(datas, c, mode)
            for x, o, s in zip(datas, outs, st):
                assert s in (0, 102) and o == O.compress(x, c, mode), (i, mode, len(x), growth, nsym, s)
            back, st2, _ = eng.decompress_many(outs, c, caps=[len(x) + 8 for x in datas])
            for x, o, b, s in zip(datas, outs, back, st2):
                w = O.decompress(o, c, 0, cap=(len(x) + 8 + 7) & ~7)
                assert O.STATUS[s] == w[0] and (w[0] != "Ok" or b == w[1]), (i, mode, len(x))
