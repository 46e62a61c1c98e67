"""Pins the CPU oracle's INFLATE side against the reference's own tests
 and against the reference's differential inflater
bin/puff/puff.c (built into oracle/_ref).  No GPU."""
import os
import random
import zlib as pyzlib

import pytest

import _oracle as O
from conftest import golden

HELLO = b"Hello world\n"
ABCD = b"ABCDEABCD ABCDEABCD"
STORED = bytes([0x01, 0x0C, 0x00, 0xF3, 0xFF]) + HELLO
FIXED = bytes([0xF3, 0x48, 0xCD, 0xC9, 0xC9, 0x57, 0x28, 0xCF, 0x2F, 0xCA, 0x49, 0xE1, 0x02, 0x00])
DYN = bytes([0x3D, 0xC6, 0x39, 0x11, 0x00, 0x00, 0x0C, 0x02, 0x30, 0x2B, 0xB5, 0x52, 0x1E, 0xFF, 0x96, 0x38,
             0x16, 0x96, 0x5C, 0x1E, 0x94, 0xCB, 0x6D, 0x01])
GZ_HDR = bytes([0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0x03])


# inflate.zig:357-393
def test_raw_vectors():
    assert O.decompress(STORED, O.RAW)[:2] == ("Ok", HELLO)
    assert O.decompress(FIXED, O.RAW)[:2] == ("Ok", HELLO)
    assert O.decompress(DYN, O.RAW)[:2] == ("Ok", ABCD)


# inflate.zig:395-452
def test_gzip_vectors():
    foot_hello = bytes([0xD5, 0xE0, 0x39, 0xB7, 0x0C, 0, 0, 0])
    assert O.decompress(GZ_HDR + STORED + foot_hello, O.GZIP)[:2] == ("Ok", HELLO)
    hdr4 = bytes([0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0x04, 0x03])
    assert O.decompress(hdr4 + FIXED + foot_hello, O.GZIP)[:2] == ("Ok", HELLO)
    assert O.decompress(GZ_HDR + DYN + bytes([0x17, 0x1C, 0x39, 0xB4, 0x13, 0, 0, 0]), O.GZIP)[:2] == ("Ok", ABCD)
    named = bytes([0x1F, 0x8B, 0x08, 0x08, 0xE5, 0x70, 0xB1, 0x65, 0x00, 0x03, 0x68, 0x65, 0x6C, 0x6C, 0x6F, 0x2E,
                   0x74, 0x78, 0x74, 0x00]) + FIXED + foot_hello
    assert O.decompress(named, O.GZIP)[:2] == ("Ok", HELLO)


# inflate.zig:454-479
def test_zlib_vector():
    blob = bytes([0x78, 0x9C]) + STORED + bytes([0x1C, 0xF2, 0x04, 0x47])
    assert O.decompress(blob, O.ZLIB)[:2] == ("Ok", HELLO)


# inflate.zig:481-542 -- the 40-case error taxonomy
FUZZ = [
    ("deflate-stream", None, "@deflate-stream.expect"),
    ("empty-distance-alphabet01", None, b""),
    ("empty-distance-alphabet02", None, b""),
    ("end-of-stream", "EndOfStream", None),
    ("invalid-distance", "InvalidMatch", None),
    ("invalid-tree01", "IncompleteHuffmanTree", None),
    ("invalid-tree02", "IncompleteHuffmanTree", None),
    ("invalid-tree03", "IncompleteHuffmanTree", None),
    ("lengths-overflow", "InvalidDynamicBlockHeader", None),
    ("out-of-codes", "InvalidCode", None),
    ("puff01", "WrongStoredBlockNlen", None),
    ("puff02", "EndOfStream", None),
    ("puff03", None, bytes([0x0A])),
    ("puff04", "InvalidCode", None),
    ("puff05", "EndOfStream", None),
    ("puff06", "EndOfStream", None),
    ("puff08", "InvalidCode", None),
    ("puff09", None, b"P"),
    ("puff10", "InvalidCode", None),
    ("puff11", "InvalidMatch", None),
    ("puff12", "InvalidDynamicBlockHeader", None),
    ("puff13", "IncompleteHuffmanTree", None),
    ("puff14", "EndOfStream", None),
    ("puff15", "IncompleteHuffmanTree", None),
    ("puff16", "InvalidDynamicBlockHeader", None),
    ("puff17", "InvalidDynamicBlockHeader", None),
    ("fuzz1", "InvalidDynamicBlockHeader", None),
    ("fuzz2", "InvalidDynamicBlockHeader", None),
    ("fuzz3", "InvalidMatch", None),
    ("fuzz4", "OversubscribedHuffmanTree", None),
    ("puff18", "OversubscribedHuffmanTree", None),
    ("puff19", "OversubscribedHuffmanTree", None),
    ("puff20", "OversubscribedHuffmanTree", None),
    ("puff21", "OversubscribedHuffmanTree", None),
    ("puff22", "OversubscribedHuffmanTree", None),
    ("puff23", "InvalidDynamicBlockHeader", None),
    ("puff24", "InvalidDynamicBlockHeader", None),
    ("puff25", "OversubscribedHuffmanTree", None),
    ("puff26", "InvalidDynamicBlockHeader", None),
    ("puff27", "InvalidDynamicBlockHeader", None),
]


@pytest.mark.parametrize("flags", [0, 1], ids=["rfc", "reference-strict-q6"])
@pytest.mark.parametrize("name,err,out", FUZZ, ids=[f[0] for f in FUZZ])
def test_fuzz_corpus_error_taxonomy(name, err, out, flags):
    data = golden("fuzz", name + ".input")
    st, got, _ = O.decompress(data, O.RAW, flags=flags)
    if err is not None:
        assert st == err
    else:
        assert st == "Ok"
        want = golden("fuzz", out[1:]) if isinstance(out, str) else out
        assert got == want


def test_fuzz_corpus_agrees_with_puff_on_success_vs_error():
    # bin/fuzz_puff.zig:42-50: both must agree on error-vs-success and on bytes
    if not O.puff_available():
        pytest.skip("oracle/_ref/libpuff.so not built")
    for name, err, out in FUZZ:
        data = golden("fuzz", name + ".input")
        st, got, _ = O.decompress(data, O.RAW)
        rc, pout = O.puff(data)
        assert (st == "Ok") == (rc == 0), name
        if rc == 0:
            assert got == pout, name


# inflate.zig:544-563 "bug 18967": two concatenated zlib streams
def test_two_concatenated_zlib_streams():
    a, b = golden("fuzz", "first.input"), golden("fuzz", "second.input")
    st, o1, used = O.decompress(a + b, O.ZLIB)
    assert st == "Ok" and used == len(a)
    st, o2, used2 = O.decompress((a + b)[used:], O.ZLIB)
    assert st == "Ok" and used2 == len(b)
    assert o1 + o2 == golden("fuzz", "first.expect") + golden("fuzz", "second.expect")


# flate.zig:255-265
def test_dont_read_past_stream_end():
    blob = bytes([0x08, 0xD7, 0x63, 0xF8, 0xCF, 0xC0, 0xC0, 0x00, 0xC1, 0xFF, 0xFF, 0x43, 0x30, 0x03, 0x03, 0xC3,
                  0xFF, 0xFF, 0xFF, 0x01, 0x83, 0x95, 0x0B, 0xF5])
    want = bytes([0x00, 0xFF, 0x00, 0x00, 0x00, 0xFF, 0x00, 0x00, 0x00, 0xFF, 0x00, 0xFF, 0xFF, 0xFF, 0x00, 0xFF,
                  0xFF, 0xFF, 0x00, 0x00, 0x00, 0x00, 0xFF, 0xFF, 0xFF])
    assert O.decompress(blob, O.ZLIB)[:2] == ("Ok", want)


# flate.zig:267-295
def test_zlib_header_errors():
    assert O.decompress(bytes([0x78]), O.ZLIB)[0] == "EndOfStream"
    assert O.decompress(bytes([0x79, 0x94]), O.ZLIB)[0] == "BadZlibHeader"
    assert O.decompress(bytes([0x88, 0x98]), O.ZLIB)[0] == "BadZlibHeader"
    assert O.decompress(bytes([0x78, 0xDA, 0x03, 0x00, 0x00, 0x00, 0x00, 0x00]), O.ZLIB)[0] == "WrongZlibChecksum"
    assert O.decompress(bytes([0x78, 0xDA, 0x03, 0x00, 0x00]), O.ZLIB)[0] == "EndOfStream"


# flate.zig:297-354
def test_gzip_header_errors():
    assert O.decompress(bytes([0x1F, 0x8B]), O.GZIP)[0] == "EndOfStream"
    assert O.decompress(bytes([0x1F, 0x8B, 0x09, 0, 0, 0, 0, 0, 0, 0x03]), O.GZIP)[0] == "BadGzipHeader"
    assert O.decompress(GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0x01, 0, 0, 0, 0]), O.GZIP)[0] == "WrongGzipChecksum"
    assert O.decompress(GZ_HDR + bytes([0x03, 0, 0, 0, 0]), O.GZIP)[0] == "EndOfStream"
    assert O.decompress(GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0x01]), O.GZIP)[0] == "WrongGzipSize"
    assert O.decompress(GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0, 0, 0, 0]), O.GZIP)[0] == "EndOfStream"
    blob = bytes([0x1F, 0x8B, 0x08, 0x12, 0x00, 0x09, 0x6E, 0x88, 0x00, 0xFF, 0x48, 0x65, 0x6C, 0x6C, 0x6F, 0x00,
                  0x99, 0xD6, 0x01, 0x00, 0x00, 0xFF, 0xFF, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00])
    assert O.decompress(blob, O.GZIP)[:2] == ("Ok", b"")


def test_invalid_block_type():
    assert O.decompress(bytes([0x07, 0x00]), O.RAW)[0] == "InvalidBlockType"  # inflate.zig:267


# quirk Q6: a code-length repeat crossing the HLIT/HDIST
# boundary is legal (RFC 1951 3.2.7), is emitted by the reference's own
# encoder, and is rejected by the reference's inflater.
def test_q6_cross_boundary_repeat(rfc1951):
    data = rfc1951[20395:20395 + 1662]
    c = O.compress(data, O.RAW, 6)
    assert pyzlib.decompress(c, -15) == data
    assert O.decompress(c, O.RAW, flags=0)[:2] == ("Ok", data)
    assert O.decompress(c, O.RAW, flags=1)[0] == "InvalidDynamicBlockHeader"
    if O.puff_available():
        rc, out = O.puff(c)
        assert rc == 0 and out == data


def test_differential_vs_puff_and_zlib_random_streams():
    # all-levels round trip (bin/roundtrip.zig:14-33) + differential inflate
    rng = random.Random(1234)
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 9))) for _ in range(200)]
    for trial in range(12):
        n = rng.choice([0, 1, 5, 100, 3000, 70000, 140000])
        kind = trial % 3
        if kind == 0:
            data = b" ".join(rng.choice(words) for _ in range(n // 5 + 1))[:n]
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        else:
            data = bytes(rng.choice(b"ab\0\0\0\0") for _ in range(n))
        for mode in (O.STORE, O.HUFFMAN, 4, 6, 9):
            for cont in (O.RAW, O.GZIP, O.ZLIB):
                c = O.compress(data, cont, mode)
                st, out, used = O.decompress(c, cont, cap=len(data) + 16)
                assert st == "Ok" and out == data and used == len(c)
                assert pyzlib.decompress(c, {0: -15, 1: 31, 2: 15}[cont]) == data
            if O.puff_available():
                rc, pout = O.puff(O.compress(data, O.RAW, mode), cap=len(data) + 16)
                assert rc == 0 and pout == data
        # zlib-produced streams decode identically
        for lvl in (1, 6, 9):
            co = pyzlib.compressobj(lvl, pyzlib.DEFLATED, -15)
            c = co.compress(data) + co.flush()
            assert O.decompress(c, O.RAW, cap=len(data) + 16)[:2] == ("Ok", data)


def test_directed_edge_streams_agree_with_puff():
    """tests/_inflate_edge_cases.py (what the GPU decoders are checked on under -m gpu): the oracle's success / failure and bytes ==
    the reference's own differential inflater (bin/puff/puff.c, oracle/_ref)."""
    import pytest
    from _inflate_edge_cases import CASES
    if not O.puff_available():
        pytest.skip("oracle/_ref/libpuff.so is built from /root/reference, which is not here")
    for n, s in sorted(CASES.items()):
        name, out, _ = O.decompress(s, O.RAW, 0, cap=80000)
        rc, pout = O.puff(s, cap=80000)
        assert (name == "Ok") == (rc == 0), (n, name, rc)
        if rc == 0:
            assert pout == out, n
