"""Pins the CPU oracle's COMPRESS side against every result the reference's own
tests hold for the path.  No GPU."""
import json
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN, golden

LEVELS = [4, 5, 6, 7, 8, 9]


def L(c):
    return O.tok_lit(c if isinstance(c, int) else ord(c))


M = O.tok_match


# ---- (1) exact token lists, deflate.zig:533-575 ----
@pytest.mark.parametrize("container", [O.RAW, O.GZIP, O.ZLIB])
def test_tokenization_known_answers(container):
    cases = [
        (b"Blah blah blah blah blah!",
         [L("B"), L("l"), L("a"), L("h"), L(" "), L("b"), M(5, 18), L("!")]),
        (b"ABCDEABCD ABCDEABCD",
         [L("A"), L("B"), L("C"), L("D"), L("E"), L("A"), L("B"), L("C"), L("D"), L(" "),
          L("A"), M(10, 8)]),
    ]
    header = {O.RAW: 0, O.GZIP: 10, O.ZLIB: 2}[container]
    total = {O.RAW: 0, O.GZIP: 18, O.ZLIB: 6}[container]
    for data, want in cases:
        d = O.Deflate(container, 6, log_tokens=True)
        d.write(data)
        d.flush()
        assert list(d.tokens()) == want
        d.close()
    # header/footer byte counts (deflate.zig:570-572), with a token-discarding
    # writer the container bytes are all that is written: here we check the
    # header is what precedes the first block and the footer size.
    d = O.Deflate(container, 6)
    assert len(d.output()) == header
    d.finish()
    # empty input: fixed empty final block `03 00` (Q5)
    assert len(d.output()) == total + 2
    d.close()


# ---- (2) token counts per level, deflate.zig:610-680 ----
TOKEN_COUNTS = [
    (("rfc1951.txt",), [7675, 7672, 7599, 7594, 7598, 7599]),
    (("block_writer", "huffman-null-max.input"), [257] * 6),
    (("block_writer", "huffman-pi.input"), [2570, 2564, 2564, 2564, 2564, 2564]),
    (("block_writer", "huffman-text.input"), [235, 234, 234, 234, 234, 234]),
    (("fuzz", "roundtrip1.input"), [333, 331, 331, 331, 331, 331]),
    (("fuzz", "roundtrip2.input"), [334] * 6),
]


def expand_tokens(tokens):
    out = bytearray()
    for t in tokens:
        d = O.tok_decode(t)
        if d[0] == "L":
            out.append(d[1])
        else:
            _, dist, length = d
            assert 1 <= dist <= len(out)
            for _ in range(length):
                out.append(out[-dist])
    return bytes(out)


@pytest.mark.parametrize("path,counts", TOKEN_COUNTS)
def test_token_counts_and_expansion(path, counts):
    data = golden(*path)
    for level, want in zip(LEVELS, counts):
        toks = O.tokenize(data, level)
        assert len(toks) == want, (path, level)
        assert expand_tokens(toks) == data


# ---- (3) compressed sizes + round trip, flate.zig:87-244 ----
SIZE_CASES = [
    (("rfc1951.txt",), [11513, 11217, 11139, 11126, 11122, 11119], 20287, 36967),
    (("fuzz", "roundtrip1.input"), [373, 370, 370, 370, 370, 370], 393, 393),
    (("fuzz", "roundtrip2.input"), [373, 373, 373, 373, 373, 373], 394, 394),
    (("fuzz", "deflate-stream.expect"), [351, 347, 347, 347, 347, 347], 498, 747),
]
CONT_SIZE = {O.RAW: 0, O.GZIP: 18, O.ZLIB: 6}
WBITS = {O.RAW: -15, O.GZIP: 31, O.ZLIB: 15}


@pytest.mark.parametrize("path,gzip_sizes,huff_size,store_size", SIZE_CASES)
def test_compressed_sizes_and_roundtrip(path, gzip_sizes, huff_size, store_size):
    data = golden(*path)
    for container in (O.RAW, O.GZIP, O.ZLIB):
        for level, gz in zip(LEVELS, gzip_sizes):
            want = gz - 18 + CONT_SIZE[container]
            c = O.compress(data, container, level)
            assert len(c) == want, (path, container, level)
            st, out, used = O.decompress(c, container)
            assert st == "Ok" and out == data and used == len(c)
            assert pyzlib.decompress(c, WBITS[container]) == data
            # writer interface: bytes fed in pieces give the same stream
            d = O.Deflate(container, level)
            for i in range(0, len(data), 1000):
                d.write(data[i:i + 1000])
            d.finish()
            assert d.output() == c
            d.close()
        for mode, gz in ((O.HUFFMAN, huff_size), (O.STORE, store_size)):
            want = gz - 18 + CONT_SIZE[container]
            c = O.compress(data, container, mode)
            assert len(c) == want, (path, container, mode)
            st, out, _ = O.decompress(c, container)
            assert st == "Ok" and out == data
            assert pyzlib.decompress(c, WBITS[container]) == data


# ---- (4) exact bytes, deflate.zig:721-748 and flate.zig:356-384 ----
def test_store_and_huffman_hello_world_bytes():
    data = b"Hello world!"
    expected = bytes([0x1, 0xC, 0x0, 0xF3, 0xFF]) + data
    assert O.compress(data, O.RAW, O.STORE) == expected
    assert O.compress(data, O.RAW, O.HUFFMAN) == expected


def test_public_interface_vectors():
    plain = b"Hello world\n"
    block = bytes([0x01, 0x0C, 0x00, 0xF3, 0xFF]) + plain
    gz = bytes([0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0x03]) + block + bytes(
        [0xD5, 0xE0, 0x39, 0xB7, 0x0C, 0, 0, 0])
    zl = bytes([0x78, 0x9C]) + block + bytes([0x1C, 0xF2, 0x04, 0x47])
    assert O.compress(plain, O.GZIP, O.STORE) == gz
    assert O.compress(plain, O.ZLIB, O.STORE) == zl
    assert O.compress(plain, O.RAW, O.STORE) == block
    for c, blob in ((O.GZIP, gz), (O.ZLIB, zl), (O.RAW, block)):
        assert O.decompress(blob, c)[:2] == ("Ok", plain)
        for mode in (O.STORE, O.HUFFMAN, 6):
            assert O.decompress(O.compress(plain, c, mode), c)[:2] == ("Ok", plain)
    assert O.crc32(plain) == 0xB739E0D5 == pyzlib.crc32(plain)
    assert O.adler32(plain) == 0x1CF20447 == pyzlib.adler32(plain)
    assert O.crc32(b"ABCDEABCD ABCDEABCD") == 0xB4391C17


# ---- (5) 43 golden block files x {eof 0/1}, block_writer.zig:599-706 ----
def _cases():
    with open(os.path.join(GOLDEN, "block_writer_tokens.json")) as f:
        cases = json.load(f)
    for c in cases:
        c["tok"] = np.array([O.tok_lit(t[0]) if len(t) == 1 else O.tok_match(t[0], t[1])
                             for t in c["tokens"]], dtype=np.uint32)
    return cases


def _check_block(fn, tokens, input_bytes, want):
    got = O.block_write(fn, tokens, False, input_bytes)
    assert got == want
    assert got[0] & 1 == 0
    got = bytearray(O.block_write(fn, tokens, True, input_bytes))
    assert got[0] & 1 == 1
    got[0] &= 0xFE
    assert bytes(got) == want


@pytest.mark.parametrize("fn", ["wb", "dyn", "huff"])
def test_block_writer_goldens(fn):
    checked = 0
    for c in _cases():
        if c["input"] and c["want"]:
            inp = golden("block_writer", c["input"])
            want = golden("block_writer", c["want"].replace("{s}", fn))
            _check_block(fn, c["tok"], inp, want)
            checked += 1
        if fn == "huff":
            continue
        want = golden("block_writer", c["want_no_input"].replace("{s}", fn))
        _check_block(fn, c["tok"], None, want)
        checked += 1
    if fn == "huff":
        inp = golden("block_writer", "huffman-rand-max.input")
        _check_block("huff", np.zeros(0, np.uint32), inp, golden("block_writer", "huffman-rand-max.huff.expect"))
        checked += 1
    assert checked == {"wb": 17, "dyn": 17, "huff": 9}[fn]


def test_token_lists_expand_to_inputs():
    # the fixture tokens decode to their input files (sanity of the transcription)
    for c in _cases():
        if c["input"]:
            assert expand_tokens(c["tok"]) == golden("block_writer", c["input"])


# ---- (6) Huffman known answers, huffman_encoder.zig:363-422, 497-536 ----
def test_huffman_19_symbol_known_answer():
    freqs = [8, 1, 1, 2, 5, 10, 9, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 5]
    codes, lens = O.huffman_generate(freqs, 7)
    assert list(lens) == [3, 6, 6, 5, 3, 2, 2, 6, 0, 0, 0, 0, 0, 0, 0, 0, 6, 5, 3]
    assert sum(f * l for f, l in zip(freqs, lens)) == 141
    want = {5: 0x0, 6: 0x2, 0: 0x1, 4: 0x5, 18: 0x3, 3: 0x7, 17: 0x17, 1: 0x0F, 2: 0x2F, 7: 0x1F, 16: 0x3F}
    for sym, code in want.items():
        assert codes[sym] == code


def test_fixed_literal_code_bitstream():
    codes = np.zeros(286, np.uint16)
    lens = np.zeros(286, np.uint16)
    O.lib().fo_fixed_literal_codes(codes.ctypes.data, lens.ctypes.data)
    acc, nb, out = 0, 0, bytearray()
    for c, l in zip(codes, lens):
        acc |= int(c) << nb
        nb += int(l)
        while nb >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            nb -= 8
    assert nb == 0
    assert bytes(out) == golden("fixed_codes.bin")


def test_huffman_small_alphabets():
    # <= 2 used symbols: length 1, codes 0,1 in symbol order (huffman_encoder.zig:79-87)
    codes, lens = O.huffman_generate([0, 7, 0, 3], 15)
    assert list(lens) == [0, 1, 0, 1] and codes[1] == 0 and codes[3] == 1
    codes, lens = O.huffman_generate([0] * 30, 15)
    assert list(lens) == [0] * 30


# ---- (7) Lookup / SlidingWindow / Token unit values ----
def test_lookup_add_prev():  # Lookup.zig:86-109
    data = bytes([1, 2, 3, 4, 5, 6, 7, 8] * 3 + [1, 2, 3])
    a = np.frombuffer(data, np.uint8)
    prev = np.zeros(len(data), np.uint16)
    head = np.zeros(32768, np.uint16)
    chain = np.zeros(65536, np.uint16)
    O.lib().fo_lookup_add_all(a.ctypes.data, len(data), prev.ctypes.data, head.ctypes.data, chain.ctypes.data)
    for i in range(len(data)):
        assert prev[i] == (i - 8 if 8 <= i < 24 else 0)
    v = O.lib().fo_hash4(a[2:].ctypes.data)
    assert head[v] == 18 and chain[18] == 10 and chain[10] == 2


def test_lookup_bulk_add_equals_add():  # Lookup.zig:111-125
    data = b"Lorem ipsum dolor sit amet, consectetur adipiscing elit."
    a = np.frombuffer(data, np.uint8)
    prev = np.zeros(len(data), np.uint16)
    h1, c1 = np.zeros(32768, np.uint16), np.zeros(65536, np.uint16)
    h2, c2 = np.zeros(32768, np.uint16), np.zeros(65536, np.uint16)
    O.lib().fo_lookup_add_all(a.ctypes.data, len(data), prev.ctypes.data, h1.ctypes.data, c1.ctypes.data)
    O.lib().fo_lookup_bulk_add(a.ctypes.data, len(data), h2.ctypes.data, c2.ctypes.data)
    assert (h1 == h2).all() and (c1 == c2).all()


def test_window_match():  # SlidingWindow.zig:125-143
    data = np.frombuffer(b"Blah blah blah blah blah!", np.uint8)
    m = lambda p, c, ml: O.lib().fo_window_match(data.ctypes.data, data.size, p, c, ml)
    assert m(1, 6, 0) == 18 and m(1, 11, 0) == 13 and m(1, 16, 0) == 8 and m(1, 21, 0) == 0
    assert m(15, 20, 0) == 4 and m(15, 20, 3) == 4 and m(15, 20, 4) == 0


def test_token_codes():  # Token.zig:282-327
    lc = lambda length: O.lib().fo_length_code(length - 3)
    assert lc(4) == 258 and O.lib().fo_length_extra_bits(258) == 0
    assert lc(11) == 265 and lc(12) == 265 and O.lib().fo_length_extra_bits(265) == 1
    assert lc(130) == 280 and O.lib().fo_length_extra_bits(280) == 4
    assert lc(258) == 285 and lc(257) == 284 and lc(3) == 257
    assert O.lib().fo_distance_code(192 - 1) == 14 and O.lib().fo_distance_extra_bits(14) == 6
    assert O.lib().fo_distance_code(0) == 0 and O.lib().fo_distance_code(32767) == 29
    assert O.lib().fo_distance_code(256) == 16 and O.lib().fo_distance_code(24576) == 29


# ---- config #1 and the survey-model cross-checks (labelled survey-model) ----
def test_zero_inputs_and_config1():
    import hashlib
    assert O.compress(b"", O.RAW, 6) == bytes([0x03, 0x00])
    assert O.compress(b"\0", O.RAW, 6) == bytes([0x63, 0x00, 0x00])
    assert O.compress(b"\0" * 262, O.RAW, 6) == bytes([0x63, 0x60, 0x18, 0x05, 0x0C, 0x0C, 0x00])
    for n in (65535, 65536):
        c = O.compress(b"\0" * n, O.RAW, 6)
        assert len(c) == 78 and c[:14] == bytes.fromhex("edc081000000008 0a0fda917a902".replace(" ", ""))
    toks = O.tokenize(b"\0" * 65535, 6)
    assert len(toks) == 257
    z = b"\0" * (4 << 20)
    toks = O.tokenize(z, 6)
    assert len(toks) == 16259
    assert [O.tok_decode(t) for t in toks[:3]] == [("L", 0), ("L", 0), ("M", 1, 258)]
    assert O.tok_decode(toks[-1]) == ("M", 1, 254)
    gz = O.compress(z, O.GZIP, 6)
    assert len(gz) == 4098
    assert gz[:24].hex() == "1f8b0800000000000003" + "edc08100000000c2b0fb5307d256"
    # body tail, CRC32 LE (0x1147406a), ISIZE LE (0x00400000)
    assert gz[-12:].hex() == "0000df06" + "6a404711" + "00004000"
    assert hashlib.sha256(gz).hexdigest() == "ead941ecab79abb0d46919b9bfcb699490a1d47de07ae41b3d97a0f4f3d10b06"
    assert pyzlib.decompress(gz, 31) == z
    assert O.decompress(gz, O.GZIP, cap=len(z) + 16)[:2] == ("Ok", z)
    assert O.compress(z, O.GZIP, 4)[10:-8] == gz[10:-8] == O.compress(z, O.GZIP, 9)[10:-8]


def test_rfc1951_stream_hashes_survey_model(rfc1951):
    import hashlib
    want = {
        4: "358279a7f4b8d630cc6f037267ed1816991c26ecaf504fd2573f01d7c5af3f86",
        5: "8484b440be86f22f1b62ed1afb6b0e0541c6930db758250da579746494b0c206",
        6: "d1ed52b4ab4f57cc9e3a09500d705f4a12fd34b87f6071f0c02ac359680f74e8",
        7: "0b219260eac2e6a004bf44b4993835de394442dfae0b83d483138944fc2b11d9",
        8: "77665c2b811b3a49653ff7ccb4845b1f4a14c9b1b5706d1c9025d6b387635d28",
        9: "74ae6f0684dbfe0a671de7ef800d9ff9592b657ce3af2dc1bf11c523deb0d75b",
    }
    for level, h in want.items():
        assert hashlib.sha256(O.compress(rfc1951, O.GZIP, level)).hexdigest() == h
    huff = O.compress(rfc1951, O.RAW, O.HUFFMAN)
    assert len(huff) == 20269
    assert hashlib.sha256(huff).hexdigest() == "b64eaa0d31e1c7adf797dbb0d87fa6668c0ea36c78a35ab12354fb20a2e75c66"
