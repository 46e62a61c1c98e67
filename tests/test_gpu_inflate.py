"""GPU parity tests of the inflate path: outputs and per-stream status (the reference's
error names, inflate.zig:487-527) identical to the oracle's, through the C ABI."""
import numpy as np
import pytest

import _oracle as O
from conftest import golden
from gpu_util import engine
from test_oracle_inflate_pins import ABCD, DYN, FIXED, FUZZ, GZ_HDR, HELLO, STORED

pytestmark = pytest.mark.gpu


def test_reference_vectors():
    eng = engine()
    outs, st, used = eng.decompress_many([STORED, FIXED, DYN], 0)
    assert st == [0, 0, 0] and outs == [HELLO, HELLO, ABCD]
    assert used == [len(STORED), len(FIXED), len(DYN)]
    foot = bytes([0xD5, 0xE0, 0x39, 0xB7, 0x0C, 0, 0, 0])
    named = bytes([0x1F, 0x8B, 0x08, 0x08, 0xE5, 0x70, 0xB1, 0x65, 0x00, 0x03, 0x68, 0x65, 0x6C, 0x6C, 0x6F, 0x2E,
                   0x74, 0x78, 0x74, 0x00]) + FIXED + foot
    gz = [GZ_HDR + STORED + foot, GZ_HDR + DYN + bytes([0x17, 0x1C, 0x39, 0xB4, 0x13, 0, 0, 0]), named]
    outs, st, _ = eng.decompress_many(gz, 1)
    assert st == [0, 0, 0] and outs == [HELLO, ABCD, HELLO]
    zl = bytes([0x78, 0x9C]) + STORED + bytes([0x1C, 0xF2, 0x04, 0x47])
    outs, st, _ = eng.decompress_many([zl], 2)
    assert st == [0] and outs == [HELLO]


@pytest.mark.parametrize("flags", [0, 1])
def test_fuzz_corpus_statuses(flags):
    eng = engine()
    datas = [golden("fuzz", f[0] + ".input") for f in FUZZ]
    outs, st, _ = eng.decompress_many(datas, 0, flags=flags)
    for (name, err, out), got, s, d in zip(FUZZ, outs, st, datas):
        want_st, want_out, _ = O.decompress(d, 0, flags=flags)
        assert O.STATUS[s] == want_st, name
        assert O.STATUS[s] == (err or "Ok"), name
        if err is None:
            assert got == want_out, name


def test_header_footer_errors():
    eng = engine()
    z = [bytes([0x78]), bytes([0x79, 0x94]), bytes([0x88, 0x98]), bytes([0x78, 0xDA, 0x03, 0, 0, 0, 0, 0]),
         bytes([0x78, 0xDA, 0x03, 0, 0])]
    _, st, _ = eng.decompress_many(z, 2)
    assert [O.STATUS[s] for s in st] == ["EndOfStream", "BadZlibHeader", "BadZlibHeader", "WrongZlibChecksum",
                                         "EndOfStream"]
    g = [bytes([0x1F, 0x8B]), bytes([0x1F, 0x8B, 0x09, 0, 0, 0, 0, 0, 0, 0x03]),
         GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0x01, 0, 0, 0, 0]), GZ_HDR + bytes([0x03, 0, 0, 0, 0]),
         GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0x01]), GZ_HDR + bytes([0x03, 0, 0, 0, 0, 0, 0, 0, 0])]
    _, st, _ = eng.decompress_many(g, 1)
    assert [O.STATUS[s] for s in st] == ["EndOfStream", "BadGzipHeader", "WrongGzipChecksum", "EndOfStream",
                                         "WrongGzipSize", "EndOfStream"]
    _, st, _ = eng.decompress_many([bytes([0x07, 0x00])], 0)
    assert O.STATUS[st[0]] == "InvalidBlockType"


def test_two_concatenated_zlib_streams():
    eng = engine()
    a, b = golden("fuzz", "first.input"), golden("fuzz", "second.input")
    outs, st, used = eng.decompress_many([a + b], 2)
    assert st == [0] and used == [len(a)]
    outs2, st2, used2 = eng.decompress_many([(a + b)[used[0]:]], 2)
    assert st2 == [0] and outs[0] + outs2[0] == golden("fuzz", "first.expect") + golden("fuzz", "second.expect")


def test_roundtrip_all_modes_and_q6(rfc1951):
    eng = engine()
    from flate_amd import synth
    rng = np.random.default_rng(77)
    datas = [b"", b"x", rfc1951, synth.text(synth.SEED_TEXT, 200000).tobytes(),
             rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), bytes(300000),
             synth.silesia_like(synth.SEED_SILESIA, 400000).tobytes(),
             rfc1951[20395:20395 + 1662]]  # Q6 reproducer
    for container in (0, 1, 2):
        for mode in (0, 1, 6, 9):
            streams = [O.compress(d, container, mode) for d in datas]
            outs, st, used = eng.decompress_many(streams, container, caps=[len(d) + 8 for d in datas])
            assert st == [0] * len(datas), (container, mode, st)
            assert outs == datas
            assert used == [len(s) for s in streams]
    # strict mode rejects the cross-boundary repeat like the reference does
    c = O.compress(datas[-1], 0, 6)
    _, st, _ = eng.decompress_many([c], 0, flags=1)
    assert O.STATUS[st[0]] == "InvalidDynamicBlockHeader"
    # too small an output slot is reported
    _, st, _ = eng.decompress_many([O.compress(rfc1951, 0, 6)], 0, caps=[1000])
    assert st == [100]


def test_gpu_compress_then_gpu_inflate():
    eng = engine()
    from flate_amd import synth
    data = synth.text(synth.SEED_TEXT + 9, 20 * 65535).tobytes()
    chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
    for container in (0, 1):
        comp, st = eng.compress_many(chunks, container, 6)
        assert st == [0] * len(chunks)
        outs, st, _ = eng.decompress_many(comp, container, caps=[65536] * len(chunks))
        assert st == [0] * len(chunks) and b"".join(outs) == data
