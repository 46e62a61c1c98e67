"""GPU parity tests of the inflate path: outputs and per-stream status (the reference's
error names, inflate.zig:487-527) identical to the oracle's, through the C ABI."""
import zlib as pyzlib

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


def _mutants(seed, n_per_base=120):
    """Differential fuzzing in the spirit of the reference's bin/fuzz_puff.zig: valid streams of
    every block type, then truncated, bit-flipped, byte-smashed and spliced."""
    from flate_amd import synth
    rng = np.random.default_rng(seed)
    text = synth.text(synth.SEED_TEXT + 33, 6000).tobytes()
    rnd = rng.integers(0, 256, 700, dtype=np.uint8).tobytes()
    bases = []
    for data in (text, text[:300], rnd, bytes(900), b"abc" * 200, b"", text[:2000] + rnd[:300] + bytes(500)):
        for container in (0, 1, 2):
            for mode in (0, 1, 4, 9):
                bases.append((container, O.compress(data, container, mode)))
        bases.append((0, pyzlib.compress(data, 1)[2:-4]))     # zlib's encoder: fixed blocks, other tree shapes
        bases.append((0, pyzlib.compress(data, 9)[2:-4]))
    out = []
    for container, b in bases:
        out.append((container, b))
        for _ in range(n_per_base // 10):
            m = bytearray(b)
            kind = int(rng.integers(0, 5))
            if kind == 0 and len(m) > 1:
                m = m[: int(rng.integers(0, len(m)))]
            elif kind == 1 and len(m):
                for _ in range(int(rng.integers(1, 4))):
                    i = int(rng.integers(0, len(m)))
                    m[i] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2 and len(m):
                i = int(rng.integers(0, len(m)))
                m[i] = int(rng.integers(0, 256))
            elif kind == 3 and len(m) > 4:
                i = int(rng.integers(0, len(m) - 2))
                m[i:i + 2] = rng.integers(0, 256, 2, dtype=np.uint8).tobytes()
            else:
                other = bases[int(rng.integers(0, len(bases)))][1]
                cut = int(rng.integers(0, len(m) + 1))
                m = m[:cut] + other[int(rng.integers(0, len(other) + 1)):]
            out.append((container, bytes(m)))
    return out


@pytest.mark.parametrize("flags", [0, 1])
def test_differential_fuzz_against_oracle(flags):
    # same status name as the oracle (= the reference's error for that input) on every mutant,
    # same bytes and same consumed count whenever the stream decodes
    eng = engine()
    muts = _mutants(4242)
    cap = 1 << 16
    for container in (0, 1, 2):
        streams = [m for c, m in muts if c == container]
        outs, st, used = eng.decompress_many(streams, container, flags, caps=[cap] * len(streams))
        bad = []
        for i, s in enumerate(streams):
            name, want, wused = O.decompress(s, container, flags, cap=cap)
            got_name = O.STATUS[st[i]]
            if got_name != name or (name == "Ok" and (outs[i] != want or used[i] != wused)):
                bad.append((i, len(s), got_name, name, len(outs[i]), len(want), used[i], wused))
        assert not bad, (container, flags, len(bad), bad[:5])
