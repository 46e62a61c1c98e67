"""GPU parity tests of the inflate path: outputs and per-stream status (the reference's
error names, inflate.zig:487-527) identical to the oracle's, through the C ABI."""
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from conftest import golden
from gpu_util import engine
from test_oracle_inflate_pins import ABCD, DYN, FIXED, FUZZ, GZ_HDR, HELLO, STORED

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wave", "workgroup"])
def inflate_path(request, monkeypatch):
    """Every test of this file runs twice: with the library's own choice (these short streams go to k_inflate,
    one wave per stream) and with every stream sent through k_inflate_par first (a workgroup per stream;
    what it does not finish -- every error case -- is decoded again by k_inflate)."""
    if request.param == "workgroup":
        monkeypatch.setenv("FLATE_HIP_INFLATE_PAR", "1")
    else:
        monkeypatch.delenv("FLATE_HIP_INFLATE_PAR", raising=False)
    return request.param


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


def _long_streams():
    from flate_amd import synth
    rng = np.random.default_rng(77)
    text = synth.text(synth.SEED_TEXT + 5, 3 << 20).tobytes()
    sil = synth.silesia_like(synth.SEED_SILESIA, 2 << 20).tobytes()
    cases = []

    def z(name, data, level=6, strategy=pyzlib.Z_DEFAULT_STRATEGY):
        for container, wbits in ((0, -15), (1, 31), (2, 15)):
            co = pyzlib.compressobj(level, pyzlib.DEFLATED, wbits, 9, strategy)
            cases.append((name, container, data, co.compress(data) + co.flush()))

    z("text-l6", text[:1 << 20])
    z("text-l1", text[1 << 20:(1 << 20) + 700001], 1)
    z("text-l9", text[2 << 20:(2 << 20) + 333333], 9)
    z("mix", sil[:1500000])
    z("random", rng.integers(0, 256, 300000, dtype=np.uint8).tobytes())
    z("huffman-only", text[:400000], 6, pyzlib.Z_HUFFMAN_ONLY)
    z("rle", text[:400000], 6, pyzlib.Z_RLE)
    z("fixed", text[:150000], 6, pyzlib.Z_FIXED)
    z("zeros", bytes(3 << 20))
    z("period-2", b"ab" * 400000, 9)
    z("period-7", b"abcdefg" * 90000, 9)
    z("sparse", bytes(np.where(rng.random(500000) < 0.01, rng.integers(1, 256, 500000), 0).astype(np.uint8)))
    return cases


def test_long_streams(inflate_path):
    """Streams long enough for k_inflate_par by the library's own rule (zlib-made: other block sizes and tree
    shapes than the reference's encoder, fixed and stored blocks, runs), all containers; bytes and consumed
    counts; then the same streams damaged, against the oracle's status names."""
    eng = engine()
    cases = _long_streams()
    for container in (0, 1, 2):
        grp = [c for c in cases if c[1] == container]
        outs, st, used = eng.decompress_many([c[3] for c in grp], container, caps=[len(c[2]) + 8 for c in grp])
        for c, o, s_, u in zip(grp, outs, st, used):
            assert s_ == 0 and o == c[2] and u == len(c[3]), (c[0], container, s_, len(o), u)
    # the library's own long streams (one stream per input, levels 6 and huffman-only)
    data = cases[0][2]
    for mode in (6, 1):
        comp, st = eng.compress_many([data, data[:500000]], 1, mode)
        assert st == [0, 0]
        outs, st, _ = eng.decompress_many(comp, 1, caps=[len(data) + 8] * 2)
        assert st == [0, 0] and outs == [data, data[:500000]]
    # damaged long streams: same status name as the oracle (= the reference's error), same bytes when it decodes
    rng = np.random.default_rng(78)
    muts = []
    for name, container, data, comp in cases:
        if name not in ("text-l9", "fixed", "period-7", "random") or len(comp) < 100:
            continue
        m = bytearray(comp)
        muts.append((container, bytes(m[:len(m) // 2]), len(data)))
        m2 = bytearray(comp)
        m2[int(rng.integers(len(m2) // 2, len(m2)))] ^= 0x10
        muts.append((container, bytes(m2), len(data)))
        m3 = bytearray(comp)
        m3[-1] ^= 0xFF
        muts.append((container, bytes(m3), len(data)))
        muts.append((container, comp + b"trailing bytes", len(data)))
    for container in (0, 1, 2):
        grp = [m for m in muts if m[0] == container]
        outs, st, used = eng.decompress_many([m[1] for m in grp], container, caps=[m[2] + 8 for m in grp])
        for m, o, s_, u in zip(grp, outs, st, used):
            name, want, wused = O.decompress(m[1], container, 0, cap=m[2] + 8)
            assert O.STATUS[s_] == name, (container, len(m[1]), O.STATUS[s_], name)
            if name == "Ok":
                assert o == want and u == wused


@pytest.mark.parametrize("ring", ["", "2048", "32768"])
def test_directed_streams_that_fuzzing_rarely_builds(ring, monkeypatch):
    """tests/_inflate_edge_cases.py: fixed-block symbols 286 / 287, distance codes 30 / 31, a distance equal to / one more than what
    has been written, matches at the rings' edges, stored blocks of 0 and 65535 bytes, a wrong NLEN, zlib's long code-length runs.
    Status, bytes and consumed count of every decoder == the oracle's (which agrees with puff.c on them: test_oracle_inflate_pins)."""
    from _inflate_edge_cases import CASES
    if ring:
        monkeypatch.setenv("FLATE_HIP_INFLATE_RING", ring)
    eng = engine()
    names = sorted(CASES)
    streams = [CASES[n] for n in names]
    back, st, used = eng.decompress_many(streams, O.RAW, caps=[80000] * len(streams))
    for n, s, b, stt, u in zip(names, streams, back, st, used):
        name, want, wused = O.decompress(s, O.RAW, 0, cap=80000)
        assert O.STATUS[stt] == name, (n, O.STATUS[stt], name)
        if name == "Ok":
            assert b == want and u == wused, n


@pytest.mark.parametrize("ring", ["2048", "32768"])
def test_fast_round_edges(ring, monkeypatch):
    """(k_inflate itself, both ring sizes: the long-stream kernels are switched off.)  What the fast rounds of k_inflate decide on (kernels_inflate.h): matches at the distances where the source moves
    from the LDS ring to the output buffer (ring - 260 = 1788), at the largest distance, matches that read the round's
    own output (runs, short periods) next to literals and far matches, rounds of more than 260 bytes, and output slots
    that are exactly full or 8 bytes short (the oracle's status for those).  zlib makes the streams (dynamic blocks)."""
    monkeypatch.setenv("FLATE_HIP_INFLATE_PAR", "0")
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "0")
    monkeypatch.setenv("FLATE_HIP_INFLATE_RING", ring)
    eng = engine()
    rng = np.random.default_rng(4321)
    parts = []
    for d in (259, 260, 261, 1787, 1788, 1789, 1790, 2047, 2048, 2049, 4095, 32767, 32768):
        blk = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        parts.append(blk + blk[:24 + d % 7])  # a match of 24..30 bytes at distance d
    edges = b"".join(parts)
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 9)), dtype=np.uint8)) for _ in range(300)]
    mixed = bytearray()
    while len(mixed) < 600000:
        k = int(rng.integers(0, 10))
        if k == 0:
            mixed += bytes([int(rng.integers(0, 256))]) * int(rng.integers(3, 700))        # runs: distance 1, up to 258 a token
        elif k == 1:
            mixed += bytes(rng.integers(0, 256, 2, dtype=np.uint8)) * int(rng.integers(2, 300))  # period 2
        elif k == 2:
            mixed += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))  # literals
        else:
            mixed += b" ".join(words[int(i)] for i in rng.integers(0, 300, int(rng.integers(1, 12))))
    mixed = bytes(mixed)
    mixed = mixed[:len(mixed) & ~7]
    edges = edges[:len(edges) & ~7]
    streams, datas, caps = [], [], []
    for data in (edges, mixed, mixed[:65528], edges[:32768] + mixed[:32768]):
        for level in (1, 6, 9):
            co = pyzlib.compressobj(level, pyzlib.DEFLATED, -15, 9)
            comp = co.compress(data) + co.flush()
            for cap in (len(data) + 8, len(data), len(data) - 8):
                streams.append(comp)
                datas.append(data)
                caps.append(cap)
    outs, st, used = eng.decompress_many(streams, 0, caps=caps)
    for comp, data, cap, o, s_, u in zip(streams, datas, caps, outs, st, used):
        name, want, wused = O.decompress(comp, 0, 0, cap=cap)
        assert O.STATUS[s_] == name, (len(data), cap, O.STATUS[s_], name)
        assert (name == "Ok") == (cap >= len(data))
        if name == "Ok":
            assert o == data and o == want and u == wused == len(comp)


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


def test_gpu_inflate_agrees_with_puff_on_mutated_streams():
    """The reference's differential harness (bin/fuzz_puff.zig:42-50): raw deflate input through
    puff.c and through the inflater; both fail or both produce the same bytes.  Here the inflater
    is the GPU kernel (default flags: RFC-conformant headers, as puff accepts them)."""
    if not O.puff_available():
        pytest.skip("oracle/_ref/libpuff.so not built")
    eng = engine()
    rng = np.random.default_rng(77)
    from flate_amd import synth
    base = [O.compress(synth.text(synth.SEED_TEXT + k, 3000 + 517 * k).tobytes(), O.RAW, lvl)
            for k, lvl in enumerate((4, 6, 9, O.HUFFMAN, O.STORE))]
    base += [O.compress(bytes(5000), O.RAW, 6), O.compress(rng.integers(0, 256, 4000, dtype=np.uint8).tobytes(), O.RAW, 6)]
    cases = list(base)
    for s in base:
        b = bytearray(s)
        for _ in range(12):
            m = bytearray(b)
            kind = int(rng.integers(0, 4))
            if kind == 0 and len(m) > 4:
                del m[int(rng.integers(1, len(m))):]
            elif kind == 1:
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:
                i = int(rng.integers(0, len(m)))
                m[i:i + 3] = rng.integers(0, 256, 3, dtype=np.uint8).tobytes()
            else:
                m += rng.integers(0, 256, 5, dtype=np.uint8).tobytes()
            cases.append(bytes(m))
    outs, st, used = eng.decompress_many(cases, O.RAW, caps=[1 << 17] * len(cases))
    agree_ok = agree_err = 0
    for c, o, s in zip(cases, outs, st):
        rc, want = O.puff(c, cap=1 << 17)
        if rc == 0:
            assert s == 0 and o == want, (rc, s)
            agree_ok += 1
        else:
            assert s != 0, (rc, s)
            agree_err += 1
    assert agree_ok >= len(base) and agree_err > 0


def test_cli_gzip_gunzip_round_trip(tmp_path):
    """tools/gzip.py / tools/gunzip.py (bin/gzip.zig:20, bin/gunzip.zig:25-27): the .gz is the oracle's,
    gunzip restores the file, a second member appended to the file is decoded too."""
    import importlib.util
    engine()
    from conftest import ROOT

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    from flate_amd import synth
    data = synth.text(synth.SEED_TEXT, 200001).tobytes()
    f = tmp_path / "a.txt"
    f.write_bytes(data)
    assert load("gzip").main([str(f)]) == 0
    gz = (tmp_path / "a.txt.gz").read_bytes()
    assert gz == O.compress(data, O.GZIP, 6)
    f.unlink()
    assert load("gunzip").main([str(f) + ".gz"]) == 0
    assert f.read_bytes() == data
    (tmp_path / "b.gz").write_bytes(gz + O.compress(b"second member", O.GZIP, 9))
    assert load("gunzip").main([str(tmp_path / "b.gz")]) == 0
    assert (tmp_path / "b").read_bytes() == data + b"second member"
    assert load("gunzip").main([str(f)]) == 1  # not a .gz name


def test_host_buffers_get_nothing_of_an_earlier_call_beyond_out_len():
    """Host-buffer calls stage the output slots on the device: what comes back beyond out_len[i] (the copy takes whole
    ranges of slots) is zeros or the caller's own bytes, never what an earlier call left in the staging buffer."""
    eng = engine()
    L, h = eng._L, eng._h
    rng = np.random.default_rng(3)
    first = [bytes([0x5A]) * 50000 for _ in range(12)]           # an earlier call fills the staging buffer with 0x5A
    outs, st, _ = eng.decompress_many([pyzlib.compress(d, 6) for d in first], 2, caps=[50000] * 12)
    assert st == [0] * 12 and outs == first
    datas = [bytes(rng.integers(1, 90, int(n), dtype=np.uint8)) for n in (3000, 17, 12000, 1, 700, 9000, 40, 2500)]
    streams = [pyzlib.compress(d, 6) for d in datas]
    n = len(streams)
    for slot in (16384, 400000):  # (dense: the range is copied as it is; sparse: packed on the device first)
        in_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in streams], out=in_off[1:])
        blob = np.frombuffer(b"".join(streams), dtype=np.uint8)
        out_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(slot)).astype(np.uint64)
        out = np.full(n * slot + 8, 0xAA, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        consumed = np.zeros(n, dtype=np.uint64)
        rc = L.flate_hip_decompress_batch(h, blob.ctypes.data, in_off.ctypes.data, n, 2, 0, out.ctypes.data, out_off.ctypes.data,
                                          out_len.ctypes.data, status.ctypes.data, consumed.ctypes.data, 0)
        assert rc == 0 and list(status) == [0] * n
        for i, d in enumerate(datas):
            got = out[i * slot:(i + 1) * slot]
            assert int(out_len[i]) == len(d) and got[:len(d)].tobytes() == d
            rest = np.unique(got[len(d):])
            assert set(int(v) for v in rest) <= {0x00, 0xAA}, (slot, i, rest[:8])
