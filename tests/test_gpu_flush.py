"""GPU parity tests of sync flush (Compressor.write / flush / finish, deflate.zig:335-367):
the stream with its flush markers is byte-identical to the oracle's streaming compressor fed the
same calls, LZ history across the flushes included."""
import io
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from gpu_util import engine

pytestmark = pytest.mark.gpu

WBITS = {0: -15, 1: 31, 2: 15}


def _oracle_stream(data, flushes, finish, container, level, tokens=False):
    d = O.Deflate(container, level, log_tokens=tokens)
    prev = 0
    for f in flushes:
        d.write(data[prev:f])
        d.flush()
        prev = f
    if finish:
        d.write(data[prev:])
        d.finish()
    out = d.output()
    toks = d.tokens() if tokens else None
    d.close()
    return out, toks


def _cases():
    from flate_amd import synth
    rng = np.random.default_rng(5)
    text = synth.text(synth.SEED_TEXT + 11, 400000).tobytes()
    rnd = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    z = 65536 - 262
    return [
        (b"", [0], True),
        (b"", [0, 0], True),
        (b"hello hello hello hello", [5], True),
        (b"hello hello hello hello", [5, 5, 23], True),
        (b"abcabcabcabcabcabc" * 10, [1, 2, 3, 4, 5, 6, 7, 8, 100, 179, 180], True),
        (text[:1000], [1000], False),
        (text[:30000], [100, 101, 5000, 29999], True),
        (text[:70000], [40000], True),
        (text[:70000], [65535], True),
        (text[:70000], [65536], True),
        (text[:70000], [z - 1], True), (text[:70000], [z], True), (text[:70000], [z + 1], True),
        (text[:70000], [z + 100, z + 200], True),
        (text[:140000], [z + 50, 65536 + 100, 98304 - 100, 98304, 98304 + 5], True),
        (text, [1000, 70000, 70001, 200000, 399999], True),
        (text, [400000], False),
        (text[:200000], list(range(0, 200000, 7919)), True),
        (rnd[:100000], [33333, 66666], True),
        (bytes(150000), [1, 65000, 65536, 100000], True),
        ((text[:3000] + rnd[:500]) * 40, [12345, 70000, 100000], True),
    ]


CASES = _cases()


@pytest.mark.parametrize("level", [4, 6, 9])
def test_flush_streams_match_oracle(level):
    eng = engine()
    for i, (data, flushes, finish) in enumerate(CASES):
        got, st = eng.compress_flush(data, flushes, finish, O.RAW, level)
        assert st == 0, (i, st)
        want, wtok = _oracle_stream(data, flushes, finish, O.RAW, level, tokens=True)
        if got != want:
            toks = eng.debug_tokens(0)
            m = min(len(toks), len(wtok))
            bad = np.nonzero(toks[:m] != wtok[:m])[0]
            first = int(bad[0]) if bad.size else m
            raise AssertionError((i, level, len(data), flushes[:6], len(got), len(want), "first differing token", first,
                                  O.tok_decode(toks[first]) if first < len(toks) else None,
                                  O.tok_decode(wtok[first]) if first < len(wtok) else None))
        if finish:
            assert pyzlib.decompress(got, -15) == data


@pytest.mark.parametrize("container", [1, 2])
def test_flush_with_containers(container):
    eng = engine()
    for data, flushes, finish in CASES[2:12]:
        got, st = eng.compress_flush(data, flushes, finish, container, 6)
        assert st == 0
        assert got == _oracle_stream(data, flushes, finish, container, 6)[0]
        if finish:
            assert pyzlib.decompress(got, WBITS[container]) == data


def test_flushed_prefix_is_decodable_and_a_prefix():
    # what a flush has pushed out must already decode to the data written so far
    eng = engine()
    data = CASES[15][0]
    flushes = [1000, 70000, 200000]
    full, st = eng.compress_flush(data, flushes, True, O.RAW, 6)
    assert st == 0
    for k in range(1, len(flushes) + 1):
        part, st = eng.compress_flush(data[:flushes[k - 1]], flushes[:k], False, O.RAW, 6)
        assert st == 0 and full.startswith(part) and part.endswith(b"\x00\x00\xff\xff")
        d = pyzlib.decompressobj(-15)
        assert d.decompress(part) == data[:flushes[k - 1]]


def test_compressor_object_flush_mirror():
    # flate.zig compressor(): write / flush / write / finish through the Python mirror
    engine()
    from flate_amd import gzip
    data = CASES[15][0][:150000]
    sink = io.BytesIO()
    c = gzip.compressor(sink, gzip.Options(level=gzip.Level.default))
    c.write(data[:50000])
    c.flush()
    n1 = len(sink.getvalue())
    d = pyzlib.decompressobj(31)
    assert d.decompress(sink.getvalue()) == data[:50000]
    c.write(data[50000:])
    c.flush()
    c.finish()
    out = sink.getvalue()
    assert n1 < len(out) and pyzlib.decompress(out, 31) == data
    assert out == _oracle_stream(data, [50000, 150000], True, O.GZIP, 6)[0]


def test_flush_fuzz_against_oracle():
    # random data structure x random flush positions (clustered around the window boundaries)
    from test_gpu_stream import _fuzz_input
    eng = engine()
    rng = np.random.default_rng(99)
    for i in range(40):
        data = _fuzz_input(2000 + i)[: int(rng.integers(1, 250000))]
        n = len(data)
        marks = [65536 - 262, 65536, 98304 - 262, 98304, 131072 - 262, 131072, 32768]
        pts = set()
        for _ in range(int(rng.integers(1, 9))):
            if rng.random() < 0.6:
                pts.add(int(min(n, max(0, rng.choice(marks) + rng.integers(-300, 300)))))
            else:
                pts.add(int(rng.integers(0, n + 1)))
        flushes = sorted(pts)
        if rng.random() < 0.3:
            flushes.append(flushes[-1])  # flush twice
        finish = bool(rng.random() < 0.8)
        if not finish:
            flushes = [f for f in flushes if f < n] + [n]
        for level in (4, 6, 9):
            got, st = eng.compress_flush(data, flushes, finish, O.RAW, level)
            assert st == 0
            want = _oracle_stream(data, flushes, finish, O.RAW, level)[0]
            assert got == want, (i, level, n, flushes, finish, len(got), len(want))


@pytest.mark.parametrize("mode", [0, 1])
def test_simple_compressors_flush(mode):
    # SimpleCompressor.flush (deflate.zig:474-478): what the buffer holds as a block of its own,
    # then the empty stored block; blocks of 65535 bytes restart after every flush
    eng = engine()
    base = CASES[15][0]
    seqs = [(b"", [0], True), (b"abc", [0, 1, 1, 3], True), (base[:1000], [1000], False),
            (base[:200000], [65535, 131070, 131071], True), (base[:200000], [70000, 70000 + 65535], True),
            (base[:140000], [65534, 65536], False)]
    for container in (0, 1, 2):
        for data, flushes, finish in seqs:
            fl = flushes if finish or flushes[-1] == len(data) else flushes + [len(data)]
            got, st = eng.compress_flush(data, fl, finish, container, mode)
            assert st == 0
            assert got == _oracle_stream(data, fl, finish, container, mode)[0], (mode, container, len(data), fl, finish)
            if finish:
                assert pyzlib.decompress(got, WBITS[container]) == data


@pytest.mark.parametrize("container", [O.RAW, O.GZIP, O.ZLIB])
def test_incremental_compressor_object_matches_streaming_oracle(container):
    """The Compressor mirror compresses only the retained tail of the stream at every flush (the
    reference keeps its 64 KiB window, deflate.zig:335-337): same bytes as the oracle's streaming
    compressor fed the same calls, for a stream many windows long, at levels 4 / 6 / 9 and in the
    simple modes; the tail it keeps stays bounded."""
    import io
    from flate_amd import api, synth
    rng = np.random.default_rng(99)
    data = (synth.text(synth.SEED_TEXT + 2, 300000).tobytes() + bytes(70000) +
            synth.silesia_like(synth.SEED_SILESIA + 2, 400000).tobytes())
    cuts = sorted(set(int(x) for x in rng.integers(1, len(data), 14)) | {65536, 65535, 98304, 131072, 131073})
    for mode in (4, 6, 9, O.HUFFMAN, O.STORE):
        w = io.BytesIO()
        c = api._Compressor(container, mode, w)
        d = O.Deflate(container, mode)
        prev = 0
        for k, f in enumerate(cuts):
            c.write(data[prev:f]); d.write(data[prev:f])
            c.flush(); d.flush()
            if k % 5 == 0:
                c.flush(); d.flush()  # twice in a row
            assert w.getvalue() == d.output(), (mode, k, f)
            assert len(c._buf) <= api._KEEP + api._STEP + (f - prev) + 1
            prev = f
        c.write(data[prev:]); d.write(data[prev:])
        c.finish(); d.finish()
        assert w.getvalue() == d.output(), mode
        with pytest.raises(api.InvalidState):
            c.write(b"x")


def test_flush_cost_does_not_grow_with_the_stream():
    """flush k of n costs O(new bytes): late flushes of a long stream take about as long as early ones."""
    import io
    import time
    from flate_amd import api, synth
    data = synth.text(synth.SEED_TEXT + 5, 6 << 20).tobytes()
    c = api._Compressor(O.GZIP, 6, io.BytesIO())
    piece = 32768
    times = []
    for i in range(0, len(data), piece):
        c.write(data[i:i + piece])
        t0 = time.perf_counter()
        c.flush()
        times.append(time.perf_counter() - t0)
    c.finish()
    early = sorted(times[8:40])[16]
    late = sorted(times[-32:])[16]
    assert late < 2.0 * early, (early, late)
    import zlib as pyzlib
    assert pyzlib.decompress(c._wrt.getvalue(), 31) == data


def test_flush_streams_take_the_windows_at_levels_4_to_7_and_both_ways_agree():
    """Round 6: streams with sync-flush points at levels 4-7 on k_lz_chain<true> / k_lz_parse<true> (the three positions before a
    flush point stay out of the hash table, no match crosses one): no k_lz_sort / k_lz_match launch by the library's own choice, bytes
    == the oracle's Deflate object fed the same writes and flushes; flush points inside matches and runs, at the slide table's edges,
    several in a row.  The sort / match tiles (FLATE_HIP_STREAM_WINDOWS=0, and levels 8-9) are held to the same streams: this file's
    parity tests again in a process with the knob set."""
    import subprocess
    import sys
    from flate_amd import synth
    eng = engine()
    rng = np.random.default_rng(31)
    text = synth.text(synth.SEED_TEXT + 3, 500000).tobytes()
    z = 65536 - 262
    inputs = [
        (text[:300000], [5, 100, z - 1, z, z + 1, z + 32768, 131072 - 2, 200000, 200001, 200002, 200003]),
        (bytes(150000), [1, 2, 3, 70000, z + 3]),
        ((text[:89] * 4000)[:250000], [44, z - 4, z + 32768 + 2, 249999]),
        (b"".join(text[k * 300:k * 300 + 120] + bytes(int(rng.integers(4, 2000))) for k in range(200)), [1000, 65274, 65278, 98040, 150000]),
        (text[:70000], [69997, 69998, 69999, 70000]),
    ]
    for level in (4, 5, 6, 7):
        for data, fl in inputs:
            fl = sorted(f for f in fl if f <= len(data))
            for finish in (True, False):
                f2 = fl if finish else [f for f in fl if f < len(data)] + [len(data)]
                eng.profile_enable(True)
                eng.profile_reset()
                got, st = eng.compress_flush(data, f2, finish, O.GZIP, level)
                prof = eng.profile_read()
                eng.profile_enable(False)
                assert st in (0, 102) and got == _oracle_stream(data, f2, finish, O.GZIP, level)[0], (level, len(data), f2, finish)
                # (periodic data -- the zeros, the period of 89 -- is handed to the tiles by the windows themselves: DESIGN 4b)
                if data is not inputs[1][0] and data is not inputs[2][0]:
                    assert "k_lz_parse" in prof and "k_lz_match" not in prof, (level, len(data), prof)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    me = os.path.abspath(__file__)
    env = dict(os.environ, FLATE_HIP_STREAM_WINDOWS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", me + "::test_flush_streams_match_oracle",
                        me + "::test_flush_with_containers", me + "::test_flush_fuzz_against_oracle",
                        me + "::test_incremental_compressor_object_matches_streaming_oracle"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
