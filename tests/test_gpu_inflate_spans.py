"""GPU parity tests of the span path of the inflater (one long stream decoded by many workgroups at once:
kernels_inflate_par.h "spans", k_span_scan / k_inflate_span / k_span_resolve): bytes, consumed counts and the
reference's error names (inflate.zig:487-527) against the oracle, through the C ABI.

The library takes the path for streams of at least FLATE_HIP_INFLATE_SPANS compressed bytes (default 128 KiB);
most tests here lower that bound so that ordinary test-sized streams -- and every damaged one -- go through it."""
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from gpu_util import engine
from test_gpu_inflate import _long_streams, _mutants

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["symbols", "two_runs"])
def _span_decode_mode(request, monkeypatch):
    """Every test of this module twice: a span decoded ONCE, in 16-bit symbols (a byte's value, or which byte of the 32 KiB
    before the span it equals) -- the default since round 5 --, and round 4's two decodes with two fillings of that history
    (FLATE_HIP_SPAN_TWO_RUNS=1: kept as the cross-check of the first)."""
    if request.param == "two_runs":
        monkeypatch.setenv("FLATE_HIP_SPAN_TWO_RUNS", "1")
    else:
        monkeypatch.delenv("FLATE_HIP_SPAN_TWO_RUNS", raising=False)
    yield


def _kernels(eng, fn):
    eng.profile_reset()
    eng.profile_enable(True)
    try:
        res = fn()
    finally:
        prof = eng.profile_read()
        eng.profile_enable(False)
    return res, prof


@pytest.mark.parametrize("container", [0, 1, 2])
@pytest.mark.parametrize("mode", [6, 4, 9, O.HUFFMAN, O.STORE])
def test_one_long_stream_takes_the_span_path(container, mode):
    """The reference's own stream (oracle-made, 6 MiB of the Silesia-like mix): decoded by spans -- the profile shows
    the kernels -- byte for byte, with the container's checksum and the consumed count."""
    from flate_amd import synth
    eng = engine()
    data = synth.silesia_like(synth.SEED_SILESIA + 11 + mode, 6 * 1024 * 1024 + 4321).tobytes()
    comp = O.compress(data, container, mode)
    assert len(comp) >= 600 * 1024  # (the library cuts streams of at least 128 KiB)
    (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many([comp], container, caps=[len(data)]))
    assert st == [0] and used == [len(comp)]
    assert outs[0] == data
    assert "k_inflate_span" in prof and "k_span_scan" in prof, prof
    assert prof.get("k_inflate_par", (0.0, 0))[0] < 1.0, prof  # (it ran, and left at once: the stream was done)


def test_history_dependent_tails_are_resolved(monkeypatch):
    """Data whose every byte is a copy of a copy ... of the first few (periods, runs, a text made of one sentence):
    the tail of every span depends on the history before it all the way to the start of the stream."""
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "40000")
    eng = engine()
    rng = np.random.default_rng(5)
    sentence = b"the quick brown fox jumps over the lazy dog; "
    cases = [b"ab" * 3_000_000, bytes(5_000_000), b"abcdefg" * 700_000, sentence * 120_000,
             bytes(np.where(rng.random(4_000_000) < 0.002, rng.integers(1, 256, 4_000_000), 7).astype(np.uint8))]
    for data in cases:
        for level in (1, 9):  # zlib's encoder: long blocks, distance-1 runs
            co = pyzlib.compressobj(level, pyzlib.DEFLATED, 31)
            comp = co.compress(data) + co.flush()
            (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many([comp], 1, caps=[len(data)]))
            assert st == [0] and used == [len(comp)] and outs[0] == data, (len(data), level)
        comp = O.compress(data, 2, 6)
        outs, st, used = eng.decompress_many([comp], 2, caps=[len(data)])
        assert st == [0] and used == [len(comp)] and outs[0] == data


def test_long_streams_of_every_kind_with_a_low_bound(monkeypatch):
    """test_gpu_inflate's long streams (zlib-made: other block sizes and tree shapes, fixed and stored blocks, runs)
    with the bound lowered so that all of them are cut; then damaged, against the oracle's status names."""
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "20000")
    eng = engine()
    cases = _long_streams()
    for container in (0, 1, 2):
        grp = [c for c in cases if c[1] == container]
        outs, st, used = eng.decompress_many([c[3] for c in grp], container, caps=[len(c[2]) + 8 for c in grp])
        for c, o, s_, u in zip(grp, outs, st, used):
            assert s_ == 0 and o == c[2] and u == len(c[3]), (c[0], container, s_, len(o), u)
    rng = np.random.default_rng(79)
    muts = []
    for name, container, data, comp in cases:
        if len(comp) < 30000:
            continue
        m = bytearray(comp)
        muts.append((container, bytes(m[:len(m) // 2]), len(data)))            # truncated
        for _ in range(3):                                                    # one bit somewhere
            m2 = bytearray(comp)
            m2[int(rng.integers(0, len(m2)))] ^= 1 << int(rng.integers(0, 8))
            muts.append((container, bytes(m2), len(data)))
        m3 = bytearray(comp)
        m3[-1] ^= 0xFF                                                        # the footer
        muts.append((container, bytes(m3), len(data)))
        muts.append((container, comp + b"trailing bytes", len(data)))
        muts.append((container, comp + comp, len(data)))                      # two members: consumed stops behind the first
        muts.append((container, comp, (len(data) // 8) * 8 - 8))              # a slot too small (the engine rounds slots up to 8)
    for container in (0, 1, 2):
        grp = [m for m in muts if m[0] == container]
        outs, st, used = eng.decompress_many([m[1] for m in grp], container, caps=[m[2] for m in grp])
        for m, o, s_, u in zip(grp, outs, st, used):
            cap = (m[2] + 7) & ~7  # (what the engine makes of it)
            name, want, wused = O.decompress(m[1], container, 0, cap=cap)
            assert O.STATUS[s_] == name, (container, len(m[1]), cap, O.STATUS[s_], name)
            if name == "Ok":
                assert o == want and u == wused, (container, len(m[1]))


def test_differential_fuzz_through_the_span_path(monkeypatch):
    """The mutants of test_gpu_inflate (every block type, truncated, bit-flipped, spliced) with the bound at its
    lowest: whatever the span path cannot finish must come out of the old path with the reference's verdict."""
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "64")
    eng = engine()
    muts = _mutants(4242, n_per_base=60)
    for container in (0, 1, 2):
        grp = [m[1] for m in muts if m[0] == container]
        for k in range(0, len(grp), 48):  # (the path takes batches of at most 64 long streams)
            part = grp[k:k + 48]
            outs, st, used = eng.decompress_many(part, container, caps=[70000] * len(part))
            for b, o, s_, u in zip(part, outs, st, used):
                name, want, wused = O.decompress(b, container, 0, cap=70000)
                assert O.STATUS[s_] == name, (container, len(b), O.STATUS[s_], name)
                if name == "Ok":
                    assert o == want and u == wused


def test_runs_of_stored_blocks_are_cut_at_stored_headers():
    """Incompressible stretches between text (zlib and the oracle store them): the only places to cut at inside a run
    of stored blocks are the stored headers; targets inside a stored block, before a run, behind its last block."""
    from flate_amd import synth
    eng = engine()
    rng = np.random.default_rng(17)
    text = synth.text(synth.SEED_TEXT + 31, 3 << 20).tobytes()
    noise = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    datas = [text[:1 << 20] + noise(2 << 20) + text[1 << 20:2 << 20] + noise(70000) + text[2 << 20:] + noise(1 << 20),
             noise(3 << 20) + text[:300000] + noise(3 << 20),
             noise(5 << 20)]
    for data in datas:
        for comp, container in ((pyzlib.compress(data, 6), 2), (O.compress(data, 1, 6), 1), (O.compress(data, 0, O.STORE), 0)):
            assert len(comp) >= 600 * 1024
            (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many([comp], container, caps=[len(data)]))
            assert st == [0] and used == [len(comp)] and outs[0] == data
            assert "k_inflate_span" in prof and prof.get("k_inflate_par", (0.0, 0))[0] < 1.0, prof


def test_a_pool_that_runs_out_leaves_the_stream_to_the_old_path(monkeypatch):
    """More than 32 output bytes per compressed byte: run A's pool (sized by the compressed bytes) runs out, the spans
    report it, the stream is decoded by k_inflate_par / k_inflate as before -- same bytes, same status."""
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "2000")
    eng = engine()
    rng = np.random.default_rng(23)
    # long runs between short noisy stretches: about 300:1, several blocks
    parts = []
    for i in range(40):
        parts.append(bytes([i & 255]) * 250_000)
        parts.append(rng.integers(0, 256, 600, dtype=np.uint8).tobytes())
    data = b"".join(parts)
    for container, comp in ((2, pyzlib.compress(data, 6)), (1, O.compress(data, 1, 6))):
        assert 2000 <= len(comp) * 32 < len(data)
        (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many([comp], container, caps=[len(data)]))
        assert st == [0] and used == [len(comp)] and outs[0] == data
        assert "k_span_scan" in prof  # (the path was tried)


@pytest.mark.parametrize("container", [0, 1, 2])
def test_more_long_streams_than_the_first_rule_takes_are_cut_once_each(container, monkeypatch):
    """33 to about 140 long streams (config #5's shape): every stream is cut once and both runs of the second spans go
    in one launch with the first spans (run B's bytes in the pool as well: k_span_fix kind 3); damaged streams among
    them come out of the old path with the reference's names."""
    from flate_amd import synth
    monkeypatch.setenv("FLATE_HIP_INFLATE_SPANS", "60000")
    eng = engine()
    rng = np.random.default_rng(41 + container)
    text = synth.text(synth.SEED_TEXT + 51, 6 << 20).tobytes()
    sil = synth.silesia_like(synth.SEED_SILESIA + 51, 6 << 20).tobytes()
    datas, streams = [], []
    wb = {0: -15, 1: 31, 2: 15}[container]
    for i in range(56):
        src = text if i % 3 else sil
        n = int(rng.integers(300_000, 700_000))
        o = int(rng.integers(0, len(src) - n))
        d = src[o:o + n]
        if i % 4 == 0:
            c = O.compress(d, container, [6, 9, O.HUFFMAN, 4][(i // 4) % 4])
        else:
            co = pyzlib.compressobj(int(rng.integers(1, 10)), pyzlib.DEFLATED, wb)
            c = co.compress(d) + co.flush()
        datas.append(d)
        streams.append(c)
    assert 32 < sum(len(c) >= 32768 for c in streams) <= 140  # (more than 32, at most 0.55 x the CUs: the twin launch)
    (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many(streams, container, caps=[len(d) for d in datas]))
    assert st == [0] * len(streams) and outs == datas and used == [len(c) for c in streams]
    assert "k_inflate_span" in prof, prof
    # the same batch with damage: a flipped bit, a truncation, a slot too small, two members back to back
    bad = list(streams)
    caps = [len(d) + 8 for d in datas]
    m = bytearray(bad[5]); m[len(m) // 3] ^= 0x10; bad[5] = bytes(m)
    bad[9] = bad[9][:len(bad[9]) // 2]
    caps[13] = (len(datas[13]) // 8) * 8 - 8
    bad[17] = bad[17] + bad[18]
    outs, st, used = eng.decompress_many(bad, container, caps=caps)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        name, want, wused = O.decompress(b, container, 0, cap=(cap + 7) & ~7)
        assert O.STATUS[st[i]] == name, (i, O.STATUS[st[i]], name)
        if name == "Ok":
            assert outs[i] == want and used[i] == wused, i


def test_a_batch_of_long_and_short_streams():
    from flate_amd import synth
    eng = engine()
    big = synth.text(synth.SEED_TEXT + 21, 5 << 20).tobytes()
    small = [synth.text(synth.SEED_TEXT + 22 + i, 3000 + 977 * i).tobytes() for i in range(20)]
    streams = [O.compress(big, 1, 6)] + [O.compress(s, 1, 6) for s in small] + [O.compress(big[: 3 << 20], 1, O.HUFFMAN)]
    datas = [big] + small + [big[: 3 << 20]]
    (outs, st, used), prof = _kernels(eng, lambda: eng.decompress_many(streams, 1, caps=[len(d) for d in datas]))
    assert st == [0] * len(streams) and outs == datas and used == [len(s) for s in streams]
    assert "k_inflate_span" in prof
