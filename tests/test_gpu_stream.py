"""GPU parity tests of the whole-stream path (levels 4..9, inputs longer than 65535 bytes):
one deflate stream per input, byte-identical to the reference's sliding-window compressor
(deflate.zig:304-321, SlidingWindow.zig:36-44, Lookup.zig:43-51) as restated by the oracle."""
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from gpu_util import engine

pytestmark = pytest.mark.gpu

WBITS = {0: -15, 1: 31, 2: 15}
ZONE = 65536 - 262  # first buffer offset the tokenizer leaves for after the slide


def _cases():
    from flate_amd import synth
    rng = np.random.default_rng(77)
    text = synth.text(synth.SEED_TEXT + 5, 1 << 20).tobytes()
    rnd = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    noise4 = rng.integers(97, 101, 300000, dtype=np.uint8).tobytes()
    blk = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    cases = {
        "zeros65536": bytes(65536),
        "zeros65537": bytes(65537),
        "zeros98304": bytes(98304),
        "zeros200k": bytes(200000),
        # one repeated byte: windows that k_lz_sort recognises and k_lz_match answers directly, and the
        # windows around the place where the run ends (inside the tile, in its 288-byte lookahead, after it)
        "ff150k": b"\xff" * 150000,
        "zeros_then_text": bytes(65536 + 100) + text[:50000],
        "zeros_then_x_in_lookahead": bytes(65536 + 287) + b"x" + bytes(40000),
        "zeros_then_x_after_lookahead": bytes(65536 + 288) + b"x" + bytes(70000),
        "text_then_zeros": text[:40000] + bytes(200000),
        "text65536": text[:65536],
        "text_zone": text[:ZONE + 40],
        "text98303": text[:98303],
        "text98304": text[:98304],
        "text98305": text[:98305],
        "text131072": text[:131072],
        "text300k": text[7:300007],
        "text1m": text,
        "rand100k": rnd[:100000],          # stored blocks; raw slice lost after slides
        "rand65536": rnd[:65536],
        "rand300k": rnd,
        "noise4": noise4,                   # ~1 token per 2 bytes: block ends anywhere
        "period32768": blk * 5,             # matches at distance exactly 32768 across slides
        "period32767": (blk[:32767] * 6)[:190000],
        "period32769": ((blk + b"x") * 6)[:190000],
        "period16": (bytes(range(16)) * 20000)[:250000],
        "mix": text[:70000] + rnd[:50000] + bytes(40000) + text[1000:90000] + noise4[:60000],
        "sparse": bytes(b if (i % 97 == 0) else 0
                        for i, b in enumerate(rng.integers(0, 256, 150000, dtype=np.uint8))),
    }
    # lengths around every place where the window fills up or the tokenizer stops for a slide
    for k in (0, 1):
        for d in (-1, 0, 1):
            n = 65536 + 32768 * k + d
            cases["text_fill%d%+d" % (k, d)] = text[333:333 + n]
            n = ZONE + 32768 * k + d
            cases["noise_zone%d%+d" % (k, d)] = noise4[:n] if n > 65535 else noise4[:65536 + k + d + 1]
    return cases


CASES = _cases()


def test_stream_tokens_match_oracle():
    eng = engine()
    # debug_tokens sees the last pass only: keep to inputs that all take the whole-stream path
    names = [n for n in CASES if len(CASES[n]) > 65535]
    for level in (4, 6, 9):
        outs, st = eng.compress_many([CASES[n] for n in names], O.RAW, level)
        assert st == [0] * len(names)
        for i, n in enumerate(names):
            want = O.tokenize(CASES[n], level)
            got = eng.debug_tokens(i)
            assert len(got) == len(want), (n, level, len(got), len(want))
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (n, level, int(bad[0]), O.tok_decode(got[bad[0]]), O.tok_decode(want[bad[0]]))


@pytest.mark.parametrize("level", [4, 5, 6, 7, 8, 9])
def test_stream_bytes_match_oracle(level):
    eng = engine()
    names = [n for n in CASES if level in (6, 9) or not n.startswith(("text_fill", "noise_zone"))]
    outs, st = eng.compress_many([CASES[n] for n in names], O.RAW, level)
    assert st == [0] * len(names)
    for n, got in zip(names, outs):
        want = O.compress(CASES[n], O.RAW, level)
        assert got == want, (n, level, len(got), len(want))
        assert pyzlib.decompress(got, -15) == CASES[n]


@pytest.mark.parametrize("container", [1, 2])
def test_stream_containers(container):
    eng = engine()
    names = ["text300k", "rand100k", "zeros200k", "mix", "period32768"]
    outs, st = eng.compress_many([CASES[n] for n in names], container, 6)
    assert st == [0] * len(names)
    for n, got in zip(names, outs):
        assert got == O.compress(CASES[n], container, 6), n
        assert pyzlib.decompress(got, WBITS[container]) == CASES[n]


def test_mixed_batch_of_chunks_and_streams():
    # short inputs (chunk path) and long ones (whole-stream path) interleaved in one call
    eng = engine()
    datas = [b"", CASES["text131072"], b"abc" * 100, CASES["text98304"][:65535], CASES["rand100k"], bytes(70000),
             CASES["text300k"][:4000]]
    for container in (0, 1):
        outs, st = eng.compress_many(datas, container, 6)
        assert st == [0] * len(datas)
        for d, got in zip(datas, outs):
            assert got == O.compress(d, container, 6), len(d)


def test_stream_roundtrip_on_gpu_inflate():
    eng = engine()
    names = ["text1m", "mix", "rand300k"]
    outs, st = eng.compress_many([CASES[n] for n in names], 1, 6)
    assert st == [0] * len(names)
    dec, dst, _ = eng.decompress_many(outs, 1, 0, [len(CASES[n]) for n in names])
    assert dst == [0] * len(names)
    for n, d in zip(names, dec):
        assert d == CASES[n]


def test_big_single_stream_roundtrip():
    # 64 MiB in one stream: too long for the oracle in a unit test; inflate must give it back
    # and the result must not depend on how many tiles one launch takes
    import os
    eng = engine()
    from flate_amd import synth
    data = synth.text(synth.SEED_TEXT + 9, 64 << 20).tobytes()
    outs, st = eng.compress_many([data], 1, 6)
    assert st == [0]
    assert pyzlib.decompress(outs[0], 31) == data
    head = data[:3 << 20]
    a, st = eng.compress_many([head], 0, 6)
    os.environ["FLATE_HIP_MAX_PASS_CHUNKS"] = "7"
    try:
        b, st2 = eng.compress_many([head], 0, 6)
    finally:
        del os.environ["FLATE_HIP_MAX_PASS_CHUNKS"]
    assert st == [0] and st2 == [0] and a[0] == b[0]
    assert a[0] == O.compress(head, 0, 6)


def _fuzz_input(seed):
    """Long inputs built from literal runs and copies whose distances cluster around the
    window size, so that candidates sit on both sides of every slide boundary."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(65536, 400000))
    out = bytearray(rng.integers(0, 256, int(rng.integers(300, 40000)), dtype=np.uint8).tobytes())
    while len(out) < n:
        kind = int(rng.integers(0, 6))
        if kind == 0:
            out += rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes()
        elif kind == 1:
            out += bytes(int(rng.integers(1, 2000)))
        elif kind == 2:
            out += rng.integers(97, 100, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
        else:
            d = int(rng.choice([32768, 32767, 32769, 32768 - 262, 32506, 32505, 65536, 1, 4, 300,
                                int(rng.integers(1, 33000)), 32768 - int(rng.integers(0, 600))]))
            d = min(d, len(out))
            ln = int(rng.integers(4, 700))
            for _ in range(ln):
                out.append(out[-d])
    return bytes(out[:n])


def test_stream_fuzz_against_oracle():
    eng = engine()
    datas = [_fuzz_input(1000 + i) for i in range(24)]
    for level in (4, 6, 8, 9):
        outs, st = eng.compress_many(datas, O.RAW, level)
        assert st == [0] * len(datas)
        for i, (d, got) in enumerate(zip(datas, outs)):
            if got != O.compress(d, O.RAW, level):
                want = O.tokenize(d, level)
                toks = eng.debug_tokens(i)
                m = min(len(toks), len(want))
                bad = np.nonzero(toks[:m] != want[:m])[0]
                first = int(bad[0]) if bad.size else m
                raise AssertionError((i, level, len(d), "first differing token", first,
                                      O.tok_decode(toks[first]) if first < len(toks) else None,
                                      O.tok_decode(want[first]) if first < len(want) else None))


def test_config1_vector_on_gpu():
    # BASELINE.json configs[0]: gzip level 6 of a 4 MiB all-zero buffer (SURVEY.md 8c vector)
    import hashlib
    eng = engine()
    z = bytes(4 << 20)
    outs, st = eng.compress_many([z], O.GZIP, 6)
    assert st == [0] and len(outs[0]) == 4098
    assert hashlib.sha256(outs[0]).hexdigest() == "ead941ecab79abb0d46919b9bfcb699490a1d47de07ae41b3d97a0f4f3d10b06"
    for level in (4, 9):
        o, st = eng.compress_many([z], O.GZIP, level)
        assert st == [0] and o[0][10:-8] == outs[0][10:-8]


def test_stream_passes_are_split_by_bytes():
    # FLATE_HIP_MAX_STREAM_PASS_MIB bounds the scratch of one whole-stream pass: same bytes either way
    import os
    eng = engine()
    names = ["text300k", "mix", "rand300k", "text1m", "zeros200k"]
    datas = [CASES[n] for n in names]
    a, st = eng.compress_many(datas, 1, 6)
    os.environ["FLATE_HIP_MAX_STREAM_PASS_MIB"] = "1"
    try:
        b, st2 = eng.compress_many(datas, 1, 6)
    finally:
        del os.environ["FLATE_HIP_MAX_STREAM_PASS_MIB"]
    assert st == [0] * 5 and st2 == [0] * 5 and a == b


def test_long_stream_takes_the_grouped_stitch():
    # more than 256 segments in one piece: the segment walk goes over groups (k_st_stitch_a/b/c)
    eng = engine()
    from flate_amd import synth
    data = synth.text(synth.SEED_TEXT + 21, 12 << 20).tobytes()
    outs, st = eng.compress_many([data, data[: 9 << 20]], O.RAW, 6)
    assert st == [0, 0]
    assert outs[0] == O.compress(data, O.RAW, 6)
    assert outs[1] == O.compress(data[: 9 << 20], O.RAW, 6)
    got, st = eng.compress_flush(data, [5 << 20, (5 << 20) + 3, 11 << 20], True, O.RAW, 6)
    d = O.Deflate(O.RAW, 6)
    d.write(data[: 5 << 20]); d.flush(); d.write(data[5 << 20:(5 << 20) + 3]); d.flush()
    d.write(data[(5 << 20) + 3:11 << 20]); d.flush(); d.write(data[11 << 20:]); d.finish()
    assert st == 0 and got == d.output()


def test_adversarial_long_streams_match_oracle():
    # long runs of the extreme token shapes: one match per 258 bytes, stored blocks only,
    # every match at the maximum distance, one literal per byte with tiny alphabets
    eng = engine()
    rng = np.random.default_rng(123)
    blk = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    datas = [bytes(24 << 20),
             rng.integers(0, 256, 12 << 20, dtype=np.uint8).tobytes(),
             blk * 384,
             rng.integers(0, 2, 6 << 20, dtype=np.uint8).tobytes()]
    for level in (4, 6):
        outs, st = eng.compress_many(datas, O.GZIP, level)
        assert st == [0] * len(datas)
        for d, got in zip(datas, outs):
            assert got == O.compress(d, O.GZIP, level), (level, len(d))


def test_periodic_streams_do_not_loop_in_the_grouped_fix(monkeypatch):
    # in a stream of one repeated byte (or any data of period 258 k) a group of windows parsed from a guessed entry
    # never falls in step with the true parse, and every fix launch settles only one more group.  The pass gives up after
    # FL_STREAM_FIX_MAX launches and takes the sort / match tiles: same bytes, a bounded number of launches.
    monkeypatch.setenv("FLATE_HIP_STREAM_WINDOWS", "1")
    monkeypatch.setenv("FLATE_HIP_STREAM_GROUP", "4")
    eng = engine()
    datas = [bytes(24 << 20), (b"ab" * 129) * 40000, bytes(3 << 20) + b"x" + bytes(5 << 20)]
    for d in datas:
        eng.profile_enable(True)
        eng.profile_reset()
        outs, st = eng.compress_many([d], O.GZIP, 6)
        prof = eng.profile_read()
        eng.profile_enable(False)
        assert st == [0] and outs[0] == O.compress(d, O.GZIP, 6), len(d)
        assert prof.get("k_lz_parse", (0, 0))[1] <= 4, prof  # the first launch + FL_STREAM_FIX_MAX
    # text does settle in the first fix launch and stays on the windows
    from flate_amd import synth
    d = synth.text(synth.SEED_TEXT, 6 << 20).tobytes()
    eng.profile_enable(True)
    eng.profile_reset()
    outs, st = eng.compress_many([d], O.RAW, 6)
    prof = eng.profile_read()
    eng.profile_enable(False)
    assert st == [0] and outs[0] == O.compress(d, O.RAW, 6)
    assert prof["k_lz_parse"][1] == 2 and "k_lz_match" not in prof, prof


def test_long_streams_of_levels_8_and_9_take_the_windows_of_the_sparse_chain_tokenizer(monkeypatch):
    """Round 6: whole streams of levels 8-9 (no flush points) on k_lz_links / k_lz_walk<true, true> -- the windows of
    k_lz_parse<true> for the sparse-chain tokenizer -- by the library's own estimate: no k_lz_sort / k_lz_match launch, bytes ==
    oracle; groups of windows parsed from a guess and again from their true entry (FLATE_HIP_STREAM_GROUP forces small groups);
    periodic data, where a guessed parse never meets the true one, goes to the tiles after FL_STREAM_FIX_MAX fix launches."""
    from flate_amd import synth
    text = synth.text(synth.SEED_TEXT + 5, 40 << 20).tobytes()
    tar = synth.tar_like(synth.SEED_TAR, 64 << 20).tobytes()
    sil = synth.silesia_like(synth.SEED_SILESIA + 9, 3 << 20).tobytes()
    eng = engine()
    # (the estimate takes the windows when the streams fill the chip: from about 50 MiB a pass)
    for level, datas in ((9, [tar]), (8, [text, tar[:(24 << 20) + 1]])):
        eng.profile_enable(True)
        eng.profile_reset()
        outs, st = eng.compress_many(datas, O.GZIP, level)
        prof = eng.profile_read()
        eng.profile_enable(False)
        assert st == [0] * len(datas)
        for d, o in zip(datas, outs):
            assert o == O.compress(d, O.GZIP, level), (level, len(d))
        assert "k_lz_walk" in prof and "k_lz_match" not in prof and "k_lz_sort" not in prof, prof
    monkeypatch.setenv("FLATE_HIP_STREAM_WINDOWS", "1")
    monkeypatch.setenv("FLATE_HIP_STREAM_GROUP", "2")
    for level in (8, 9):
        datas = [sil, text[:900_000], bytes(400_000) + text[:300_000] + bytes(300_000), (b"ab" * 129) * 3000, tar[:65_536 * 3 + 17]]
        eng.profile_enable(True)
        eng.profile_reset()
        outs, st = eng.compress_many(datas, O.ZLIB, level)
        prof = eng.profile_read()
        eng.profile_enable(False)
        assert st == [0] * len(datas)
        for d, o in zip(datas, outs):
            assert o == O.compress(d, O.ZLIB, level), (level, len(d))
        assert prof["k_lz_walk"][1] <= 4, prof  # the first launch + FL_STREAM_FIX_MAX


def _edge_stream(seed=7, a=65273, steps=6, total=140000, base=34000):
    """A stream whose first window's LAST target (position a < 65274) starts a lazy chain of `steps` improving matches (4, 5, ...
    bytes), the last step finding 258 bytes at a + steps >= 65279: that call is made after the slide, when the window holds the
    whole lookahead again (deflate.zig:304-321) -- 65536 - 65279 = 257 bytes are NOT all a tokenizer may look at there."""
    rng = np.random.default_rng(seed)

    def junk(n):  # bytes 128..255, never part of T
        return rng.integers(128, 256, n, dtype=np.uint8).tobytes()
    T = rng.integers(0, 128, steps + 258 + 8, dtype=np.uint8).tobytes()
    pieces = [T[k:2 * k + 4] for k in range(steps)] + [T[steps:steps + 258]]  # 4 + k bytes that match at a + k; 258 at a + steps
    pre = bytearray(junk(min(a, max(40000, base + 4000 + 50 * steps))))
    pos = base  # (copies at or below 32768 are gone when the calls behind the slide look for them: Lookup.zig:43-51)
    for pc in pieces:
        pre[pos:pos + len(pc)] = pc
        pos += len(pc) + 40
    body = bytes(pre) + junk(a - len(pre))
    return (body + T + junk(max(0, total - a - len(T))))[:total]


@pytest.mark.parametrize("windows", ["", "1", "0"])
def test_lazy_chain_across_a_slide_sees_the_whole_lookahead(windows, monkeypatch):
    if windows:
        monkeypatch.setenv("FLATE_HIP_STREAM_WINDOWS", windows)
    eng = engine()
    # (steps, a, length): the 258-byte match at a + steps >= 65279; long lazy chains (levels 8-9: lazy = 128 / 258) that end far behind
    # the last target; streams that end right behind the window, or exactly with a window
    cases = ((6, 65273, 140000), (9, 65273, 140000), (7, 65270, 140000), (30, 65273, 140000), (60, 65250, 200000),
             (6, 65273, 65273 + 6 + 258), (6, 65273, 65273 + 6 + 258 + 9), (12, 65273, 65536 + 32768), (6, 65273, 65536 + 65536),
             (6, 65273, 140000, 32400), (6, 65273, 140000, 32700), (8, 65272, 150000, 20000))  # (the copies around / below the new window's start)
    for steps, a, total, *rest in cases:
        d = _edge_stream(steps=steps, a=a, total=total, base=rest[0] if rest else 34000)
        for level in (5, 6, 7, 8, 9):
            want = O.tokenize(d, level)
            pos, hit = 0, False
            for t in want:
                dd = O.tok_decode(t)
                hit = hit or (dd[0] == "M" and dd[2] == 258 and pos == a + steps)
                pos += dd[2] if dd[0] == "M" else 1
            if steps <= 7 and not rest:
                assert hit, (steps, a, level)  # (the scenario is what the oracle makes of the input; longer chains: levels 8-9 only)
            outs, st = eng.compress_many([d, d[:100000]], O.RAW, level)
            assert st == [0, 0]
            assert outs[0] == O.compress(d, O.RAW, level), (windows, steps, a, total, level)
            assert outs[1] == O.compress(d[:100000], O.RAW, level), (windows, steps, a, total, level)


def test_whole_stream_tokens_match_the_independent_slide_fixtures():
    # tests/golden/slide: inputs of 150-300 KB with token lists from a pure-Python model of the reference that is
    # independent of the oracle (tests/golden/make_slide_fixtures.py): the GPU's whole-stream path against it directly
    import hashlib
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "slide", "fixtures.json")) as f:
        fix = json.load(f)
    eng = engine()
    for key in sorted(fix):
        name, level = key.split("@")
        with open(os.path.join(GOLDEN, "slide", name + ".bin"), "rb") as f:
            data = f.read()
        outs, st = eng.compress_many([data], O.RAW, int(level))
        assert st == [0], key
        toks = eng.debug_tokens(0)
        assert len(toks) == fix[key]["tokens"], key
        assert hashlib.sha256(np.ascontiguousarray(toks, dtype="<u4").tobytes()).hexdigest() == fix[key]["sha256"], key
        assert pyzlib.decompress(outs[0], -15) == data


def test_whole_stream_path_on_the_chunk_tokenizer():
    """Round 5: batches of many long streams (levels 4-7, no flush points) take k_lz_chain / k_lz_parse<true> -- the reference's
    window after every slide handled as a chunk, a workgroup per stream walking its windows in order -- instead of the sort /
    match pair.  The library picks that path by an estimate (many streams); FLATE_HIP_STREAM_WINDOWS=1 forces it for every
    whole-stream pass without flush points: the token lists, the bytes (incl. the slide fixtures written from the Zig sources,
    the adversarial streams, the fuzz inputs) must be the oracle's all the same (its own process: the knob is read once)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    me = os.path.abspath(__file__)
    env = dict(os.environ, FLATE_HIP_STREAM_WINDOWS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", me + "::test_stream_tokens_match_oracle",
                        me + "::test_stream_bytes_match_oracle", me + "::test_stream_containers",
                        me + "::test_mixed_batch_of_chunks_and_streams", me + "::test_stream_fuzz_against_oracle",
                        me + "::test_config1_vector_on_gpu", me + "::test_stream_passes_are_split_by_bytes",
                        me + "::test_adversarial_long_streams_match_oracle",
                        me + "::test_whole_stream_tokens_match_the_independent_slide_fixtures"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_many_long_streams_take_the_chunk_tokenizer_by_default():
    """300 streams of 70-400 KB at level 6 (the estimate picks the new path: many streams, few windows each): bytes == oracle."""
    from flate_amd import synth
    eng = engine()
    rng = np.random.default_rng(5)
    text = synth.text(synth.SEED_TEXT + 21, 8 << 20).tobytes()
    sil = synth.silesia_like(synth.SEED_SILESIA + 3, 8 << 20).tobytes()
    streams = []
    for i in range(300):
        src = text if i % 3 else sil
        n = int(rng.integers(65536, 400_000)) if i % 7 else int(rng.choice([65536, 65537, 98042, 98043, 131072, 131073, 163840]))
        a = int(rng.integers(0, len(src) - n))
        streams.append(src[a:a + n])
    streams[5] = bytes(200_000)                      # one repeated byte: the windows get chains all the same
    streams[6] = bytes(70_000) + text[:100_000]
    outs, st = eng.compress_many(streams, O.GZIP, 6)
    assert st == [0] * len(streams)
    for i in list(range(0, 300, 11)) + [5, 6, 299]:
        assert outs[i] == O.compress(streams[i], O.GZIP, 6), (i, len(streams[i]))
    back, st2, _ = eng.decompress_many(outs, O.GZIP, caps=[len(s) for s in streams])
    assert st2 == [0] * len(streams) and back == streams
