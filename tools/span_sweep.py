#!/usr/bin/env python3
"""One-off sweep of the span path of the inflater (not part of the test suite): batches of long streams made of mixed
stretches (text, noise, runs, records), from zlib at random levels / strategies and from the library's own compressor;
decoded with the span path on at a random lower bound and with it off: both must agree in status, bytes and consumed
count -- for damaged streams as well -- and undamaged ones must give the input back.
usage: python tools/span_sweep.py [seed=1] [rounds=20] [big|many]   (big: stretches of up to 8 MiB instead of 1 MiB;
many: batches of 34-70 streams -- every stream cut once, both runs in one launch)"""
import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
big = len(sys.argv) > 3 and sys.argv[3] == "big"
many = len(sys.argv) > 3 and sys.argv[3] == "many"
rng = np.random.default_rng(seed)
eng = Engine(0)
text = synth.text(synth.SEED_TEXT + seed, 8 << 20).tobytes()
sil = synth.silesia_like(synth.SEED_SILESIA + seed, 8 << 20).tobytes()


def stretch():
    k = int(rng.integers(0, 6))
    n = int(rng.integers(1, 1 << int(rng.integers(14, 24) if big else rng.integers(8, 21))))
    n = min(n, 7 << 20)
    if k == 0:
        o = int(rng.integers(0, len(text) - n)); return text[o:o + n]
    if k == 1:
        o = int(rng.integers(0, len(sil) - n)); return sil[o:o + n]
    if k == 2:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 3:
        return bytes([int(rng.integers(0, 256))]) * n
    if k == 4:
        per = rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8).tobytes(); return (per * (n // len(per) + 1))[:n]
    z = np.zeros(n, dtype=np.uint8); w = rng.integers(0, n, n // 90 + 1); z[w] = rng.integers(1, 256, len(w)); return z.tobytes()


def make():
    data = b"".join(stretch() for _ in range(int(rng.integers(1, 5 if many else 12))))
    container = int(rng.integers(0, 3))
    if rng.random() < 0.5:
        wb = {0: -15, 1: 31, 2: 15}[container]
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, wb, int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        comp = co.compress(data)
        if rng.random() < 0.3:
            comp += co.flush(zlib.Z_FULL_FLUSH)  # (an empty stored block in the middle of nothing)
        comp += co.flush()
    else:
        mode = int(rng.choice([0, 1, 4, 5, 6, 7, 8, 9]))
        c, st = eng.compress_many([data], container, mode)
        assert st[0] in (0, 102), st
        if st[0] == 102:  # (the reference's own Q1 stream does not inflate to the input: DESIGN.md section 3 -- another one)
            return make()
        comp = c[0]
    return data, comp, container


bad = 0
for rd in range(rounds):
    container = None
    batch = []
    want_n = int(rng.integers(34, 71)) if many else int(rng.integers(1, 7))
    while len(batch) < want_n:
        d, c, k = make()
        if container is None:
            container = k
        if k == container:
            batch.append((d, c))
    streams, caps, truth = [], [], []
    for d, c in batch:
        r = rng.random()
        if r < 0.55:
            streams.append(c); caps.append(len(d) + int(rng.integers(0, 64))); truth.append(d)
        elif r < 0.7:
            m = bytearray(c); m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(m)); caps.append(len(d) + 64); truth.append(None)
        elif r < 0.8:
            streams.append(c[:int(rng.integers(0, len(c)))]); caps.append(len(d) + 64); truth.append(None)
        elif r < 0.9:
            streams.append(c + c); caps.append(len(d) + 64); truth.append(d)
        else:
            streams.append(c); caps.append(max(8, len(d) - int(rng.integers(1, 5000)))); truth.append(None)
    os.environ["FLATE_HIP_INFLATE_SPANS"] = "0"
    eng._sync_env()  # (the library reads its knobs once per handle)
    ref = eng.decompress_many(streams, container, caps=caps)
    bound = str(int(rng.choice([64, 2000, 40000, 131072])))
    os.environ["FLATE_HIP_INFLATE_SPANS"] = bound
    eng._sync_env()  # (the library reads its knobs once per handle)
    got = eng.decompress_many(streams, container, caps=caps)
    for i in range(len(streams)):
        same = got[1][i] == ref[1][i] and (got[1][i] != 0 or (got[0][i] == ref[0][i] and got[2][i] == ref[2][i]))
        right = truth[i] is None or (got[1][i] == 0 and got[0][i] == truth[i])
        if not (same and right):
            bad += 1
            print("MISMATCH round %d stream %d bound %s container %d: status %d / %d, %d bytes in" % (rd, i, bound, container, got[1][i], ref[1][i], len(streams[i])))
    print("round %d: %d streams (%s bytes), bound %s, statuses %s" % (rd, len(streams), [len(s) for s in streams][:8], bound, got[1][:16]), flush=True)
print("mismatches", bad)
