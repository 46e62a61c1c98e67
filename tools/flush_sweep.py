#!/usr/bin/env python3
"""One-off sweep of sync-flush points put where they hurt: inside matches and runs, right before positions whose only candidate lies in
the three bytes a flush keeps out of the hash table, at and around the window's edges (65274 + 32768 k: the slide table), several in
a row, at 0 and at the end; structured inputs of 1 KiB .. 400 KiB; every level; finish or not.  GPU (the windows of levels 4-7, the
tiles of levels 8-9, or what FLATE_HIP_STREAM_WINDOWS says) == the oracle's Deflate object fed the same writes and flushes.
usage: flush_sweep.py [seed] [cases]"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine, synth
from test_gpu_flush import _oracle_stream
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
eng = Engine(0)
text = synth.text(synth.SEED_TEXT + seed, 600000).tobytes()
bad = 0
for i in range(cases):
    n = int(rng.choice([1000, 30000, 65535, 65536, 70000, 140000, 200000, 400000]))
    kind = int(rng.integers(0, 5))
    o = int(rng.integers(0, len(text) - n))
    if kind == 0: d = text[o:o + n]
    elif kind == 1: d = bytes(n)
    elif kind == 2: d = (text[o:o + 97] * (n // 97 + 1))[:n]                       # short period: matches everywhere
    elif kind == 3:                                                                 # runs of zeros between text
        d = b"".join(text[o + k * 500:o + k * 500 + int(rng.integers(5, 400))] + bytes(int(rng.integers(4, 3000))) for k in range(n // 800 + 1))[:n]
    else: d = rng.integers(0, 4, n, dtype=np.uint8).tobytes()                        # tiny alphabet: a token every few bytes
    n = len(d)
    fl = set()
    for _ in range(int(rng.integers(1, 9))):
        r = rng.random()
        if r < 0.3: f = int(rng.integers(0, n + 1))
        elif r < 0.6: f = 65274 + 32768 * int(rng.integers(0, 6)) + int(rng.integers(-6, 7))
        elif r < 0.75: f = 65536 + 32768 * int(rng.integers(0, 6)) + int(rng.integers(-4, 5))
        elif r < 0.85: f = int(rng.choice([0, 1, 2, 3, 4, n, n - 1, n - 2, n - 3, n - 4]))
        else:
            f = int(rng.integers(0, n + 1)); fl.update([f + 1, f + 2, f + 3, f + 4])
        fl.add(f)
    fl = sorted(f for f in fl if 0 <= f <= n)
    finish = bool(rng.random() < 0.7)
    if not finish:
        fl = [f for f in fl if f < n] + [n]
    for level in (4, 5, 6, 7, 8, 9, 0, 1):
        c = int(rng.integers(0, 3))
        got, s = eng.compress_flush(d, fl, finish, c, level)
        want = _oracle_stream(d, fl, finish, c, level)[0]
        if s not in (0, 102) or got != want:
            bad += 1
            print("FLUSH MISMATCH case", i, "kind", kind, "n", n, "level", level, "container", c, "flushes", fl, "finish", finish, "status", s, flush=True)
    if i % 10 == 9:
        print("case", i, "done, mismatches so far:", bad, flush=True)
print("FLUSH SWEEP", "FAILED" if bad else "OK", bad)
