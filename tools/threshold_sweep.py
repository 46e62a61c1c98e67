#!/usr/bin/env python3
"""One-off sweep of the match finder's THRESHOLDS (deflate.zig:233-266, 154-205): inputs made so that the candidates of a call lie
exactly at, one before and one behind the chain budget (`chain`, a quarter of it from `good` bytes in hand), match exactly `nice`,
`lazy`, `good` bytes or one less / one more, with a longer candidate behind them and better matches at the next positions.  Random
data does not put candidates there; the oracle says what the reference makes of each input, the GPU has to agree -- as chunks and as
whole streams (the case at a window's edge too).  usage: threshold_sweep.py [seed] [cases]"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
eng = Engine(0)
LV = {4: (4, 4, 16, 16), 5: (8, 16, 32, 32), 6: (8, 16, 128, 128), 7: (8, 32, 128, 256), 8: (32, 128, 258, 1024), 9: (32, 258, 258, 4096)}  # good, lazy, nice, chain


def junk(n):
    return rng.integers(128, 256, n, dtype=np.uint8).tobytes()


def near(v):
    return max(3, int(v) + int(rng.integers(-2, 3)))


def make(level, total):
    good, lazy, nice, chain = LV[level]
    S = rng.integers(0, 64, 600, dtype=np.uint8).tobytes()
    copies = []  # farthest first
    # a long candidate, far
    copies.append(S[:near(rng.choice([nice + 4, 258, lazy + 3, 40]))])
    # fillers that share the first 4 (or good + 1) bytes: around the budget
    share = int(rng.choice([4, good, good + 1, 6]))
    budget = int(rng.choice([chain, chain // 4, chain // 4 + 1, chain // 2]))
    k = max(0, budget + int(rng.integers(-4, 3)))
    k = min(k, (30000 - 2000) // (share + 3))
    fill = [S[:share] + junk(3)[: 1 + int(rng.integers(0, 3))] for _ in range(k)]
    # candidates at the thresholds, nearest
    nearc = [S[:near(rng.choice([good, lazy, nice, good - 1, lazy - 1, nice - 1, 5, 7]))] for _ in range(int(rng.integers(0, 4)))]
    # better matches at the next positions (lazy evaluation)
    nextc = [S[o:o + near(rng.choice([good, lazy, nice, 9, 33, 258]))] for o in (1, 2, 3) if rng.random() < 0.5]
    parts = copies + fill + nearc
    order = list(range(len(nextc)))
    body = bytearray()
    for pc in nextc:
        body += pc + junk(5)
    for pc in parts:
        body += pc + junk(2 + int(rng.integers(0, 3)))
    body = bytes(body)
    lead = junk(max(0, total - len(body) - len(S) - 200))
    return lead + body + S + junk(200 - int(rng.integers(0, 150)))


bad = 0
for i in range(cases):
    for level in (4, 5, 6, 7, 8, 9):
        good, lazy, nice, chain = LV[level]
        total = int(rng.choice([20000, 50000, 65535, 65535]))
        d = make(level, total)[-65535:]
        # as a chunk; as a stream with the target call in the first window's interior, at its edge, in the second window
        s1 = junk(int(rng.integers(0, 400))) + d + junk(int(rng.integers(70000, 90000)))
        edge = 65274 + int(rng.integers(-300, 20)) - (len(d) - 200)
        s2 = (junk(max(0, edge)) + d + junk(80000)) if edge > 0 else s1
        s3 = junk(32768 + int(rng.integers(0, 3000))) + s2
        datas = [d, s1, s2, s3]
        c = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, c, level)
        for j, (x, o, s) in enumerate(zip(datas, outs, st)):
            if s != 0 or o != O.compress(x, c, level):
                bad += 1
                print("THRESHOLD MISMATCH case", i, "level", level, "input", j, "len", len(x), "container", c, "status", s, flush=True)
    if i % 10 == 9:
        print("case", i, "done, mismatches so far:", bad, flush=True)
print("THRESHOLD SWEEP", "FAILED" if bad else "OK", bad)
