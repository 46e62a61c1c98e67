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
from _adversarial import LV, junk as _junk, threshold_input


def junk(n):
    return _junk(rng, n)


def make(level, total):
    return threshold_input(rng, level, total)


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
