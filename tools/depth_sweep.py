#!/usr/bin/env python3
"""One-off sweep of the Huffman builder's length limits (15 bits for literals / lengths and distances, 7 for the code-length code:
huffman_encoder.zig): symbol frequencies that grow like Fibonacci numbers or powers of two -- a tree deeper than the limit -- for
literals, for match lengths and distances, and for the code lengths themselves.  GPU == oracle, as chunks and streams, modes 1 and
4-9.  usage: depth_sweep.py [seed] [cases]"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
eng = Engine(0)


from _adversarial import skewed as _skewed, match_skew as _match_skew


def skewed(total, growth, nsym):
    return _skewed(rng, total, growth, nsym)


def match_skew(total, growth):
    return _match_skew(rng, total, growth)


bad = 0
for i in range(cases):
    growth = float(rng.choice([1.618, 2.0, 1.5, 1.3, 3.0]))
    nsym = int(rng.integers(10, 60))
    n = int(rng.choice([3000, 20000, 65535, 65535, 200000]))
    datas = [skewed(n, growth, nsym), match_skew(min(n, 120000), float(rng.choice([1.2, 1.618, 2.0]))),
             skewed(n // 2, growth, nsym) + match_skew(min(n, 60000) // 2 + 4100, 1.618)]
    for mode in (1, 4, 6, 9):
        c = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, c, mode)
        for j, (x, o, s) in enumerate(zip(datas, outs, st)):
            want = O.compress(x, c, mode)
            if s not in (0, 102) or o != want:
                bad += 1
                print("DEPTH MISMATCH case", i, "mode", mode, "input", j, "len", len(x), "growth", growth, "nsym", nsym, "status", s, flush=True)
        back, st2, _ = eng.decompress_many(outs, c, caps=[len(x) + 8 for x in datas])
        for j, (x, b, s) in enumerate(zip(datas, back, st2)):
            w = O.decompress(outs[j], c, 0, cap=(len(x) + 8 + 7) & ~7)
            if O.STATUS[s] != w[0] or (w[0] == "Ok" and b != w[1]):
                bad += 1
                print("DEPTH INFLATE MISMATCH case", i, "mode", mode, "input", j, flush=True)
print("DEPTH SWEEP", "FAILED" if bad else "OK", bad)
