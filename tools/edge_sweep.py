#!/usr/bin/env python3
"""One-off sweep of the slide edge (tests/test_gpu_stream.py::_edge_stream): lazy chains of improving matches that start at one of a
window's last targets and end in a long match found after the slide; random start, chain length, window number and stream length;
levels 4-9 through the library's own choice of path.  usage: edge_sweep.py [seed] [cases]"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _edge_stream
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
eng = Engine(0)
bad = 0
chunk_mode = len(sys.argv) > 3 and sys.argv[3] == "chunk"
for i in range(cases if chunk_mode else 0):
    # the same chain at the end of sub-pass A's targets of k_lz_parse (49152), at a segment's end, at the chunk's end
    steps = int(rng.integers(1, 60))
    a = int(rng.choice([int(rng.integers(49090, 49153)), 48 * int(rng.integers(400, 1000)) - int(rng.integers(0, 8)), 65535 - 258 - steps - int(rng.integers(0, 40))]))
    total = int(rng.choice([65535, int(rng.integers(a + steps + 258, 65536))]))
    span = steps * steps // 2 + 44 * steps + 400  # what the copies take
    base = int(rng.integers(500, a - span - 100))
    d = _edge_stream(seed=int(rng.integers(1, 1 << 30)), steps=steps, a=a, total=total, base=base)
    datas = [d[:65535], d[:min(65535, len(d))][:-1], d[3:65535]]
    for level in (4, 5, 6, 7, 8, 9):
        c = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, c, level)
        for x, o, s in zip(datas, outs, st):
            if s != 0 or o != O.compress(x, c, level):
                bad += 1
                print("EDGE MISMATCH (chunk) case", i, "steps", steps, "a", a, "len", len(x), "level", level, "container", c, "status", s, flush=True)
for i in range(0 if chunk_mode else cases):
    steps = int(rng.integers(1, 70))
    a = int(rng.integers(65150, 65274))
    k = int(rng.integers(0, 3))  # the window whose edge it is
    total = int(rng.choice([a + steps + 258, a + steps + 258 + int(rng.integers(1, 400)), 65536 + 32768 * (k + 1), int(rng.integers(140000, 260000))]))
    base = int(rng.choice([34000, int(rng.integers(500, 40000)), int(rng.integers(32000, 33200))]))
    d = _edge_stream(seed=int(rng.integers(1, 1 << 30)), steps=steps, a=a, total=max(total, a + steps + 258), base=base)
    if k:  # the same edge one or two windows later: junk of bytes that T does not hold in front
        d = rng.integers(128, 256, 32768 * k, dtype=np.uint8).tobytes() + d
    datas = [d, d[:len(d) - int(rng.integers(0, 300))]]
    for level in (4, 5, 6, 7, 8, 9):
        c = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, c, level)
        for x, o, s in zip(datas, outs, st):
            if s != 0 or o != O.compress(x, c, level):
                bad += 1
                print("EDGE MISMATCH case", i, "steps", steps, "a", a, "k", k, "len", len(x), "level", level, "container", c, "status", s, flush=True)
print("EDGE SWEEP", "FAILED" if bad else "OK", bad)
