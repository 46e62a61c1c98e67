#!/usr/bin/env python3
"""Wide parity sweep of the chunk path on run-heavy inputs (what the run logic, the offset mode and the closed-form
segments of kernels_walk.h and the deferred re-parse of kernels_parse.h must get exactly right): random mixtures of
runs of one byte (every length), junk, repeated blocks, sparse zeros and text, levels 4-9, against the oracle.
Usage: runny_sweep.py [seed] [rounds]   (one round = 48 inputs x 6 levels)"""
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import _oracle as O
from flate_amd import Engine, synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = Engine(0)
rng = np.random.default_rng(seed)
text = synth.text(synth.SEED_TEXT + seed, 1 << 20).tobytes()


def piece():
    kind = int(rng.integers(0, 7))
    if kind == 0:  # a run
        return bytes([int(rng.integers(0, 4))]) * int(rng.choice([1, 3, 4, 5, 7, 8, 9, 63, 64, 65, 257, 258, 259, 300, 322, 323, 324, 600, 4096, 9000]))
    if kind == 1:  # junk
        return rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
    if kind == 2:  # text
        a = int(rng.integers(0, len(text) - 5000))
        return text[a:a + int(rng.integers(1, 5000))]
    if kind == 3:  # a short period
        k = int(rng.integers(2, 9))
        return (rng.integers(0, 4, k, dtype=np.uint8).tobytes() * 400)[:int(rng.integers(1, 1500))]
    if kind == 4:  # sparse zeros
        n = int(rng.integers(50, 3000))
        z = np.zeros(n, dtype=np.uint8)
        z[rng.integers(0, n, max(1, n // int(rng.integers(20, 200))))] = rng.integers(1, 256)
        return z.tobytes()
    if kind == 5:  # a block seen before comes back (distances on both sides of 32768 too)
        return b"@BLOCK@"
    return bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 70))


bad = 0
for rd in range(rounds):
    datas = []
    for i in range(48):
        n = int(rng.choice([65535, 65535, 40000, 12345, 700, int(rng.integers(0, 65536))]))
        parts, k = [], 0
        block = rng.integers(0, 8, int(rng.integers(8, 600)), dtype=np.uint8).tobytes()
        while k < n:
            p = piece()
            if p == b"@BLOCK@":
                p = block
            parts.append(p)
            k += len(p)
        datas.append(b"".join(parts)[:n])
    for mode in (4, 5, 6, 7, 8, 9):
        outs, st = eng.compress_many(datas, 0, mode)
        for i, (d, got, s) in enumerate(zip(datas, outs, st)):
            if s != 0 or got != O.compress(d, 0, mode):
                bad += 1
                print("MISMATCH round %d input %d level %d (%d bytes, status %d)" % (rd, i, mode, len(d), s), flush=True)
                with open(os.path.join(ROOT, "gpurun_out", "runny_bad_%d_%d_%d.bin" % (seed, rd, i)), "wb") as f:
                    f.write(d)
    print("round", rd, "done, mismatches so far:", bad, flush=True)
print("RUNNY SWEEP", "FAILED" if bad else "OK", bad)
