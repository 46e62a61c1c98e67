#!/usr/bin/env python3
"""One-off: a batch of several thousand structured chunks in ONE pass (so that the back end runs its
many-blocks variants: k_encode_wave) through levels 4-7 against the oracle, every chunk.
Usage: big_batch_sweep.py [seed] [n_chunks]"""
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FLATE_HIP_NO_PIN_MIRROR"] = "1"  # (the pinned mirrors would cut the batch into sub-batches)
import numpy as np

import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _fuzz_input

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4500
eng = Engine(0)
rng = np.random.default_rng(seed)
pool = [_fuzz_input(int(rng.integers(1, 1 << 30))) for _ in range(40)]
datas = []
for i in range(n):
    d = pool[int(rng.integers(0, len(pool)))]
    ln = int(rng.integers(0, 65536)) if i % 7 == 0 else int(rng.integers(0, 6000))
    o = int(rng.integers(0, max(1, len(d) - ln)))
    datas.append(d[o:o + ln])
bad = 0
for mode in (4, 5, 6, 7):
    container = int(rng.integers(0, 3))
    outs, st = eng.compress_many(datas, container, mode)
    for i, (d, got, s) in enumerate(zip(datas, outs, st)):
        if s != 0 or got != O.compress(d, container, mode):
            bad += 1
            if bad < 5:
                print("MISMATCH chunk", i, "mode", mode, "container", container, "len", len(d), "status", s)
    back, st2, _ = eng.decompress_many(outs, container, 0, [len(d) + 8 for d in datas])
    if st2 != [0] * len(datas) or back != datas:
        bad += 1
        print("INFLATE MISMATCH mode", mode)
    print("mode", mode, "container", container, "done, mismatches so far:", bad, flush=True)
print("BIG BATCH SWEEP", "OK" if bad == 0 else "FAILED", bad)
sys.exit(1 if bad else 0)
