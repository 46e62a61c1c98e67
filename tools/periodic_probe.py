#!/usr/bin/env python3
"""Small whole streams of periodic data (zeros, `ab`, counters in text) at level 6: ms per call through the library's own choice of
path and with the sort / match tiles forced (FLATE_HIP_STREAM_WINDOWS=0); bytes == oracle."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import _oracle as O
from flate_amd import Engine
eng = Engine(0)
for name, d in (("zeros 0.2 MiB", bytes(200000)), ("zeros 4 MiB", bytes(4 << 20)), ("abab 1 MiB", b"ab" * (1 << 19)), ("text-ish", (b"the quick brown fox jumps over the lazy dog %d " * 3000) % tuple(range(3000)))):
    for _ in range(3): outs, st = eng.compress_many([d], O.GZIP, 6)
    t0 = time.perf_counter()
    for _ in range(10): outs, st = eng.compress_many([d], O.GZIP, 6)
    print("%-16s W=%s: %.2f ms a call" % (name, os.environ.get("FLATE_HIP_STREAM_WINDOWS", "-"), (time.perf_counter() - t0) * 100), outs[0] == O.compress(d, O.GZIP, 6))
