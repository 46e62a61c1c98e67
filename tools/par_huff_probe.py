#!/usr/bin/env python3
"""One long huffman-only gzip stream (config #4's round trip) through the inflate path: which kernel, why redo."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 16 << 20
eng = Engine(0)
data = synth.silesia_like(synth.SEED_SILESIA, n).tobytes()
outs, st = eng.compress_many([data], 1, 1)
comp = outs[0]
print("compressed", len(comp), "status", st)
for cap in (len(data) + 64, len(data)):
    eng.profile_enable(True); eng.profile_reset()
    o, s, c = eng.decompress_many([comp], 1, 0, caps=[cap])
    prof = eng.profile_read()
    print("cap", cap, "status", s, "equal", o[0] == data, {k: round(v[0], 3) for k, v in prof.items()}, "redo reason", int(eng.phase_cycles()[60]))
