#!/usr/bin/env python3
"""Shader-clock phases of the block planner (workgroup 0, huffman-only block); build with EXTRA=-DFL_PLAN_PROF."""
import sys, os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
eng = Engine(0)
data = synth.text(synth.SEED_TEXT, 65535 * 64).tobytes()
for rep in range(2):
    eng.compress_many([data], 1, 1)
t = eng.phase_cycles().astype(np.int64)
names = ["start -> lit sort done", "bit_counts (lit)", "assign codes (lit)", "generate_codegen", "codegen code", "sizes + header"]
for i, n in enumerate(names):
    print("%-28s %9d cycles" % (n, t[49 + i] - t[48 + i]))
print("%-28s %9d cycles" % ("total", t[54] - t[48]))
