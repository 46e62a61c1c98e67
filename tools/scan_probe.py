#!/usr/bin/env python3
"""k_span_scan's counters (library built with -DFP_SCAN_PROF): windows, survivors of the cheap tests, headers that
decode, cycles of thread 0 per step.  usage: FLATE_HIP_LIB=... python tools/scan_probe.py [MiB] [mode] [text|silesia]"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
eng = Engine(0)
n = mib << 20
data = (synth.text(synth.SEED_TEXT, n) if kind == "text" else synth.silesia_like(synth.SEED_SILESIA, n)).tobytes()
comps, st = eng.compress_many([data], 1, mode)
t0 = eng.phase_cycles().astype(np.int64)
outs, st, used = eng.decompress_many(comps, 1, caps=[len(data)])
t = eng.phase_cycles().astype(np.int64) - t0
assert st == [0] and outs[0] == data
w = max(1, int(t[10]))
print("windows %d, survivors of step 1 per window %.1f, passes of step 2 per window %.2f" % (t[10], t[11] / w, t[12] / w))
for k, nm in {13: "step 0 (stored headers)", 14: "step 1 (cheap tests, Kraft sum)", 15: "step 2 (a lane per header)", 16: "step 3 (the parser)"}.items():
    print("%-34s %9.0f cycles per window" % (nm, t[k] / w))
