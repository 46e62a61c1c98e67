#!/usr/bin/env python3
"""Debug: one input through compress with the WK_PROF build, counters of k_lz_walk printed.
usage: FLATE_HIP_LIB=flate_amd/lib/var/lib_wkprof.so python tools/walk_dbg.py level name [size]"""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import default_engine
eng = default_engine()
level = int(sys.argv[1]); name = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 65535
if os.path.exists(name):   # a file: chunk number n of it (65535-byte chunks)
    with open(name, "rb") as f:
        f.seek(65535 * n); data = f.read(65535)
else:
    data = {"x_then_zeros": b"x" + bytes(n - 1), "zeros_then_x": bytes(n - 1) + b"x", "zeros_x_zeros": bytes(n // 2) + b"x" + bytes(n - n // 2 - 1)}[name]
t0 = eng.phase_cycles().astype(np.int64)
print("level %d %s %d bytes ..." % (level, name, n), end="", flush=True)
t = time.time()
outs, st = eng.compress_many([data], O.RAW, level)
dt = time.time() - t
c = eng.phase_cycles().astype(np.int64) - t0
ok = st == [0] and outs[0] == O.compress(data, O.RAW, level)
print(" %.1f ms %s  kernel %.0f us/wave  waves %d trips/wave %.0f (max %d) rounds %d  per wave: walk steps %.0f runs %.0f measures %.0f moves %.0f  caps %d %d" % (dt * 1e3, "ok" if ok else "MISMATCH", c[41] / max(c[42], 1) / 2100.0, c[42], c[40] / max(c[42], 1), c[51], c[43], c[49] / max(c[42], 1), c[47] / max(c[42], 1), c[48] / max(c[42], 1), c[50] / max(c[42], 1), c[45], c[46]), flush=True)
