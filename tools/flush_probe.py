#!/usr/bin/env python3
"""One stream of argv[1] MiB of text with a sync flush every argv[2] KiB at level argv[3]: ms per call (host buffers) and the kernels,
by the library's own choice of path and with the sort / match tiles forced."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
mib = float(sys.argv[1]); every = int(sys.argv[2]) * 1024; level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = int(mib * (1 << 20))
d = synth.text(synth.SEED_TEXT, n).tobytes()
fl = list(range(every, n, every))
eng = Engine(0)
for w in ("0", None):
    if w is None: os.environ.pop("FLATE_HIP_STREAM_WINDOWS", None)
    else: os.environ["FLATE_HIP_STREAM_WINDOWS"] = w
    for _ in range(2): out, st = eng.compress_flush(d, fl, True, 1, level)
    eng.profile_enable(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3): out, st = eng.compress_flush(d, fl, True, 1, level)
    wall = (time.perf_counter() - t0) / 3 * 1e3
    prof = eng.profile_read(); eng.profile_enable(False)
    ks = sum(v[0] for v in prof.values()) / 3
    print("%.0f MiB, a flush every %d KiB (%d), level %d, W=%s: %.2f ms a call, %.2f ms of kernels (%.1f GB/s)  " % (mib, every // 1024, len(fl), level, w or "-", wall, ks, n / ks / 1e6) +
          "  ".join("%s %.2f" % (k.replace("k_lz_", "").replace("k_", ""), v[0] / 3) for k, v in sorted(prof.items(), key=lambda x: -x[1][0])[:4]), len(out), st)
