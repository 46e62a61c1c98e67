#!/usr/bin/env python3
"""Counters of k_lz_walk (WK_PROF build) under load: n chunks of an input kind at a level, one pass.
usage: FLATE_HIP_LIB=flate_amd/lib/var/lib_wkprof.so python tools/walk_load_probe.py kind level [n_chunks]"""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from flate_amd import Engine, synth
kind, level = sys.argv[1], int(sys.argv[2]); nc = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
n = nc * 65535
if kind == "tar": data = synth.tar_like(synth.SEED_TAR, n)
elif kind == "text": data = synth.text(synth.SEED_TEXT, n)
elif kind == "records": data = synth._records(4242, n)
else:
    data = np.zeros(n, dtype=np.uint8); k = n // 97 + 1
    where = (synth.splitmix64(4242, k) % np.uint64(n)).astype(np.int64)
    data[where] = (synth.splitmix64(4243, k) & np.uint64(0xFF)).astype(np.uint8)
data = data.tobytes()
chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
eng = Engine(0)
eng.compress_many(chunks[:8], 0, level)
t0 = eng.phase_cycles().astype(np.int64)
t = time.time(); eng.compress_many(chunks, 0, level); dt = time.time() - t
c = eng.phase_cycles().astype(np.int64) - t0
nw = max(int(c[42]), 1)
print("%s L%d %d chunks: call %.1f ms; per wave: %.0f us, trips %.0f (max %d), walk steps %.0f, runs %.0f, measure iterations %.0f, moves %.0f; rounds per chunk %.1f; per wave us: path following %.0f, trip loops %.0f (round 0: %.0f)"
      % (kind, level, nc, dt * 1e3, c[41] / nw / 2100.0, c[40] / nw, c[51], c[49] / nw, c[47] / nw, c[48] / nw, c[50] / nw, c[43] / nc, c[52] / nw / 2100.0, c[53] / nw / 2100.0, c[54] / nw / 2100.0))
print("gathers per wave (all lanes): %.0f, rank checks %.0f; per byte of input: %.2f gathers" % (c[55] / nw / 64, c[56] / nw / 64, c[55] / 64 / (nc * 65535.0)))
