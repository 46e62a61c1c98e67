#!/usr/bin/env python3
"""2 MiB of text with a sync flush every 64 / 7 / 1000 bytes at level 6 (up to 300 000 flush points): ms per call on the sort / match
tiles and by the library's own choice (the windows), bytes == oracle."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import _oracle as O
from flate_amd import Engine, synth
from test_gpu_flush import _oracle_stream
eng = Engine(0)
d = synth.text(synth.SEED_TEXT, 2 << 20).tobytes()
for every in (64, 7, 1000):
    fl = list(range(every, len(d), every))
    want = _oracle_stream(d, fl, True, 0, 6)[0]
    for w in ("0", None):
        if w is None: os.environ.pop("FLATE_HIP_STREAM_WINDOWS", None)
        else: os.environ["FLATE_HIP_STREAM_WINDOWS"] = w
        got, st = eng.compress_flush(d, fl, True, 0, 6)
        t0 = time.perf_counter()
        for _ in range(3): got, st = eng.compress_flush(d, fl, True, 0, 6)
        print("flush every %d bytes (%d points), W=%s: %.1f ms a call, equal oracle %s, status %d" % (every, len(fl), w or "-", (time.perf_counter() - t0) / 3 * 1e3, got == want, st), flush=True)
