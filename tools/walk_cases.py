#!/usr/bin/env python3
"""Each input of tests/test_gpu_compress.py's case table on its own through compress (levels 4..9), against the
oracle's token list, with the time each call takes -- printed before and after, so that a call that hangs is the last line."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import test_gpu_compress as T
from flate_amd import default_engine
eng = default_engine()
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 4, 5, 7, 8, 9]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for level in levels:
    for name, data in T.CASES.items():
        if only and name not in only:
            continue
        print("level %d %-14s %6d bytes ..." % (level, name, len(data)), end="", flush=True)
        t0 = time.time()
        outs, st = eng.compress_many([data], O.RAW, level)
        dt = time.time() - t0
        want = O.tokenize(data, level)
        got = eng.debug_tokens(0)
        ok = st == [0] and len(got) == len(want) and not np.any(got != want) and outs[0] == O.compress(data, O.RAW, level)
        print(" %8.1f ms  %s" % (dt * 1e3, "ok" if ok else "MISMATCH"), flush=True)
