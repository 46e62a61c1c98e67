#!/usr/bin/env python3
"""Pinned host-buffer compress (level 6, 256 MiB text, 65535-byte chunks), eight calls, the time of each: for a rocprofv3
timeline of one process (tools/e2e_timeline.py) -- the call takes 10.4 ms in some processes and 16.6 ms in others."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth, _capi
eng = Engine(0); L = _capi.lib()
n = 256 << 20
data = synth.text(synth.SEED_TEXT, n)
off = synth.split_offsets(n, 65535).astype(np.uint64); k = len(off) - 1
caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, 6) + 7) & ~7 for i in range(k)], dtype=np.uint64)
oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
out_len = np.zeros(k, dtype=np.uint64); status = np.zeros(k, dtype=np.int32)
p_in = torch.from_numpy(data).pin_memory(); p_out = torch.zeros(int(oo[-1]) + 8, dtype=torch.uint8).pin_memory()
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t0 = time.perf_counter()
    rc = L.flate_hip_compress_batch(eng._h, p_in.data_ptr(), off.ctypes.data, k, 0, 6, p_out.data_ptr(), oo.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0 and not status.any()
print("pinned calls, ms:", " ".join("%.2f" % x for x in ts))
