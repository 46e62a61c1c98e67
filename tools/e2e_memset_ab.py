#!/usr/bin/env python3
"""Host-buffer compress (pinned, 256 MiB text, level 6) with the output slots cleared in line / on the side stream, in ONE process (the link's state differs from process to process)."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
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
def run():
    rc = L.flate_hip_compress_batch(eng._h, p_in.data_ptr(), off.ctypes.data, k, 0, 6, p_out.data_ptr(), oo.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
    assert rc == 0 and not status.any()
def med(reps=7):
    run(); run(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
for rnd in range(3):
    for v in ("1", "0"):
        os.environ["FLATE_HIP_MEMSET_INLINE"] = v; eng._sync_env()
        t = med()
        print("round %d, FLATE_HIP_MEMSET_INLINE=%s: %.2f ms (%.1f GB/s)" % (rnd, v, t * 1e3, n / t / 1e9), flush=True)
