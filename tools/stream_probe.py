#!/usr/bin/env python3
"""Per-kernel time of the whole-stream compress path: 256 MiB of text as streams of argv[1] KiB (1024), level argv[2] (6)."""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
n = 256 << 20
kib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
data = synth.text(synth.SEED_TEXT, n)
off = synth.split_offsets(n, kib * 1024); k = len(off) - 1
caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, level) + 7) & ~7 for i in range(k)], dtype=np.uint64)
oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off.astype(np.int64)).to(dev); ot = torch.from_numpy(oo.astype(np.int64)).to(dev)
out = torch.empty(int(oo[-1]) + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(k, dtype=torch.int64, device=dev); st = torch.zeros(k, dtype=torch.int32, device=dev)
run = lambda: eng.compress_device(d.data_ptr(), io.data_ptr(), k, 0, level, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
run(); torch.cuda.synchronize(); eng.profile_reset(); eng.profile_enable(True)
for _ in range(3): run()
torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
tot = sum(v[0] for v in prof.values()) / 3
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
print("%d streams of %d KiB, level %d: ratio %.3f  %.1f MB/s  %.2f ms of kernels, %.2f ms wall (%.1f MB/s)" % (k, kib, level, float(ol.sum()) / n, n / tot / 1e3, tot, wall, n / wall / 1e3))
print("  ".join("%s %.2f" % (kk, v[0] / 3) for kk, v in sorted(prof.items(), key=lambda x: -x[1][0])))
