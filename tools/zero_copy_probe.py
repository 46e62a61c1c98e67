#!/usr/bin/env python3
"""Can a kernel write the produced bytes straight into pinned host memory at link speed?  256 MiB of text compressed on the
device, then k_gather_copy with a pinned host buffer as destination, against the DMA copy of the packed bytes."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
n = 256 << 20
dev = torch.device("cuda:0")
data = synth.text(synth.SEED_TEXT, n)
off = synth.split_offsets(n, 65535); k = len(off) - 1
caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, 6) + 7) & ~7 for i in range(k)], dtype=np.uint64)
oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off.astype(np.int64)).to(dev); ot = torch.from_numpy(oo.astype(np.int64)).to(dev)
out = torch.zeros(int(oo[-1]) + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(k, dtype=torch.int64, device=dev); st = torch.zeros(k, dtype=torch.int32, device=dev)
eng.compress_device(d.data_ptr(), io.data_ptr(), k, 0, 6, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
torch.cuda.synchronize()
total = int(ol.sum())
host = torch.empty(int(oo[-1]) + 64, dtype=torch.uint8).pin_memory()
dpk = torch.empty(total + 64, dtype=torch.uint8, device=dev)
doff = torch.zeros(k + 1, dtype=torch.int64, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = t(lambda: eng.gather_streams_device(out.data_ptr(), ot.data_ptr(), ol.data_ptr(), k, dpk.data_ptr(), doff.data_ptr()))
b = t(lambda: eng.gather_streams_device(out.data_ptr(), ot.data_ptr(), ol.data_ptr(), k, host.data_ptr(), doff.data_ptr()))
c = t(lambda: host[:total].copy_(dpk[:total], non_blocking=True))
e = t(lambda: host[:int(oo[-1])].copy_(out[:int(oo[-1])], non_blocking=True))
print("produced %.1f MB of %.1f MB of slots" % (total / 1e6, int(oo[-1]) / 1e6))
print("pack to device memory %.2f ms; pack straight into pinned host memory %.2f ms (%.1f GB/s); DMA of the packed bytes %.2f ms (%.1f GB/s); DMA of the whole slots %.2f ms"
      % (a * 1e3, b * 1e3, total / b / 1e9, c * 1e3, total / c / 1e9, e * 1e3))
ok = bool((host[:total].numpy() == dpk[:total].cpu().numpy()).all())
print("same bytes:", ok)
