#!/usr/bin/env python3
"""Batched gunzip probe: N members of M KiB each (made by the GPU compressor, whole-stream gzip
level 6), inflated in one call; prints the time per call.  Usage: inflate_probe.py [N] [KiB] [reps]"""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flate_amd import Engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eng = Engine(0)
data = synth.silesia_like(synth.SEED_SILESIA, n * kib * 1024) if os.environ.get("PROBE_SILESIA") else \
    synth.text(synth.SEED_TEXT, n * kib * 1024)
blob = data.tobytes()
members = [blob[i * kib * 1024:(i + 1) * kib * 1024] for i in range(n)]
gz, st = eng.compress_many(members, 1, 6)
assert st == [0] * n
caps = [kib * 1024] * n
t = []
for _ in range(reps):
    t0 = time.perf_counter()
    dec, dst, _c = eng.decompress_many(gz, 1, 0, caps)
    t.append(time.perf_counter() - t0)
assert dst == [0] * n and all(d == m for d, m in zip(dec, members))
# device-resident timing
dev = torch.device("cuda:0")
comp = torch.from_numpy(np.frombuffer(b"".join(gz), dtype=np.uint8).copy()).to(dev)
off = np.zeros(n + 1, dtype=np.int64); np.cumsum([len(g) for g in gz], out=off[1:])
ooff = np.arange(n + 1, dtype=np.int64) * kib * 1024
d_off = torch.from_numpy(off).to(dev); d_ooff = torch.from_numpy(ooff).to(dev)
out = torch.empty(n * kib * 1024 + 8, dtype=torch.uint8, device=dev)
olen = torch.zeros(n, dtype=torch.int64, device=dev); ost = torch.zeros(n, dtype=torch.int32, device=dev)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.decompress_device(comp.data_ptr(), d_off.data_ptr(), n, 1, 0, out.data_ptr(), d_ooff.data_ptr(), olen.data_ptr(), ost.data_ptr())
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
assert int(ost.abs().sum()) == 0
ms = min(ts) * 1e3
print("inflate %d x %d KiB: %.2f ms device-resident (%.1f MB/s), host path %.1f ms; ratio %.3f" %
      (n, kib, ms, n * kib * 1024 / ms / 1e3, min(t) * 1e3, sum(len(g) for g in gz) / (n * kib * 1024)))
if os.environ.get("PROBE_COUNTS"):
    pc = eng.phase_cycles()
    print("fast rounds", int(pc[40]), "rounds ending in slow", int(pc[41]), "slow symbols", int(pc[42]),
          "cycles: prologue", int(pc[44]), "walk", int(pc[45]), "match copies", int(pc[46]), "n matches", int(pc[47]))
