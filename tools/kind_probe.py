#!/usr/bin/env python3
"""Per-kernel time of compress (level 6, or argv[1]) for each segment type of the Silesia-like mix (128 MiB each)."""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
n = 128 << 20
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng_seed = 4242
def sparse(seed, size):
    z = np.zeros(size, dtype=np.uint8); k = size // 97 + 1
    where = (synth.splitmix64(seed, k) % np.uint64(size)).astype(np.int64)
    z[where] = (synth.splitmix64(seed + 1, k) & np.uint64(0xFF)).astype(np.uint8); return z
tar = synth.tar_like(rng_seed, n) if hasattr(synth, "tar_like") else None
kinds = {"text": synth.text(rng_seed, n), "records": synth._records(rng_seed, n), "xml": synth._xml(rng_seed, n),
         "random": synth.splitmix64(rng_seed, n // 8).view(np.uint8)[:n].copy(), "sparse zeros": sparse(rng_seed, n)}
if tar is not None: kinds["tar-like"] = tar
dev = torch.device("cuda:0")
for name, data in kinds.items():
    off = synth.split_offsets(n, 65535); k = len(off) - 1
    caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, level) + 7) & ~7 for i in range(k)], dtype=np.uint64)
    oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
    d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off.astype(np.int64)).to(dev); ot = torch.from_numpy(oo.astype(np.int64)).to(dev)
    out = torch.empty(int(oo[-1]) + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(k, dtype=torch.int64, device=dev); st = torch.zeros(k, dtype=torch.int32, device=dev)
    run = lambda: eng.compress_device(d.data_ptr(), io.data_ptr(), k, 0, level, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
    run(); torch.cuda.synchronize(); eng.profile_reset(); eng.profile_enable(True)
    for _ in range(3): run()
    torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
    tot = sum(v[0] for v in prof.values()) / 3
    print("%-13s ratio %.3f  %6.1f MB/s  " % (name, float(ol.sum()) / n, n / tot / 1e3) +
          "  ".join("%s %.2f" % (kk.replace("k_lz_", "").replace("k_", ""), v[0] / 3) for kk, v in sorted(prof.items())))
