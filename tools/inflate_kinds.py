#!/usr/bin/env python3
"""Inflate (k_inflate, device resident) of 64 MiB of every data kind, compressed at level argv[1] (6) in 65535-byte chunks:
MB/s of output per kind; the result is compared with the input."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
n = 64 << 20
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng_seed = 4242
def sparse(seed, size):
    z = np.zeros(size, dtype=np.uint8); k = size // 97 + 1
    where = (synth.splitmix64(seed, k) % np.uint64(size)).astype(np.int64)
    z[where] = (synth.splitmix64(seed + 1, k) & np.uint64(0xFF)).astype(np.uint8); return z
kinds = {"text": synth.text(rng_seed, n), "records": synth._records(rng_seed, n), "xml": synth._xml(rng_seed, n),
         "random": synth.splitmix64(rng_seed, n // 8).view(np.uint8)[:n].copy(), "sparse zeros": sparse(rng_seed, n),
         "zeros": np.zeros(n, dtype=np.uint8)}
if hasattr(synth, "tar_like"): kinds["tar-like"] = synth.tar_like(rng_seed, n)
dev = torch.device("cuda:0")
for name, data in kinds.items():
    off = synth.split_offsets(n, 65535); k = len(off) - 1
    caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, level) + 7) & ~7 for i in range(k)], dtype=np.uint64)
    oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
    d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off.astype(np.int64)).to(dev); ot = torch.from_numpy(oo.astype(np.int64)).to(dev)
    out = torch.zeros(int(oo[-1]) + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(k, dtype=torch.int64, device=dev); st = torch.zeros(k, dtype=torch.int32, device=dev)
    eng.compress_device(d.data_ptr(), io.data_ptr(), k, 0, level, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    # pack the streams (gather) so that inflate reads them the way the bench does
    lens = ol.cpu().numpy().astype(np.int64); co = np.zeros(k + 1, dtype=np.int64); np.cumsum(lens, out=co[1:])
    comp = torch.empty(int(co[-1]) + 8, dtype=torch.uint8, device=dev)
    oc = out.cpu().numpy(); packed = np.concatenate([oc[int(oo[i]):int(oo[i]) + int(lens[i])] for i in range(k)])
    comp[:len(packed)] = torch.from_numpy(packed).to(dev)
    c_off = torch.from_numpy(co).to(dev)
    dec = torch.empty(n + 8, dtype=torch.uint8, device=dev); dl = torch.zeros(k, dtype=torch.int64, device=dev); ds = torch.zeros(k, dtype=torch.int32, device=dev)
    run = lambda: eng.decompress_device(comp.data_ptr(), c_off.data_ptr(), k, 0, 0, dec.data_ptr(), io.data_ptr(), dl.data_ptr(), ds.data_ptr())
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ok = int(ds.abs().sum()) == 0 and bool(torch.equal(dec[:n], d))
    print("%-13s ratio %.3f  inflate %7.1f MB/s  %6.2f ms  %s" % (name, co[-1] / n, n / min(ts) / 1e6, min(ts) * 1e3, "ok" if ok else "MISMATCH"))
