#!/usr/bin/env python3
"""Would two halves of the batch on two streams (two engines) fill each other's kernel tails?  Level 6, 1 GiB of the
benchmark text: one engine over the whole batch against two engines over a half each, enqueued together."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
dev = torch.device("cuda", 0)
n = 1 << 30
data = synth.text_torch(synth.SEED_TEXT, n, device=dev)
off = synth.split_offsets(n, 65535); k = len(off) - 1
def mk(eng, lo, hi, stream):
    o = (off[lo:hi + 1] - off[lo]).astype(np.uint64); m = hi - lo
    caps = np.array([(eng.compress_bound(int(o[i + 1] - o[i]), 0, 6) + 7) & ~7 for i in range(m)], dtype=np.uint64)
    oo = np.zeros(m + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
    out = torch.empty(int(oo[-1]) + 8, dtype=torch.uint8, device=dev)
    ol = torch.zeros(m, dtype=torch.int64, device=dev); st = torch.zeros(m, dtype=torch.int32, device=dev)
    eng.set_stream(stream.cuda_stream); eng.set_sync(False)
    plan = eng.plan_compress(o, oo, 0, 6)
    base = data.data_ptr() + int(off[lo])
    return lambda: eng.compress_planned(plan, base, out.data_ptr(), ol.data_ptr(), st.data_ptr()), (out, ol, st)
def timeit(fns, reps=5):
    for f in fns: f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for f in fns: f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
e0 = Engine(0); f0, keep0 = mk(e0, 0, k, s1)
print("one engine, whole batch: %.2f ms" % timeit([f0]))
for split in (k // 2, (k // 1024) * 512):
    ea, eb = Engine(0), Engine(0)
    fa, ka = mk(ea, 0, split, s1); fb, kb = mk(eb, split, k, s2)
    print("two engines, halves at %d on two streams: %.2f ms" % (split, timeit([fa, fb])))
    del ea, eb
