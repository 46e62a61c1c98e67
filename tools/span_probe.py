#!/usr/bin/env python3
"""One long stream through the inflater: time, kernels, bytes (tuning aid for the span path).
usage: python tools/span_probe.py [MiB=64] [mode=6] [container=1] [text|silesia] [streams=1]  (MiB in all, cut into equal streams)"""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 6
container = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kind = sys.argv[4] if len(sys.argv) > 4 else "text"
eng = Engine(0)
n = mib << 20
data = (synth.text(synth.SEED_TEXT, n) if kind == "text" else synth.silesia_like(synth.SEED_SILESIA, n)).tobytes()
ns = int(sys.argv[5]) if len(sys.argv) > 5 else 1
datas = [data[i * (n // ns):(i + 1) * (n // ns)] for i in range(ns)]
comps, st = eng.compress_many(datas, container, mode)
assert st == [0] * ns
print("%d stream(s): %d -> %d bytes" % (ns, len(data), sum(len(c) for c in comps)))
for env in ("0", os.environ.get("SPAN_PROBE_MIN")):  # (SPAN_PROBE_MIN: the library's lower bound for the second run)
    if env is None:
        os.environ.pop("FLATE_HIP_INFLATE_SPANS", None)
    else:
        os.environ["FLATE_HIP_INFLATE_SPANS"] = env
        eng._sync_env()  # (the library reads its knobs once per handle)
    eng.profile_reset(); eng.profile_enable(True)
    t0 = time.time()
    outs, st, used = eng.decompress_many(comps, container, caps=[len(d) for d in datas])
    dt = time.time() - t0
    prof = eng.profile_read(); eng.profile_enable(False)
    ok = st == [0] * ns and outs == datas and used == [len(c) for c in comps]
    print("spans %-4s: ok=%s status=%s wall %.1f ms (host copies included)  kernels: %s" % (
        "off" if env == "0" else "on", ok, sorted(set(st)), dt * 1e3, {k: round(v[0], 2) for k, v in prof.items()}))
