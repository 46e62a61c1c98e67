#!/usr/bin/env python3
"""Debug: a batch dumped by tools/parity_sweep.py (SWEEP_DUMP=1) through the inflater with the paths switched on and off."""
import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flate_amd import Engine
f = sys.argv[1]; container = int(sys.argv[2])
raw = open(f, "rb").read()
outs, datas, p = [], [], 0
while p < len(raw):
    a = int.from_bytes(raw[p:p + 4], "little"); b = int.from_bytes(raw[p + 4:p + 8], "little"); p += 8
    outs.append(raw[p:p + a]); p += a; datas.append(raw[p:p + b]); p += b
print("%d streams, %d of them >= 32768 compressed bytes" % (len(outs), sum(len(o) >= 32768 for o in outs)))
eng = Engine(0)
def bad(back, st):
    return [(i, st[i], len(back[i])) for i in range(len(datas)) if st[i] != 0 or back[i] != datas[i]]
for env in ({}, {"FLATE_HIP_INFLATE_SPANS": "0"}, {"FLATE_HIP_INFLATE_SPANS": "0", "FLATE_HIP_INFLATE_PAR": "0"}, {"FLATE_HIP_SPAN_TWO_RUNS": "1"}, {"FLATE_HIP_SPAN_TWIN": "0"}):
    for k in ("FLATE_HIP_INFLATE_SPANS", "FLATE_HIP_INFLATE_PAR", "FLATE_HIP_SPAN_TWO_RUNS", "FLATE_HIP_SPAN_TWIN"):
        os.environ.pop(k, None)
    os.environ.update(env); eng._sync_env()
    eng.profile_reset(); eng.profile_enable(True)
    back, st, _ = eng.decompress_many(outs, container, 0, [len(d) + 8 for d in datas])
    prof = eng.profile_read(); eng.profile_enable(False)
    print(env, "wrong:", bad(back, st), {k: round(v[0], 2) for k, v in prof.items()})
# the stream alone
os.environ.pop("FLATE_HIP_SPAN_TWIN", None); eng._sync_env()
for i, _, _ in bad(*eng.decompress_many(outs, container, 0, [len(d) + 8 for d in datas])[:2]):
    b1, s1, _ = eng.decompress_many([outs[i]], container, 0, [len(datas[i]) + 8])
    print("stream %d alone: status %s, equal %s" % (i, s1, b1[0] == datas[i]))
    z = zlib.decompressobj(-15 if container == 0 else 15 if container == 2 else 31).decompress(outs[i])
    print("   zlib: equal %s" % (z == datas[i]))
