#!/usr/bin/env python3
"""Debug: one long Silesia-like stream through the whole-stream compressor, windows path on / off: do the streams agree, do they inflate (zlib)?"""
import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
eng = Engine(0)
for mib in [int(x) for x in (sys.argv[1:] or ["8", "32", "128"])]:
    n = mib << 20
    data = synth.silesia_like(synth.SEED_SILESIA, n).tobytes()
    outs = {}
    for w in ("1",):
        os.environ["FLATE_HIP_STREAM_WINDOWS"] = w; eng._sync_env()
        comps, st = eng.compress_many([data], 1, 6)
        outs[w] = comps[0]
        try:
            back = zlib.decompress(comps[0], 31)
            ok = back == data
            first = -1 if ok else next((i for i in range(min(len(back), len(data))) if back[i] != data[i]), min(len(back), len(data)))
        except Exception as e:
            d = zlib.decompressobj(-15)
            back = d.decompress(comps[0][10:])
            ok = False
            first = "%r; raw inflate: %d bytes, equal to the input: %s, first difference %s, unused %d" % (e, len(back), back == data, next((i for i in range(min(len(back), len(data))) if back[i] != data[i]), None), len(d.unused_data))
            import struct, binascii
            print("   footer crc %08x isize %d; crc of the input %08x" % (struct.unpack("<II", comps[0][-8:]) + (binascii.crc32(data) & 0xffffffff,)))
        print("%d MiB windows=%s: status %s, %d bytes, zlib round trip %s (first difference at %s, %d bytes out)" % (mib, w, st, len(comps[0]), ok, first, len(back)), flush=True)
    a, b = outs["1"], outs["1"]
    if a != b:
        k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        print("   streams differ from byte %d on (of %d / %d)" % (k, len(a), len(b)))
