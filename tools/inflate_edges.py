#!/usr/bin/env python3
"""Directed inflate cases that fuzzing rarely builds: fixed-block symbols 286 / 287 and distance codes 30 / 31, a distance equal to /
one more than what has been written, lengths and distances at the ring's edges, stored blocks of 0 and 65535 bytes, code-length
repeats that cross from the literal into the distance table, a dynamic block without distance codes that uses a match.  Status,
bytes and consumed count: GPU == oracle, through every decoder of the library.  usage: inflate_edges.py"""
import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine


from _inflate_edge_cases import CASES as cases

eng = Engine(0)
names = sorted(cases)
streams = [cases[n] for n in names]
bad = 0
for knobs in ({}, {"FLATE_HIP_INFLATE_PAR": "0"}, {"FLATE_HIP_INFLATE_RING": "32768"}, {"FLATE_HIP_INFLATE_PAR": "1", "FLATE_HIP_INFLATE_SPANS": "64"}):
    for k in ("FLATE_HIP_INFLATE_PAR", "FLATE_HIP_INFLATE_RING", "FLATE_HIP_INFLATE_SPANS"):
        os.environ.pop(k, None)
    os.environ.update(knobs)
    caps = [80000] * len(streams)
    back, st, used = eng.decompress_many(streams, O.RAW, caps=caps)
    for n, s, b, stt, u in zip(names, streams, back, st, used):
        w = O.decompress(s, O.RAW, 0, cap=80000)
        if O.STATUS[stt] != w[0] or (w[0] == "Ok" and (b != w[1] or u != w[2])):
            bad += 1
            print("INFLATE EDGE MISMATCH", knobs, n, "gpu", O.STATUS[stt], len(b), u, "oracle", w[0], len(w[1]), w[2], flush=True)
for n, s in zip(names, streams):
    w = O.decompress(s, O.RAW, 0, cap=80000)
    print("%-24s %-22s %6d bytes" % (n, w[0], len(w[1])))
print("INFLATE EDGES", "FAILED" if bad else "OK", bad)
