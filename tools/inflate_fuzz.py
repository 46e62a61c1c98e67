#!/usr/bin/env python3
"""Differential fuzz of the batched inflater against the oracle (one-off, minutes): zlib- and library-made streams of many
kinds, whole and damaged (truncations, bit flips, byte splices, tails), through k_inflate alone (both rings) and through the
library's own choice of kernels.  Status name, bytes and consumed count must agree.  Usage: inflate_fuzz.py [seed] [rounds]"""
import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _fuzz_input
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(seed)
bad = 0
for rd in range(rounds):
    bases = []
    for i in range(24):
        d = _fuzz_input(int(rng.integers(1, 1 << 30)))
        n = int(rng.integers(200, 200000)); o = int(rng.integers(0, max(1, len(d) - n)))
        d = d[o:o + n]
        container = int(rng.integers(0, 3))
        if rng.random() < 0.5:
            wb = {0: -15, 1: 31, 2: 15}[container]
            co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, wb, 9, int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED])))
            comp = co.compress(d) + co.flush()
        else:
            comp = O.compress(d[:65535], container, int(rng.choice([1, 4, 6, 9]))); d = d[:65535]
        bases.append((container, d, comp))
    cases = []
    for container, d, comp in bases:
        cases.append((container, comp, (len(d) + 15) & ~7))
        for _ in range(10):
            m = bytearray(comp); k = int(rng.integers(0, 5))
            if k == 0: m = m[:int(rng.integers(1, len(m)))]
            elif k == 1: m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif k == 2:
                a = int(rng.integers(0, len(m))); b = int(rng.integers(0, len(m))); n = int(rng.integers(1, 40)); m[a:a + n] = m[b:b + n]
            elif k == 3: m += bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
            else: m[int(rng.integers(len(m) // 2, len(m)))] = int(rng.integers(0, 256))
            cap = ((len(d) + 15) & ~7) if rng.random() < 0.8 else max(8, (len(d) // 2) & ~7)  # (multiples of 8: decompress_many rounds the slots up)
            cases.append((container, bytes(m), cap))
    for env in ({"FLATE_HIP_INFLATE_PAR": "0", "FLATE_HIP_INFLATE_SPANS": "0", "FLATE_HIP_INFLATE_RING": "2048"},
                {"FLATE_HIP_INFLATE_PAR": "0", "FLATE_HIP_INFLATE_SPANS": "0", "FLATE_HIP_INFLATE_RING": "32768"}, {}):
        for k in ("FLATE_HIP_INFLATE_PAR", "FLATE_HIP_INFLATE_SPANS", "FLATE_HIP_INFLATE_RING"): os.environ.pop(k, None)
        os.environ.update(env)
        eng = Engine(0)
        for container in (0, 1, 2):
            grp = [c for c in cases if c[0] == container]
            outs, st, used = eng.decompress_many([c[1] for c in grp], container, caps=[c[2] for c in grp])
            for c, o, s_, u in zip(grp, outs, st, used):
                name, want, wused = O.decompress(c[1], container, 0, cap=c[2])
                if O.STATUS[s_] != name or (name == "Ok" and (o != want or u != wused)):
                    bad += 1
                    print("MISMATCH", rd, env, container, len(c[1]), c[2], O.STATUS[s_], name)
    print("round", rd + 1, "done,", len(cases), "streams x 3 settings, mismatches so far:", bad, flush=True)
print("FUZZ", "OK" if bad == 0 else "FAILED", bad)
