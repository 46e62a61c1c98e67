#!/usr/bin/env python3
"""One-off wide parity sweep (not part of the test suite: it takes minutes of CPU oracle time):
random structured inputs through the chunk path, the whole-stream path, sync flushes and inflate,
all levels / containers, against the oracle.  Usage: parity_sweep.py [seed] [rounds]"""
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _fuzz_input
from test_gpu_flush import _oracle_stream

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = Engine(0)
rng = np.random.default_rng(seed)
bad = 0
for rd in range(rounds):
    # chunk path: many short inputs
    datas = []
    for i in range(60):
        d = _fuzz_input(int(rng.integers(1, 1 << 30)))
        n = int(rng.integers(0, 65536))
        o = int(rng.integers(0, max(1, len(d) - n)))
        datas.append(d[o:o + n])
    for mode in (0, 1, 4, 5, 6, 7, 8, 9):
        container = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, container, mode)
        for d, got, s in zip(datas, outs, st):
            # status 102 (the reference's own Q1 stream): exactly when the ORACLE's inflater does not get the input back
            back_ok = mode < 4 or O.decompress(got, container, 0, cap=len(d) + 600)[:2] == ("Ok", d)
            if s != (0 if back_ok else 102) or got != O.compress(d, container, mode):
                bad += 1
                print("CHUNK MISMATCH", rd, mode, container, len(d), s)
        back, st2, _ = eng.decompress_many(outs, container, 0, [len(d) + 8 for d in datas])
        # (what a stream inflates to is what the ORACLE's inflater makes of it: the reference's own streams do not always give
        # the input back -- a block that fills its 32768 tokens with a match next to a stored block: DESIGN.md section 3)
        want = [O.decompress(o, container, 0, cap=(len(d) + 8 + 7) & ~7) for o, d in zip(outs, datas)]  # (the engine rounds slots up to 8)
        if [O.STATUS[s_] for s_ in st2] != [w[0] for w in want] or any(w[0] == "Ok" and b != w[1] for b, w in zip(back, want)):
            bad += 1
            print("INFLATE MISMATCH", rd, mode, container)
            for i, (d, b, s_) in enumerate(zip(datas, back, st2)):
                if O.STATUS[s_] != want[i][0] or (want[i][0] == "Ok" and b != want[i][1]):
                    k = next((j for j in range(min(len(b), len(d))) if b[j] != d[j]), min(len(b), len(d)))
                    print("   stream %d: %d bytes compressed, %d plain: status %d, %d bytes out, first difference at %d" % (i, len(outs[i]), len(d), s_, len(b), k))
                    if os.environ.get("SWEEP_DUMP"):
                        os.makedirs(os.path.join(ROOT, "gpurun_out", "sweep_dump"), exist_ok=True)
                        open(os.path.join(ROOT, "gpurun_out", "sweep_dump", "r%d_m%d_c%d_batch.bin" % (rd, mode, container)), "wb").write(
                            b"".join(len(o).to_bytes(4, "little") + len(dd).to_bytes(4, "little") + o + dd for o, dd in zip(outs, datas)))
    # whole-stream path
    datas = [_fuzz_input(int(rng.integers(1, 1 << 30))) for _ in range(10)]
    for mode in (4, 5, 6, 7, 8, 9):
        container = int(rng.integers(0, 3))
        outs, st = eng.compress_many(datas, container, mode)
        for d, got, s in zip(datas, outs, st):
            back_ok = O.decompress(got, container, 0, cap=len(d) + 600)[:2] == ("Ok", d)
            if s != (0 if back_ok else 102) or got != O.compress(d, container, mode):
                bad += 1
                print("STREAM MISMATCH", rd, mode, container, len(d), s)
    # sync flushes
    for d in datas[:6]:
        n = len(d)
        k = int(rng.integers(1, 8))
        fl = sorted(int(x) for x in rng.integers(0, n + 1, k))
        finish = bool(rng.random() < 0.7)
        if not finish:
            fl = [f for f in fl if f < n] + [n]
        mode = int(rng.choice([0, 1, 4, 6, 9]))
        container = int(rng.integers(0, 3))
        got, s = eng.compress_flush(d, fl, finish, container, mode)
        if s not in (0, 102) or got != _oracle_stream(d, fl, finish, container, mode)[0]:
            bad += 1
            print("FLUSH MISMATCH", rd, mode, container, n, fl, finish, s)
    print("round", rd, "done, mismatches so far:", bad, flush=True)
print("SWEEP", "FAILED" if bad else "OK", bad)
