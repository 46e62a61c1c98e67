import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import _oracle as O
from flate_amd import Engine
from test_gpu_stream import CASES
eng = Engine(0)
names = [n for n in CASES if len(CASES[n]) > 65535]
for level in (9,):
    outs, st = eng.compress_many([CASES[n] for n in names], O.RAW, level)
    for i, n in enumerate(names):
        want = O.tokenize(CASES[n], level)
        got = eng.debug_tokens(i)
        if len(got) != len(want) or (got != want).any():
            k = min(len(got), len(want))
            bad = np.nonzero(got[:k] != want[:k])[0]
            b = int(bad[0]) if bad.size else k
            pos = 0
            for t in want[:b]:
                d = O.tok_decode(t); pos += d[2] if d[0] == "M" else 1
            print("MISMATCH", n, len(CASES[n]), "tokens", len(got), len(want), "first bad token", b, "at position", pos, [O.tok_decode(x) for x in got[b:b+3]], [O.tok_decode(x) for x in want[b:b+3]])
        else:
            print("ok", n)
