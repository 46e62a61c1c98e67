import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _edge_stream
eng = Engine(0)
d = _edge_stream(steps=6, a=65273)
for level in (8, 9):
    outs, st = eng.compress_many([d], O.RAW, level)
    want = O.tokenize(d, level); got = eng.debug_tokens(0)
    k = min(len(got), len(want)); bad = np.nonzero(got[:k] != want[:k])[0]
    b = int(bad[0]) if bad.size else k
    pos = 0
    for t in want[:b]:
        dd = O.tok_decode(t); pos += dd[2] if dd[0] == "M" else 1
    print("level", level, "tokens", len(got), len(want), "first bad", b, "at", pos, [O.tok_decode(x) for x in got[b:b+4]], [O.tok_decode(x) for x in want[b:b+4]])
