import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import _oracle as O
from flate_amd import Engine
eng = Engine(0)
d = bytes(200000)
outs, st = eng.compress_many([d], O.RAW, 9)
want = O.tokenize(d, 9); got = eng.debug_tokens(0)
print("tokens", len(got), len(want))
pos = 0; last = None
for i, t in enumerate(got):
    dd = O.tok_decode(t)
    if i >= 770 and i < 790: print(i, pos, dd)
    pos += dd[2] if dd[0] == "M" else 1
print("total bytes of got tokens", pos)
