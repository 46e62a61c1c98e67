import sys, os, numpy as np
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import _oracle as O
from flate_amd import Engine
from test_gpu_stream import _fuzz_input
from test_gpu_flush import _oracle_stream
eng = Engine(0)
seed, n, flushes, finish, level = 2000, 144260, [65543, 67445, 97953, 130899, 139216], True, 4
if len(sys.argv) > 1:
    import json
    seed, n, flushes, finish, level = json.loads(sys.argv[1])
data = _fuzz_input(seed)[:n]
got, st = eng.compress_flush(data, flushes, finish, O.RAW, level)
want, wtok = _oracle_stream(data, flushes, finish, O.RAW, level, tokens=True)
toks = eng.debug_tokens(0)
print("bytes equal", got == want, len(got), len(want), "tokens", len(toks), len(wtok))
m = min(len(toks), len(wtok))
bad = np.nonzero(toks[:m] != wtok[:m])[0]
if bad.size or len(toks) != len(wtok):
    first = int(bad[0]) if bad.size else m
    pos = 0
    for t in wtok[:first]:
        d = O.tok_decode(t)
        pos += 1 if d[0] == "L" else d[2]
    print("first differing token", first, "at stream position", pos)
    for k in range(max(0, first - 2), min(m, first + 4)):
        print(k, O.tok_decode(toks[k]), O.tok_decode(wtok[k]))
else:
    # token lists equal: find first differing byte
    for i, (a, b) in enumerate(zip(got, want)):
        if a != b:
            print("first differing byte", i); break
