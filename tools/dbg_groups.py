import os, sys, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import _oracle as O
from flate_amd import Engine, synth
eng = Engine(0)
data = synth.text(synth.SEED_TEXT + 21, 12 << 20).tobytes()
os.environ["FLATE_HIP_STREAM_WINDOWS"] = "1"
n = 4784128
d = data[:n]
outs, st = eng.compress_many([d], O.RAW, 6)
got = eng.debug_tokens(0)
want = O.tokenize(d, 6)
print("tokens", len(got), len(want))
m = min(len(got), len(want))
bad = np.nonzero(got[:m] != want[:m])[0]
i = int(bad[0]) if bad.size else m
# position of token i
def pos_of(toks, upto):
    p = 0
    for t in toks[:upto]:
        t = int(t)
        p += ((t >> 15) & 0xff) + 3 if (t >> 23) & 1 else 1
    return p
p0 = pos_of(want, max(0, i - 6))
print("first differing token", i, "position of token i-6:", p0, "rel to window 145 base:", p0 - 145 * 32768, "rel to window 144 base:", p0 - 144 * 32768)
for k in range(max(0, i - 6), min(m, i + 8)):
    print(k, "got", O.tok_decode(got[k]), "want", O.tok_decode(want[k]))
