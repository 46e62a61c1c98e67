import sys, os, numpy as np
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _oracle as O
from flate_amd import Engine, synth
eng = Engine(0)
text = synth.text(synth.SEED_TEXT + 5, 1 << 20).tobytes()
cases = {"zeros70000": bytes(70000), "text65536": text[:65536], "text131072": text[:131072], "text300k": text[7:300007]}
for level in (6,):
    names = list(cases)
    outs, st = eng.compress_many([cases[n] for n in names], 0, level)
    print("status", st, [len(o) for o in outs])
    for i, n in enumerate(names):
        want = O.tokenize(cases[n], level)
        buf = np.zeros(len(cases[n]) + 16, dtype=np.uint32)
        r = eng._L.flate_hip_debug_tokens(eng._h, i, buf.ctypes.data, buf.size)
        print(n, "ntok gpu", r, "oracle", len(want))
        if r > 0:
            got = buf[:r]
            m = min(len(got), len(want))
            bad = np.nonzero(got[:m] != want[:m])[0]
            if bad.size:
                b = int(bad[0])
                print("  first diff at token", b, O.tok_decode(got[b]), O.tok_decode(want[b]), "of", m, "ndiff", bad.size)
        w = O.compress(cases[n], 0, level)
        print("  bytes equal:", outs[i] == w, len(outs[i]), len(w))
