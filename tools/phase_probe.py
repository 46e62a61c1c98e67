#!/usr/bin/env python3
"""Print the per-phase shader-clock breakdown of workgroup 0 of the tokenizer kernels."""
import sys, os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = Engine(0)
data = synth.text(synth.SEED_TEXT, 65535 * 8).tobytes()
chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)] * (n_chunks // 8)
for rep in range(2):
    outs, st = eng.compress_many(chunks, 0, level)
t = eng.phase_cycles().astype(np.int64)
def seg(name, a, b): print("%-28s %10d cycles" % (name, t[b] - t[a]))
print("n_chunks", len(chunks))
seg("sort: zero+count1", 0, 1); seg("sort: scan1", 1, 2); seg("sort: scatter1 (LDS)", 2, 3); seg("sort: scan2", 3, 4)
seg("sort: scatter2 (global)", 4, 5); seg("sort: total", 0, 5)
seg("match: stage window", 8, 9); seg("match: bucket offsets", 9, 10); seg("match: batches", 10, 11); seg("match: total", 8, 11)
seg("tok part 0: stage", 16, 23); seg("tok part 0: (a) + sub-piece jumps", 23, 17); seg("tok part 0: piece jumps", 17, 24)
seg("tok part 0: barrier", 24, 18); seg("tok part 0: (c) serial", 18, 19); seg("tok part 0: (d) + (e)", 19, 21)
seg("tok part 0: (f) count", 21, 25); seg("tok part 0: (f) emit", 25, 22); seg("tok: all parts", 16, 27)
if t[32:40].any() and os.environ.get("FL_WALK_PROF"):
    for k, nm in {32: "walk: iterations", 33: "walk: refill rounds", 34: "walk: verify rounds", 35: "walk: walking lanes (sum)",
                  36: "walk: free lanes at refill (sum)", 37: "walk: lanes verified (sum)", 38: "walk: sleeps"}.items():
        print("%-34s %10d" % (nm, t[k] // 2))
    seg("walk: stage", 8, 9); seg("walk: loop", 9, 11)
elif t[32:48].any():
    names = {32: "m2: slice init", 33: "m2: tile build", 34: "m2: bin + perm", 35: "m2: (unused)", 36: "m2: group setup", 37: "m2: cut search", 38: "m2: unit loop + deep", 39: "m2: records",
             40: "m2: # groups",
             43: "m2: # deep loop iterations"}
    for k in sorted(names):
        print("%-28s %10d" % (names[k], t[k]))
