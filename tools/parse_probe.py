#!/usr/bin/env python3
"""Counters of k_lz_parse (library built with EXTRA=-DPZ_PROF): wave-level loop statistics per 64 KiB chunk.
usage: FLATE_HIP_LIB=flate_amd/lib/libflate_hip_prof.so python tools/parse_probe.py [n_chunks] [level] [workload]"""
import sys, os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
eng = Engine(0)
nbytes = 65535 * n_chunks
if kind == "text":
    data = synth.text(synth.SEED_TEXT, nbytes)
elif kind == "tar":
    data = synth.tar_like(synth.SEED_TAR, nbytes)
else:
    data = synth.silesia_like(synth.SEED_SILESIA, nbytes)
data = data.tobytes()
chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
t0 = eng.phase_cycles().astype(np.int64)
outs, st = eng.compress_many(chunks, 0, level)
t = eng.phase_cycles().astype(np.int64) - t0
nw = max(int(t[49]), 1)
nc = len(chunks)
print("chunks %d, waves %d" % (nc, nw))
print("per chunk: fast steps %.0f (walking lanes %.1f of 64), slow blocks %.0f (lanes measured %.1f per block), outer loops %.0f, rounds %.2f"
      % (t[40] / nc, t[41] / max(t[40], 1), t[42] / nc, t[43] / max(t[42], 1), t[44] / nc, t[45] / nw))
print("per wave cycles: total %.0f, speculative phase %.0f, stitch rounds %.0f" % (t[48] / nw, t[46] / nw, t[47] / nw))
print("cycles per fast step+share of slow: %.1f" % (t[46] / max(t[40], 1)))
print("per wave cycles: fast loop %.0f, slow block %.0f (of which measure part, lanes that measure only: %.0f)" % (t[50] / nw, t[52] / nw, t[51] / nw))
print("lane 0 of each wave: measure iterations %.1f, transitions %.1f per wave" % (t[53] / nw, t[54] / nw))
print("per wave cycles: staging (incl. barrier) %.0f, path following incl. the wait for the slowest wave %.0f" % (t[55] / nw, t[56] / nw))
print("per wave cycles (k_lz_parse6 only): phase A %.0f" % (t[57] / nw))
print("per wave cycles in bursts by lanes walking when the burst starts: <= 2 lanes %.0f (%.0f bursts), 3-8 %.0f, 9-24 %.0f, more %.0f" % (t[58] / nw, t[62] / nw, t[59] / nw, t[60] / nw, t[61] / nw))
names = ["block entered", "hit: measure", "measure iterations", "match improved", "transition part", "WAIT polls", "WAIT: published", "call ended",
         "call ended with a match", "emit", "emit a match", "segment end / meet", "lazy: look further", "START_CALL", "LOAD_CAND"]
print("events of the slow block, per chunk: waves that pass (lanes per passing wave)")
for i, nm in enumerate(names):
    print("  %-26s %8.1f  (%.1f)" % (nm, t[2 * i] / nc, t[2 * i + 1] / max(t[2 * i], 1)))

if len(t) >= 128:
    print("round 0 per wave (cycles / outer loops), sub-pass A then B, averaged over chunks:")
    for sub in range(2):
        print("  %s cycles: %s" % ("AB"[sub], " ".join("%6.0f" % (t[64 + 16 * sub + w] / nc) for w in range(16))))
        print("  %s loops:  %s" % ("AB"[sub], " ".join("%6.1f" % (t[96 + 16 * sub + w] / nc) for w in range(16))))
