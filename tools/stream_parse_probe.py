#!/usr/bin/env python3
"""Counters of k_lz_parse<true> (whole-stream path on the chunk tokenizer; library built with -DPZ_PROF): per wave and WINDOW.
usage: FLATE_HIP_LIB=flate_amd/lib/var/libflate_hip_prof.so python tools/stream_parse_probe.py [streams=256] [KiB each=1024] [level=6]"""
import sys, os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
os.environ["FLATE_HIP_STREAM_WINDOWS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = Engine(0)
data = synth.text(synth.SEED_TEXT, ns * kib * 1024).tobytes()
streams = [data[i * kib * 1024:(i + 1) * kib * 1024] for i in range(ns)]
t0 = eng.phase_cycles().astype(np.int64)
outs, st = eng.compress_many(streams, 0, level)
t = eng.phase_cycles().astype(np.int64) - t0
nwin = ns * ((kib * 1024 - 65536 + 32767) // 32768 + 1)
nw = max(int(t[49]), 1)  # wave-windows (every wave adds its counters once per window)
print("%d streams of %d KiB: %d windows, %d wave-windows counted" % (ns, kib, nwin, nw))
print("per wave and window, cycles: total %.0f = staging %.0f + round 0 %.0f (bursts %.0f, slow blocks %.0f) + later rounds %.0f (path following incl. the wait for the slowest wave %.0f)"
      % (t[48] / nw, t[55] / nw, t[46] / nw, t[50] / nw, t[52] / nw, t[47] / nw, t[56] / nw))
print("per window: bursts %.0f (lanes walking at the start %.1f), visits of the slow block %.0f, rounds %.2f per wave" % (t[40] / nwin, t[41] / max(t[40], 1), t[42] / nwin, t[45] / nw))
