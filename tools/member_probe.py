#!/usr/bin/env python3
"""Which members of config #5 are slow?  Every 1 MiB member of the bench's Silesia-like buffer alone through the inflater
(a workgroup per stream, spans off) and the batch of 128 with spans on; prints the slowest members and what they are."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth
eng = Engine(0)
sil = synth.silesia_like(synth.SEED_SILESIA + 1, 128 << 20).tobytes()
members = [sil[i << 20:(i + 1) << 20] for i in range(128)]
comps, st = eng.compress_many(members, 1, 6)
os.environ["FLATE_HIP_INFLATE_SPANS"] = "0"; eng._sync_env()
rows = []
for i, c in enumerate(comps):
    eng.profile_reset(); eng.profile_enable(True)
    outs, st, used = eng.decompress_many([c], 1, caps=[1 << 20])
    prof = eng.profile_read(); eng.profile_enable(False)
    assert st == [0] and outs[0] == members[i]
    rows.append((sum(v[0] for v in prof.values()), i, len(c)))
rows.sort(reverse=True)
def kind(d):
    a = np.frombuffer(d, dtype=np.uint8)
    z = float((a == 0).mean())
    return "zeros %.2f" % z
for ms, i, n in rows[:12] + rows[-4:]:
    print("member %3d: %7d compressed bytes, %.2f ms alone (zero bytes %s, first bytes %r)" % (i, n, ms, kind(members[i]), members[i][:24]))
os.environ.pop("FLATE_HIP_INFLATE_SPANS"); eng._sync_env()
for k in range(2):
    eng.profile_reset(); eng.profile_enable(True)
    t0 = time.time(); outs, st, used = eng.decompress_many(comps, 1, caps=[1 << 20] * 128); dt = time.time() - t0
    prof = eng.profile_read(); eng.profile_enable(False)
print("all 128 with spans: %s" % {k: round(v[0], 2) for k, v in prof.items()})
