#!/usr/bin/env python3
"""One whole stream of argv[1] MiB of text at level argv[2]: ms per call (wall, device buffers), by the knobs of the environment."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
mib = float(sys.argv[1]); level = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "text"
n = int(mib * (1 << 20))
data = synth.text(synth.SEED_TEXT, n) if kind == "text" else synth.tar_like(synth.SEED_TAR, n)
off = np.array([0, n], dtype=np.int64); cap = (eng.compress_bound(n, 0, level) + 7) & ~7
oo = np.array([0, cap], dtype=np.int64)
d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off).to(dev); ot = torch.from_numpy(oo).to(dev)
out = torch.empty(cap + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(1, dtype=torch.int64, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
f = lambda: eng.compress_device(d.data_ptr(), io.data_ptr(), 1, 0, level, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
for _ in range(3): f()
torch.cuda.synchronize(); eng.profile_reset(); eng.profile_enable(True)
for _ in range(5): f()
torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
t0 = time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
print("%5.1f MiB %s L%d  W=%s G=%s: %6.2f ms wall  " % (mib, kind, level, os.environ.get("FLATE_HIP_STREAM_WINDOWS", "-"), os.environ.get("FLATE_HIP_STREAM_GROUP", "-"), wall) +
      "  ".join("%s %.2f(%d)" % (kk.replace("k_lz_", "").replace("k_", ""), v[0] / 5, v[1] // 5) for kk, v in sorted(prof.items(), key=lambda x: -x[1][0])[:4]))
