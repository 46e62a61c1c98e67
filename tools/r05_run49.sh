cd /root/repo
FLATE_HIP_LIB=$PWD/flate_amd/lib/var/lib_rkdbg.so timeout 600 python - <<'PY' 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, '/root/repo')
import numpy as np
from flate_amd import Engine, synth
eng = Engine(0)
data = synth.text(synth.SEED_TEXT, 2048 * 65535).tobytes()
chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
t0 = eng.phase_cycles().astype(np.int64)
eng.profile_reset(); eng.profile_enable(True)
outs, st = eng.compress_many(chunks, 0, 9)
prof = eng.profile_read()
t = eng.phase_cycles().astype(np.int64) - t0
print("threads that saw a rank out of line:", int(t[63]), {k: round(v[0], 2) for k, v in prof.items()})
PY
