cd /root/repo
cat > /tmp/c4.py <<'PY'
import os, sys, json, argparse
sys.path.insert(0, "/root/repo")
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "/root/repo/bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
from flate_amd import Engine, synth
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng = Engine(0); eng.set_stream(stream.cuda_stream); eng.set_sync(False)
sil = torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA, 128 << 20)).to(dev)
for rep in range(3):
    for steps in (3, 10, 30):
        job = bench.CompressJob(torch, eng, sil, 128 << 20, 1, 1)
        dt, prof = bench.timed(torch, None, eng, job.step, steps, 2, 1)
        print("steps %2d: %.3f ms per step, kernels %s" % (steps, dt / steps * 1e3, {k: round(v[0] / steps, 3) for k, v in sorted(prof.items())}))
PY
echo "== checksum beside (default)"; python /tmp/c4.py 2>/dev/null | grep steps
echo "== checksum in line"; FLATE_HIP_SIMPLE_CK_INLINE=1 python /tmp/c4.py 2>/dev/null | grep steps
