#!/usr/bin/env python3
"""Host-buffer inflate of the benchmark's 16385 streams (1 GiB): pageable, pinned with the overlapped sub-batches,
pinned as one batch."""
import sys, os, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth, _capi
eng = Engine(0); L = _capi.lib()
n = (int(sys.argv[1]) << 20) if len(sys.argv) > 1 else (1 << 30)
data = synth.text(synth.SEED_TEXT, n)
off = synth.split_offsets(n, 65535).astype(np.uint64); k = len(off) - 1
chunks = [data[int(off[i]):int(off[i+1])].tobytes() for i in range(k)]
outs, st = eng.compress_many(chunks, 0, 6)
lens = np.array([len(o) for o in outs], dtype=np.int64)
c_off = np.zeros(k + 1, dtype=np.uint64); np.cumsum(lens, out=c_off[1:].view(np.int64))
packed = np.frombuffer(b"".join(outs), dtype=np.uint8).copy()
dlen = np.zeros(k, dtype=np.uint64); status = np.zeros(k, dtype=np.int32)
def run(pin, tag):
    pi = torch.from_numpy(packed); po = torch.zeros(n + 8, dtype=torch.uint8)
    if pin: pi = pi.pin_memory(); po = po.pin_memory()
    def f():
        rc = L.flate_hip_decompress_batch(eng._h, pi.data_ptr(), c_off.ctypes.data, k, 0, 0, po.data_ptr(), off.ctypes.data, dlen.ctypes.data, status.ctypes.data, None, _capi.MEM_HOST)
        assert rc == 0 and not status.any()
    f(); f(); ts = []   # the MEDIAN of six calls after two warm-ups (the second call of a process is not the steady state: r05_host_path.txt)
    for _ in range(6):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    dt = sorted(ts)[3]
    assert np.array_equal(po.numpy()[:n], data)
    print(tag, round(n / dt / 1e6, 1), "MB/s")
run(False, "pageable")
for lim in (sys.argv[2:] or ["1024"]):
    os.environ["FLATE_HIP_HOST_PASS_CHUNKS"] = lim
    eng._sync_env()  # (the library reads its knobs once per handle)
    run(True, "pinned overlapped (sub-batches of about 4 x %s streams)" % lim)
os.environ["FLATE_HIP_HOST_PASS_CHUNKS"] = "100000"
eng._sync_env()  # (the library reads its knobs once per handle)
run(True, "pinned, one batch")
