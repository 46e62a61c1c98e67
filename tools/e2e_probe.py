#!/usr/bin/env python3
"""Host-buffer compress (level 6, 256 MiB text, 65535-byte chunks) through the C ABI: pinned and pageable, by sub-batch size."""
import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flate_amd import Engine, synth, _capi
eng = Engine(0); L = _capi.lib()
n = 256 << 20
data = synth.text(synth.SEED_TEXT, n)
off = synth.split_offsets(n, 65535).astype(np.uint64); k = len(off) - 1
caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, 6) + 7) & ~7 for i in range(k)], dtype=np.uint64)
oo = np.zeros(k + 1, dtype=np.uint64); np.cumsum(caps, out=oo[1:])
out_len = np.zeros(k, dtype=np.uint64); status = np.zeros(k, dtype=np.int32)
p_in = torch.from_numpy(data).pin_memory(); p_out = torch.zeros(int(oo[-1]) + 8, dtype=torch.uint8).pin_memory()
g_out = np.zeros(int(oo[-1]) + 8, dtype=np.uint8)
def run(inp, outp):
    rc = L.flate_hip_compress_batch(eng._h, inp, off.ctypes.data, k, 0, 6, outp, oo.ctypes.data, out_len.ctypes.data, status.ctypes.data, _capi.MEM_HOST)
    assert rc == 0 and not status.any()
def t(f, reps=6):
    # the MEDIAN of the calls after two warm-up calls (round 5: the best call of a few used to be the second call of the
    # process, the only one in which the engine's copies overlapped: profiles/r05_host_path.txt)
    f(); f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
only_pinned = bool(os.environ.get("E2E_PINNED_ONLY"))
# what the link gives THIS process's pinned buffers (their placement differs from process to process)
_d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
def _bw(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return n / (time.perf_counter() - t0) / 1e9
if not os.environ.get("E2E_NO_TOUCH"): print("this process: pinned H2D %.1f GB/s, D2H into the pinned output buffer %.1f GB/s" % (_bw(lambda: _d.copy_(p_in, non_blocking=True)), _bw(lambda: p_out[:n].copy_(_d, non_blocking=True))))
# ... and both directions at once on two streams: the engine's copies of one process either overlap (about 48 GB/s each way)
# or take turns (28.6 = half of one direction's 57): profiles/r05_host_path.txt
_d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0"); _s1, _s2 = torch.cuda.Stream(), torch.cuda.Stream()
def _both():
    with torch.cuda.stream(_s1): _d.copy_(p_in, non_blocking=True)
    with torch.cuda.stream(_s2): p_out[:n].copy_(_d2, non_blocking=True)
if not os.environ.get("E2E_NO_TOUCH"): print("this process: both directions at once %.1f GB/s each way" % _bw(_both))
def throttled():
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(l.split() for l in open(f).read().strip().splitlines())
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            pass
    return (-1, -1)
for sub in sys.argv[1:] or ["1024"]:
    os.environ["FLATE_HIP_HOST_PASS_CHUNKS"] = sub
    eng._sync_env()  # (the library reads its knobs once per handle)
    th0 = throttled()
    a = t(lambda: run(p_in.data_ptr(), p_out.data_ptr()))
    if os.environ.get("E2E_MIX"):
        a1 = t(lambda: run(p_in.data_ptr(), g_out.ctypes.data)); a2 = t(lambda: run(data.ctypes.data, p_out.data_ptr()))
        print("   pinned in + pageable out %.2f ms; pageable in + pinned out %.2f ms" % (a1 * 1e3, a2 * 1e3))
    os.environ["FLATE_HIP_RECT"] = "0"; eng._sync_env()   # the produced bytes by the copy kernel alone (no rectangle copy by the DMA engine)
    a0 = t(lambda: run(p_in.data_ptr(), p_out.data_ptr()))
    os.environ["FLATE_HIP_RECT"] = "1"; eng._sync_env()
    a1r = t(lambda: run(p_in.data_ptr(), p_out.data_ptr()))
    del os.environ["FLATE_HIP_RECT"]; eng._sync_env()
    print("   pinned, D2H by the copy kernel alone %.2f ms (%.1f GB/s); with the rectangle copy forced %.2f ms (%.1f GB/s)" % (a0 * 1e3, n / a0 / 1e9, a1r * 1e3, n / a1r / 1e9))
    b = 1.0 if only_pinned else t(lambda: run(data.ctypes.data, g_out.ctypes.data))
    th1 = throttled()
    print("   cgroup cpu.stat while timing: throttled %d times, %d us" % (th1[0] - th0[0], th1[1] - th0[1]))
    print("sub-batches of %5s chunks: pinned %6.1f GB/s (%.2f ms)   pageable %6.1f GB/s (%.2f ms)" % (sub, n / a / 1e9, a * 1e3, n / b / 1e9, b * 1e3))
po = p_out.numpy()
ok = only_pinned or all(np.array_equal(po[int(oo[i]):int(oo[i]) + int(out_len[i])], g_out[int(oo[i]):int(oo[i]) + int(out_len[i])]) for i in range(0, k, 97))
print("pinned == pageable outputs (sampled):", ok)
