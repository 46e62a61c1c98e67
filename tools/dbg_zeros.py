import os, sys, time
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import _oracle as O
from flate_amd import Engine
eng = Engine(0)
for name, d in (("zeros 0.2 MiB", bytes(200000)), ("zeros 4 MiB", bytes(4 << 20)), ("abab 1 MiB", b"ab" * (1 << 19)), ("period 258 1 MiB", (bytes(range(256)) + b"xy") * 4064)):
    for _ in range(2): outs, st = eng.compress_many([d], O.GZIP, 6)
    eng.profile_enable(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5): outs, st = eng.compress_many([d], O.GZIP, 6)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    prof = eng.profile_read(); eng.profile_enable(False)
    print("%-18s W=%s: %.2f ms a call  " % (name, os.environ.get("FLATE_HIP_STREAM_WINDOWS", "-"), wall) + "  ".join("%s %.2f(%d)" % (k.replace("k_lz_", "").replace("k_", ""), v[0] / 5, v[1] // 5) for k, v in sorted(prof.items(), key=lambda x: -x[1][0])[:5]), outs[0] == O.compress(d, O.GZIP, 6))
