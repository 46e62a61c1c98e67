cd /root/repo
python - <<'PY'
import os, sys, json
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch, time
from flate_amd import Engine, synth
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
def run(name, data, sizes, level, container):
    n = len(data); off = np.zeros(len(sizes) + 1, dtype=np.int64); np.cumsum(sizes, out=off[1:]); k = len(sizes)
    caps = np.array([(eng.compress_bound(int(s), container, level) + 7) & ~7 for s in sizes], dtype=np.int64)
    oo = np.zeros(k + 1, dtype=np.int64); np.cumsum(caps, out=oo[1:])
    d = torch.from_numpy(data).to(dev); io = torch.from_numpy(off).to(dev); ot = torch.from_numpy(oo).to(dev)
    out = torch.empty(int(oo[-1]) + 8, dtype=torch.uint8, device=dev); ol = torch.zeros(k, dtype=torch.int64, device=dev); st = torch.zeros(k, dtype=torch.int32, device=dev)
    f = lambda: eng.compress_device(d.data_ptr(), io.data_ptr(), k, container, level, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
    f(); torch.cuda.synchronize(); eng.profile_reset(); eng.profile_enable(True)
    for _ in range(3): f()
    torch.cuda.synchronize(); prof = eng.profile_read(); eng.profile_enable(False)
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
    print("%-40s %7.1f MB/s %7.2f ms  " % (name, n / wall / 1e3, wall) + "  ".join("%s %.2f" % (kk.replace("k_lz_", "").replace("k_", ""), v[0] / 3) for kk, v in sorted(prof.items(), key=lambda x: -x[1][0])[:5]))
tar = synth.tar_like()
run("tar-like 169 MiB, one gzip-9 stream", tar, [len(tar)], 9, 1)
text = synth.text(synth.SEED_TEXT, 256 << 20)
run("text 256 MiB, one stream, level 9", text, [len(text)], 9, 0)
run("text 256 MiB, 256 streams, level 9", text, [1 << 20] * 256, 9, 0)
run("text 256 MiB, 64 streams, level 8", text, [4 << 20] * 64, 8, 0)
run("text 96 MiB, 700 streams, level 9", text[:700 * 140000], [140000] * 700, 9, 0)
sil = synth.silesia_like(synth.SEED_SILESIA, 128 << 20)
run("silesia-like 128 MiB, one stream, level 9", sil, [len(sil)], 9, 1)
run("silesia-like 128 MiB, 16 streams, level 9", sil, [8 << 20] * 16, 9, 1)
PY
