#!/usr/bin/env python3
"""Two numbers DESIGN.md quotes beside the headline (never as the headline):
  * the PCIe-inclusive rate: flate_hip_compress_batch / decompress_batch on HOST buffers
    (H2D of the input, kernels, D2H of the output) for 256 MiB of the benchmark text;
  * the CPU oracle on all host cores (one process per core over the same independent chunks)."""
import multiprocessing as mp
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

CH = 65535


_BLOB = None  # inherited by the forked workers: no pickling of the input


def _work(args):
    import _oracle as O
    lo, hi, reps = args
    n = 0
    for _ in range(reps):
        for i in range(lo, hi):
            n += len(O.compress(_BLOB[i * CH:(i + 1) * CH], 0, 6))
    return n


def main():
    from flate_amd import Engine, synth, _capi
    from flate_amd.engine import MEM_HOST
    n = 256 << 20
    data = synth.text(synth.SEED_TEXT, n)
    off = synth.split_offsets(n, CH).astype(np.uint64)
    k = len(off) - 1
    eng = Engine(0)
    caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, 6) + 7) & ~7 for i in range(k)], dtype=np.uint64)
    out_off = np.zeros(k + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
    out_len = np.zeros(k, dtype=np.uint64)
    st = np.zeros(k, dtype=np.int32)
    L = _capi.lib()

    def comp():
        rc = L.flate_hip_compress_batch(eng._h, data.ctypes.data, off.ctypes.data, k, 0, 6, out.ctypes.data,
                                        out_off.ctypes.data, out_len.ctypes.data, st.ctypes.data, MEM_HOST)
        assert rc == 0 and not st.any()

    comp()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); comp(); t.append(time.perf_counter() - t0)
    n_out = int(out_len.sum())
    print("host-buffer compress (H2D + kernels + D2H, pageable memory): %.1f ms = %.0f MB/s for %d MiB" %
          (min(t) * 1e3, n / min(t) / 1e6, n >> 20))
    # inflate on host buffers: pack the streams
    comp_blob = np.concatenate([out[int(out_off[i]):int(out_off[i]) + int(out_len[i])] for i in range(k)])
    coff = np.zeros(k + 1, dtype=np.uint64); np.cumsum(out_len, out=coff[1:])
    dec = np.zeros(n + 8, dtype=np.uint8); dlen = np.zeros(k, dtype=np.uint64); dst = np.zeros(k, dtype=np.int32)
    used = np.zeros(k, dtype=np.uint64)

    def decomp():
        rc = L.flate_hip_decompress_batch(eng._h, comp_blob.ctypes.data, coff.ctypes.data, k, 0, 0, dec.ctypes.data,
                                          off.ctypes.data, dlen.ctypes.data, dst.ctypes.data, used.ctypes.data, MEM_HOST)
        assert rc == 0 and not dst.any()

    decomp()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); decomp(); t.append(time.perf_counter() - t0)
    assert np.array_equal(dec[:n], data)
    print("host-buffer inflate: %.1f ms = %.0f MB/s" % (min(t) * 1e3, n / min(t) / 1e6))

    # CPU oracle on all cores: every process compresses its own slice of the chunks, several times over
    global _BLOB
    cores = os.cpu_count() or 1
    _BLOB = data.tobytes()
    per = max(1, k // cores)
    reps = max(1, int(32 * (1 << 20) / (per * CH)))
    jobs = [(i, min(k, i + per), reps) for i in range(0, per * cores, per) if i < k]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        pool.map(_work, [(0, 1, 1)] * len(jobs))  # warm: load the oracle in every worker
        t0 = time.perf_counter()
        pool.map(_work, jobs, chunksize=1)
        dt = time.perf_counter() - t0
    total = sum((hi - lo) * r for lo, hi, r in jobs) * CH
    print("CPU oracle, level 6, %d processes on %d cores: %.0f MB/s (%.1f GiB of chunks, %.2f s)" %
          (len(jobs), cores, total / dt / 1e6, total / 2 ** 30, dt))


if __name__ == "__main__":
    main()
