#!/usr/bin/env python3
"""Run BASELINE.json's other configurations on one MI355X (parity cases, not the headline bench):
  #3 gzip level 9 on a TAR-like 177,244,160-byte buffer (65535-byte chunks)
  #4 huffman-only, one 128 MiB Silesia-like buffer as ONE stream (2049 blocks)
  #5 batched gunzip of 128 x 1 MiB gzip members (level 6, produced by the CPU oracle)
Each prints one JSON line with throughput and the parity check performed."""
import json
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import _oracle as O
from flate_amd import Engine, synth


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def dev_arrays(eng, data_np, in_off_np, container, mode):
    dev = torch.device("cuda", 0)
    n = len(in_off_np) - 1
    caps = np.array([(eng.compress_bound(int(in_off_np[i + 1] - in_off_np[i]), container, mode) + 7) & ~7
                     for i in range(n)], dtype=np.uint64)
    out_off_np = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off_np[1:])
    t = dict(data=torch.from_numpy(data_np).to(dev), in_off=torch.from_numpy(in_off_np.astype(np.int64)).to(dev),
             out_off=torch.from_numpy(out_off_np.astype(np.int64)).to(dev),
             out=torch.empty(int(out_off_np[-1]) + 8, dtype=torch.uint8, device=dev),
             out_len=torch.zeros(n, dtype=torch.int64, device=dev), status=torch.zeros(n, dtype=torch.int32, device=dev))
    return t, out_off_np


def config3(eng):
    n = 177_244_160
    # TAR-like: 512-byte headers + text bodies + zero padding
    body = synth.tar_like(synth.SEED_TAR, n)
    in_off = synth.split_offsets(n, 65535)
    t, out_off = dev_arrays(eng, body, in_off, 1, 9)
    k = len(in_off) - 1

    def run():
        eng.compress_device(t["data"].data_ptr(), t["in_off"].data_ptr(), k, 1, 9, t["out"].data_ptr(),
                            t["out_off"].data_ptr(), t["out_len"].data_ptr(), t["status"].data_ptr())

    dt = timed(run, 2)
    assert int(t["status"].abs().sum()) == 0
    lens = t["out_len"].cpu().numpy()
    # parity on a sample of chunks
    for i in np.linspace(0, k - 1, 24).astype(int):
        src = body[int(in_off[i]):int(in_off[i + 1])].tobytes()
        got = t["out"][int(out_off[i]):int(out_off[i]) + int(lens[i])].cpu().numpy().tobytes()
        assert got == O.compress(src, O.GZIP, 9), i
    return {"config": "#3 gzip level 9, TAR-like 177,244,160 B, 65535-byte chunks", "MB/s": round(n / dt / 1e6, 1),
            "ms": round(dt * 1e3, 2), "ratio": round(float(lens.sum()) / n, 4), "parity": "24 sampled chunks == oracle"}


def config3_stream(eng):
    """#3 as ONE gzip stream (whole-stream path): the config's "deep hash-chain match search" over
    a single 177 MB input.  Parity: the first 3 MiB of it as a stream of its own == oracle."""
    n = 177_244_160
    body = synth.tar_like(synth.SEED_TAR, n)
    in_off = np.array([0, n], dtype=np.uint64)
    t, out_off = dev_arrays(eng, body, in_off, 1, 9)

    def run():
        eng.compress_device(t["data"].data_ptr(), t["in_off"].data_ptr(), 1, 1, 9, t["out"].data_ptr(),
                            t["out_off"].data_ptr(), t["out_len"].data_ptr(), t["status"].data_ptr())

    dt = timed(run, 2)
    assert int(t["status"].abs().sum()) == 0
    ln = int(t["out_len"][0])
    import zlib
    got = t["out"][:ln].cpu().numpy().tobytes()
    assert zlib.decompress(got, 31) == body.tobytes()
    head = body[:3 << 20].tobytes()
    outs, st = eng.compress_many([head], 1, 9)
    assert st == [0] and outs[0] == O.compress(head, O.GZIP, 9)
    return {"config": "#3 gzip level 9, TAR-like 177,244,160 B as ONE stream (whole-stream path)",
            "MB/s": round(n / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "ratio": round(ln / n, 4),
            "parity": "zlib inflates it to the input; first 3 MiB as its own stream == oracle"}


def config4(eng):
    n = 128 << 20
    data = synth.silesia_like(synth.SEED_SILESIA, n)
    in_off = np.array([0, n], dtype=np.uint64)
    t, out_off = dev_arrays(eng, data, in_off, 1, 1)

    def run():
        eng.compress_device(t["data"].data_ptr(), t["in_off"].data_ptr(), 1, 1, 1, t["out"].data_ptr(),
                            t["out_off"].data_ptr(), t["out_len"].data_ptr(), t["status"].data_ptr())

    dt = timed(run, 3)
    assert int(t["status"].abs().sum()) == 0
    ln = int(t["out_len"][0])
    got = t["out"][:ln].cpu().numpy().tobytes()
    want = O.compress(data.tobytes(), O.GZIP, O.HUFFMAN)
    assert got == want
    return {"config": "#4 huffman-only gzip, one 128 MiB Silesia-like stream (2049 blocks)",
            "MB/s": round(n / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "ratio": round(ln / n, 4),
            "parity": "whole 128 MiB stream == oracle"}


def config5(eng):
    m, sz = 128, 1 << 20
    data = synth.silesia_like(synth.SEED_SILESIA + 1, m * sz)
    members = [O.compress(data[i * sz:(i + 1) * sz].tobytes(), O.GZIP, 6) for i in range(m)]
    blob = np.frombuffer(b"".join(members), dtype=np.uint8)
    in_off = np.zeros(m + 1, dtype=np.int64)
    np.cumsum([len(x) for x in members], out=in_off[1:])
    out_off = np.arange(m + 1, dtype=np.int64) * sz
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(blob.copy()).to(dev)
    d_inoff, d_outoff = torch.from_numpy(in_off).to(dev), torch.from_numpy(out_off).to(dev)
    d_out = torch.empty(m * sz + 8, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(m, dtype=torch.int64, device=dev)
    d_st = torch.zeros(m, dtype=torch.int32, device=dev)

    def run():
        eng.decompress_device(d_in.data_ptr(), d_inoff.data_ptr(), m, 1, 0, d_out.data_ptr(), d_outoff.data_ptr(),
                              d_len.data_ptr(), d_st.data_ptr())

    dt = timed(run, 3)
    assert int(d_st.abs().sum()) == 0
    assert np.array_equal(d_out[:m * sz].cpu().numpy(), data)
    return {"config": "#5 batched gunzip, 128 x 1 MiB gzip-6 members (CRC/ISIZE verified)",
            "MB/s": round(m * sz / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "parity": "output == input for all members"}


if __name__ == "__main__":
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    which = sys.argv[1:] or ["3", "3s", "4", "5"]
    for w in which:
        print(json.dumps({"3": config3, "3s": config3_stream, "4": config4, "5": config5}[w](eng)), flush=True)
