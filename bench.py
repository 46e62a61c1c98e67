#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DEFLATE engine.

Workload (BASELINE.json configs[1]): raw deflate, level 6, 1 GiB of synthetic
enwik-like text per GPU, cut into independent 65535-byte chunks (one stream each,
bit-exact with the reference's output for that chunk), inputs resident in HBM when
the timed region starts.  One "step" = one pass of the whole compress path over the
batch.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every
rank compresses its own 1 GiB (weak scaling) and the compressed shards are
reassembled on every rank with an RCCL all-gather, as BASELINE.json's north_star asks.

Prints ONE JSON line (rank 0).  `value` = uncompressed MB/s (1e6 B/s) of the whole job.
Extra keys: `roofline` (dominant kernel, HIP-event timed on the launch stream inside
the timed region), `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1),
`decompress` (GPU inflate of the produced streams, same batch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
CHUNK = 65535


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="uncompressed bytes per GPU")
    ap.add_argument("--workload", default="text", choices=["text", "zeros", "silesia"])
    ap.add_argument("--mode", type=int, default=6, help="0 store, 1 huffman-only, 4..9 level")
    ap.add_argument("--container", type=int, default=0, help="0 raw, 1 gzip, 2 zlib")
    ap.add_argument("--chunk", type=int, default=CHUNK)
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL reassembly of the output")
    ap.add_argument("--force-gather", action="store_true", help="run the reassembly path even with one rank (test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chunks", type=int, default=8192)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-decompress", action="store_true",
                    help="skip the inflate leg (one wave per stream: a single huge stream would take minutes)")
    return ap.parse_args()


def make_input(torch, args, rank, device):
    from flate_amd import synth
    n = args.bytes
    if args.workload == "zeros":
        return torch.zeros(n, dtype=torch.uint8, device=device)
    if args.workload == "silesia":
        return torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA + rank, n)).to(device)
    return synth.text_torch(synth.SEED_TEXT + 7919 * rank, n, device=device)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from flate_amd import Engine, synth
    from flate_amd import sharded

    eng = Engine(local)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    eng.set_sync(False)

    data = make_input(torch, args, rank, device)
    n_in = data.numel()
    off_np = synth.split_offsets(n_in, args.chunk)
    n_chunks = len(off_np) - 1
    caps = np.array([(eng.compress_bound(int(off_np[i + 1] - off_np[i]), args.container, args.mode) + 7) & ~7
                     for i in range(n_chunks)], dtype=np.uint64)
    out_off_np = np.zeros(n_chunks + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off_np[1:])
    in_off = torch.from_numpy(off_np.astype(np.int64)).to(device)
    out_off = torch.from_numpy(out_off_np.astype(np.int64)).to(device)
    out = torch.empty(int(out_off_np[-1]) + 8, dtype=torch.uint8, device=device)
    out_len = torch.zeros(n_chunks, dtype=torch.int64, device=device)
    status = torch.zeros(n_chunks, dtype=torch.int32, device=device)
    gather = None
    if (world > 1 or args.force_gather) and not args.no_gather:
        gather = sharded.OutputGather(world, rank, device, int(out_off_np[-1]))

    def step():
        eng.compress_device(data.data_ptr(), in_off.data_ptr(), n_chunks, args.container, args.mode, out.data_ptr(),
                            out_off.data_ptr(), out_len.data_ptr(), status.data_ptr())
        if gather is not None:
            gather.run(out, out_off, out_len)

    def fence():
        if world > 1 or args.force_gather:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_reset()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = status.cpu().numpy()
    lens = out_len.cpu().numpy()
    assert (st == 0).all(), "non-zero chunk status: %s" % np.unique(st)
    n_out = int(lens.sum())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_in * args.steps / dt / 1e6

    # ---- decompress leg: GPU inflate of the streams just produced (same batch) ----
    comp_off_np = np.zeros(n_chunks + 1, dtype=np.int64)
    np.cumsum(lens, out=comp_off_np[1:])
    comp = torch.empty(n_out + 8, dtype=torch.uint8, device=device)
    idx_src = torch.from_numpy(out_off_np[:-1].astype(np.int64)).to(device)
    sharded.compact(out, idx_src, out_len, comp, torch.from_numpy(comp_off_np).to(device))
    comp_off = torch.from_numpy(comp_off_np).to(device)
    dec = torch.empty(n_in + 8, dtype=torch.uint8, device=device)
    dec_len = torch.zeros(n_chunks, dtype=torch.int64, device=device)
    dec_st = torch.zeros(n_chunks, dtype=torch.int32, device=device)

    def dstep():
        eng.decompress_device(comp.data_ptr(), comp_off.data_ptr(), n_chunks, args.container, 0, dec.data_ptr(),
                              in_off.data_ptr(), dec_len.data_ptr(), dec_st.data_ptr())

    dsteps = max(1, min(args.steps, 3))
    ddt, dprof = float("nan"), {}
    if not args.no_decompress:
        dstep()
        fence()
        eng.profile_reset()
        eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(dsteps):
            dstep()
        fence()
        ddt = time.perf_counter() - t0
        dprof = eng.profile_read()
        eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([ddt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ddt = float(t.item())
    roundtrip_ok = args.no_decompress or (int(dec_st.abs().sum().item()) == 0 and bool(torch.equal(dec[:n_in], data)))
    assert roundtrip_ok or args.no_verify, "inflate(deflate(x)) != x"  # --no-verify: kernel tuning experiments only

    result = None
    if rank == 0:
        # dominant kernel of the compress path
        dom = max(prof.items(), key=lambda kv: kv[1][0]) if prof else ("none", (0.0, 1))
        dom_ms = dom[1][0] / max(dom[1][1], 1)
        algo_bytes = n_in + n_out  # read every input byte once, write every output byte once
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": measured_traffic(args, n_in, dom[0]),
                    "kernel_ms": round(dom_ms, 4),
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, data, off_np, out, out_off_np, lens)
        elif not args.no_verify:
            verify_sample(args, data, off_np, out, out_off_np, lens, 64)
        dd = max(dprof.items(), key=lambda kv: kv[1][0]) if dprof else ("none", (0.0, 1))
        result = {
            "metric": "MB/s uncompressed, deflate level 6 compress" if args.mode == 6 else
                      "MB/s uncompressed, deflate mode %d compress" % args.mode,
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "raw deflate level %d, %d MiB synthetic enwik-like text per GPU, %d-byte "
                                   "independent chunks (%d chunks), one MI355X per rank" %
                                   (args.mode, n_in >> 20, args.chunk, n_chunks)
                       if args.workload == "text" else
                       "%s, mode %d, %d MiB per GPU, %d-byte chunks" % (args.workload, args.mode, n_in >> 20, args.chunk),
                       "container": ["raw", "gzip", "zlib"][args.container], "chunk_bytes": args.chunk,
                       "bytes_per_gpu": n_in, "ratio": round(n_out / max(n_in, 1), 4),
                       "gather": "rccl all_gather of compressed shards" if gather is not None else "none"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "decompress": None if args.no_decompress else {"value": round(world * n_in * dsteps / ddt / 1e6, 2), "unit": "MB/s",
                           "ms_per_step": round(ddt / dsteps * 1e3, 3), "kernel": dd[0],
                           "roundtrip_equal": roundtrip_ok,
                           "roofline_frac": round((n_in + n_out) / (ddt / dsteps) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        print(json.dumps(result))
        sys.stdout.flush()
    if gather is not None and rank == 0:
        # the reassembled shard of this rank must be the packed streams themselves
        sizes = gather.run(out, out_off, out_len)
        torch.cuda.synchronize()
        assert sizes[rank] == n_out and torch.equal(gather.shard(rank, sizes), comp[:n_out]), "gather mismatch"
    elif gather is not None:
        gather.run(out, out_off, out_len)
    if world > 1 or args.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    return result


def measured_traffic(args, n_in, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (FETCH_SIZE + WRITE_SIZE, two separate
    rocprofv3 --pmc runs of this very command; profiles/r01_hbm_traffic_1gib.json), when the workload matches;
    PMC counters cannot be read from inside an un-profiled run, so otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic_1gib.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t["workload"] == args.workload and t["bytes_per_gpu"] == n_in and t["mode"] == args.mode and \
                args.chunk == CHUNK and kernel in t["kernels"]:
            k = t["kernels"][kernel]
            return k["fetch_bytes"] + k["write_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    return O


def verify_sample(args, data, off_np, out, out_off_np, lens, k):
    """Parity of a sample of chunks against the CPU oracle (outside the timed region)."""
    import numpy as np
    O = _oracle()
    n_chunks = len(off_np) - 1
    pick = np.unique(np.linspace(0, n_chunks - 1, k).astype(np.int64))
    for i in pick:
        src = data[int(off_np[i]):int(off_np[i + 1])].cpu().numpy().tobytes()
        got = out[int(out_off_np[i]):int(out_off_np[i]) + int(lens[i])].cpu().numpy().tobytes()
        assert got == O.compress(src, args.container, args.mode), "chunk %d differs from the oracle" % i


def cpu_baseline(args, data, off_np, out, out_off_np, lens):
    """The CPU oracle (a C port of the reference algorithm; the Zig reference itself cannot be
    built here) on a bounded sample of the same chunks, one host thread.  Doubles as the
    parity check of those chunks."""
    import numpy as np
    O = _oracle()
    n_chunks = len(off_np) - 1
    k = min(n_chunks, args.cpu_sample_chunks)
    hi = int(off_np[k])
    host = data[:hi].cpu().numpy().tobytes()
    outs_lo, outs_hi = int(out_off_np[0]), int(out_off_np[k])
    gpu_out = out[outs_lo:outs_hi].cpu().numpy()
    t0 = time.perf_counter()
    comp = [O.compress(host[int(off_np[i]):int(off_np[i + 1])], args.container, args.mode) for i in range(k)]
    dt = time.perf_counter() - t0
    for i in range(k):
        a = int(out_off_np[i]) - outs_lo
        assert gpu_out[a:a + int(lens[i])].tobytes() == comp[i], "chunk %d differs from the oracle" % i
    return {"value": round(hi / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d chunks (%d MiB) of the same input, same level, oracle/flate_oracle.c -O3 "
                      "-march=native; every sampled chunk byte-identical to the GPU output" % (k, hi >> 20)}


if __name__ == "__main__":
    main()
