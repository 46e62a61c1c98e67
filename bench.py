#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DEFLATE engine.

Default workload (BASELINE.json configs[1]): raw deflate, level 6, 1 GiB of synthetic
enwik-like text per GPU, cut into independent 65535-byte chunks (one stream each,
bit-exact with the reference's output for that chunk), inputs resident in HBM when
the timed region starts.  One "step" = one pass of the whole compress path over the
batch.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  `--gpus N`
without a launcher re-executes itself under `python -m torch.distributed.run` with N
ranks on 127.0.0.1; under a launcher (WORLD_SIZE set) it checks that the world size is N.
Every rank compresses its own shard (weak scaling, no data-path collective) and the
compressed shards are reassembled on every rank over RCCL, as BASELINE.json's
north_star asks.

Other configurations of BASELINE.json (parity-test cases, not the headline):
  --config 3   gzip level 9, TAR-like 177,244,160 bytes per rank, 65535-byte chunks
  --config 4   huffman-only gzip, ONE 128 MiB Silesia-like stream per rank, RCCL reassembly
  --config 5   batched gunzip of 128 x 1 MiB gzip-6 members per rank (the members of the whole
               job are sharded over the ranks by ISIZE), outputs == inputs
The flags of the reference's bin/deflate_bench.zig:101-114 are accepted too: -l LEVEL, -g (gzip),
-z (zlib), -o FILE (write rank 0's first stream), -c (the same to stdout), plus an input file.

Prints ONE JSON line (rank 0).  `value` = uncompressed MB/s (1e6 B/s) of the whole job.
Extra keys: `roofline` (dominant kernel, HIP-event timed on the launch stream inside the
timed region), `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1),
`cpu_baseline_all_cores`, `e2e_host` (host buffers in and out over PCIe), `decompress`
(GPU inflate of the produced streams, same batch), `other_workloads` (all-zero, Silesia-like,
1 MiB streams; smaller buffers, same run).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime for this process: torch (imported later) brings its own (flate_amd/_capi.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
CHUNK = 65535
MODE_NAMES = {0: "store", 1: "huffman-only"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--bytes", type=int, default=None, help="uncompressed bytes per GPU (default: the configuration's)")
    ap.add_argument("--workload", default=None, choices=["text", "zeros", "silesia", "tar"])
    ap.add_argument("--mode", type=int, default=None, help="0 store, 1 huffman-only, 4..9 level")
    ap.add_argument("-l", dest="level", type=int, default=None, help="compression level 4..9 (deflate_bench.zig -l)")
    ap.add_argument("--container", type=int, default=None, help="0 raw, 1 gzip, 2 zlib")
    ap.add_argument("-g", dest="gzip", action="store_true", help="gzip container (deflate_bench.zig -g)")
    ap.add_argument("-z", dest="zlib", action="store_true", help="zlib container (deflate_bench.zig -z)")
    ap.add_argument("-o", dest="out_file", default=None, help="write rank 0's first output stream to this file")
    ap.add_argument("-c", dest="to_stdout", action="store_true", help="write rank 0's first output stream to stdout")
    ap.add_argument("input_file", nargs="?", default=None, help="compress this file instead of synthetic data")
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL reassembly of the output")
    ap.add_argument("--force-gather", action="store_true", help="run the reassembly path even with one rank (test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chunks", type=int, default=4096)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-decompress", action="store_true",
                    help="skip the inflate leg (one wave per stream: a single huge stream would take minutes)")
    ap.add_argument("--no-extras", action="store_true", help="skip other_workloads / e2e_host / all-core baseline")
    a = ap.parse_args()
    # the configuration's defaults
    d = {2: dict(workload="text", mode=6, container=0, chunk=CHUNK, bytes=1 << 30),
         3: dict(workload="tar", mode=9, container=1, chunk=CHUNK, bytes=177_244_160),
         4: dict(workload="silesia", mode=1, container=1, chunk=128 << 20, bytes=128 << 20),
         5: dict(workload="silesia", mode=6, container=1, chunk=1 << 20, bytes=128 << 20)}[a.config]
    if a.level is not None:
        a.mode = a.level
    if a.gzip:
        a.container = 1
    if a.zlib:
        a.container = 2
    for k, v in d.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    a.headline = a.config == 2 and a.workload == "text" and a.mode == 6 and a.container == 0 and a.chunk == CHUNK \
        and a.input_file is None
    return a


def maybe_spawn(args):
    """`--gpus N` (N > 1) outside a launcher: run N ranks of this script under torch.distributed.run."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def make_input(torch, args, rank, device):
    import numpy as np
    from flate_amd import synth
    n = args.bytes
    if args.input_file:
        with open(args.input_file, "rb") as f:
            buf = np.frombuffer(f.read(), dtype=np.uint8)
        args.bytes = buf.size
        return torch.from_numpy(buf.copy()).to(device)
    if args.workload == "zeros":
        return torch.zeros(n, dtype=torch.uint8, device=device)
    if args.workload == "silesia":
        return torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA + rank, n)).to(device)
    if args.workload == "tar":
        return torch.from_numpy(synth.tar_like(synth.SEED_TAR + rank, n)).to(device)
    return synth.text_torch(synth.SEED_TEXT + 7919 * rank, n, device=device)


class CompressJob:
    """Device-resident buffers of one compress batch + the step function."""

    def __init__(self, torch, eng, data, chunk, container, mode):
        import numpy as np
        from flate_amd import synth
        self.torch, self.eng, self.data = torch, eng, data
        self.container, self.mode = container, mode
        device = data.device
        self.n_in = data.numel()
        self.off_np = synth.split_offsets(self.n_in, chunk)
        self.n_chunks = len(self.off_np) - 1
        caps = np.array([(eng.compress_bound(int(self.off_np[i + 1] - self.off_np[i]), container, mode) + 7) & ~7
                         for i in range(self.n_chunks)], dtype=np.uint64)
        self.out_off_np = np.zeros(self.n_chunks + 1, dtype=np.uint64)
        np.cumsum(caps, out=self.out_off_np[1:])
        self.in_off = torch.from_numpy(self.off_np.astype(np.int64)).to(device)
        self.out_off = torch.from_numpy(self.out_off_np.astype(np.int64)).to(device)
        self.out = torch.empty(int(self.out_off_np[-1]) + 8, dtype=torch.uint8, device=device)
        self.out_len = torch.zeros(self.n_chunks, dtype=torch.int64, device=device)
        self.status = torch.zeros(self.n_chunks, dtype=torch.int32, device=device)
        # the layout repeats every step: plan it once, then a step only enqueues kernels
        # (inputs beyond 65535 bytes at levels 4..9 take the whole-stream path, which is not plannable)
        self.plan = None
        try:
            self.plan = eng.plan_compress(self.off_np, self.out_off_np, container, mode)
        except Exception:
            self.plan = None

    def step(self):
        if self.plan is not None:
            self.eng.compress_planned(self.plan, self.data.data_ptr(), self.out.data_ptr(), self.out_len.data_ptr(),
                                      self.status.data_ptr())
            return
        self.eng.compress_device(self.data.data_ptr(), self.in_off.data_ptr(), self.n_chunks, self.container, self.mode,
                                 self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(),
                                 self.status.data_ptr())

    def results(self):
        import numpy as np
        st = self.status.cpu().numpy()
        lens = self.out_len.cpu().numpy()
        # 102 = FLATE_HIP_ST_REFERENCE_Q1_STREAM: the reference's own bytes, which do not inflate to the input (include/flate_hip.h)
        assert ((st == 0) | (st == 102)).all(), "non-zero chunk status: %s" % np.unique(st)
        self.q1_chunks = np.flatnonzero(st == 102)
        return lens

    def roundtrip(self, lens):
        """GPU inflate of the streams just produced against the input: every stream of status 0 gives its chunk back; the
        streams of status 102 (the reference's Q1 streams) are the only ones allowed not to.  -> (ok, n_q1)"""
        import numpy as np
        torch = self.torch
        comp, comp_off, _ = self.packed(lens)
        inf = InflateJob(torch, self.eng, comp, comp_off, self.n_chunks, self.in_off, self.n_in, self.container)
        inf.step()
        torch.cuda.synchronize()
        bad = np.zeros(self.n_chunks, dtype=bool)
        bad[np.flatnonzero(inf.dec_st.cpu().numpy() != 0)] = True
        diff = torch.nonzero(inf.dec[:self.n_in] != self.data).flatten()
        if diff.numel():
            bad[np.unique(np.searchsorted(self.off_np, diff.cpu().numpy(), side="right") - 1)] = True
        q1 = np.zeros(self.n_chunks, dtype=bool)
        q1[self.q1_chunks] = True
        return bool((bad == q1).all()), int(q1.sum())

    def packed(self, lens):
        """The produced streams back to back (device) + their offsets."""
        import numpy as np
        from flate_amd import sharded
        torch = self.torch
        n_out = int(lens.sum())
        comp_off_np = np.zeros(self.n_chunks + 1, dtype=np.int64)
        np.cumsum(lens, out=comp_off_np[1:])
        comp = torch.empty(n_out + 8, dtype=torch.uint8, device=self.data.device)
        idx_src = torch.from_numpy(self.out_off_np[:-1].astype(np.int64)).to(self.data.device)
        comp_off = torch.from_numpy(comp_off_np).to(self.data.device)
        sharded.compact(self.out, idx_src, self.out_len, comp, comp_off, engine=self.eng)
        return comp, comp_off, n_out


class InflateJob:
    def __init__(self, torch, eng, comp, comp_off, n_streams, out_off, n_out, container):
        self.eng, self.comp, self.comp_off, self.n, self.out_off, self.container = eng, comp, comp_off, n_streams, out_off, container
        dev = comp.device
        self.dec = torch.empty(n_out + 8, dtype=torch.uint8, device=dev)
        self.dec_len = torch.zeros(n_streams, dtype=torch.int64, device=dev)
        self.dec_st = torch.zeros(n_streams, dtype=torch.int32, device=dev)

    def step(self):
        self.eng.decompress_device(self.comp.data_ptr(), self.comp_off.data_ptr(), self.n, self.container, 0,
                                   self.dec.data_ptr(), self.out_off.data_ptr(), self.dec_len.data_ptr(),
                                   self.dec_st.data_ptr())


def timed(torch, dist, eng, step, steps, warmup, world):
    """warmup untimed steps, then `steps` steps bracketed by barrier + synchronize; max over ranks."""
    def fence():
        if dist is not None and dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    eng.profile_reset()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, prof


def roofline_of(prof, steps, algo_bytes, traffic=None):
    dom = max(prof.items(), key=lambda kv: kv[1][0]) if prof else ("none", (0.0, 1))
    dom_ms = dom[1][0] / max(dom[1][1], 1)
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic(dom[0]) if traffic else None,
            "kernel_ms": round(dom_ms, 4),
            "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items())}}


def main():
    args = parse()
    maybe_spawn(args)
    # ONE JSON line on stdout: libraries that print banners from C (RCCL does) get stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if "WORLD_SIZE" in os.environ and args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
        assert dist.get_world_size() == world, "RCCL world size %d != %d" % (dist.get_world_size(), world)

    from flate_amd import Engine, sharded

    # Everything -- the engine's kernels, torch's copies, the RCCL collectives -- is ordered on ONE
    # explicit (non-default) stream: the engine treats a NULL stream as "its own", which torch and
    # RCCL would not be ordered against.
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    eng = Engine(local)
    eng.set_stream(stream.cuda_stream)
    eng.set_sync(False)

    if args.config == 5:
        result = run_config5(args, torch, dist if use_dist else None, eng, world, rank, device)
    else:
        result = run_compress(args, torch, dist if use_dist else None, eng, world, rank, device)
    if rank == 0:
        if getattr(args, "stdout_stream", None) is not None:
            sys.stderr.write(json.dumps(result) + "\n")
            os.write(real_stdout, args.stdout_stream)
        else:
            os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


def run_compress(args, torch, dist, eng, world, rank, device):
    import numpy as np
    from flate_amd import sharded
    data = make_input(torch, args, rank, device)
    job = CompressJob(torch, eng, data, args.chunk, args.container, args.mode)
    n_in = job.n_in
    gather = None
    if dist is not None and not args.no_gather:
        job.step()
        for algo in (None, "all_gather"):  # grouped send/recv first; one padded all-gather if the backend refuses it
            try:
                gather = sharded.OutputGather(world, rank, device, int(job.out_off_np[-1]), engine=eng, algo=algo)
                gather.calibrate(job.out, job.out_off, job.out_len)  # one host sync, outside the timed region
                gather.run(job.out, job.out_off, job.out_len)
                torch.cuda.synchronize()
                break
            except Exception as e:  # noqa: BLE001 -- reported, and the other exchange form is tried
                sys.stderr.write("rank %d: output gather (%s) failed: %r\n" % (rank, algo or "p2p", e))
                if algo == "all_gather":
                    raise

    def step():
        job.step()
        if gather is not None:
            gather.run(job.out, job.out_off, job.out_len)

    dt, prof = timed(torch, dist, eng, step, args.steps, args.warmup, world)
    lens = job.results()
    n_out = int(lens.sum())
    if gather is not None:
        assert not gather.overflowed(), "a packed shard outgrew the calibrated gather width"
    ms_per_step = dt / args.steps * 1e3
    value = world * n_in * args.steps / dt / 1e6

    # ---- decompress leg: GPU inflate of the streams just produced (same batch) ----
    comp, comp_off, _ = job.packed(lens)
    dsteps = max(1, min(args.steps, 3))
    ddt, dprof, roundtrip_ok = float("nan"), {}, True
    if not args.no_decompress:
        inf = InflateJob(torch, eng, comp, comp_off, job.n_chunks, job.in_off, n_in, args.container)
        ddt, dprof = timed(torch, dist, eng, inf.step, dsteps, 1, world)
        roundtrip_ok = int(inf.dec_st.abs().sum().item()) == 0 and bool(torch.equal(inf.dec[:n_in], data))
        assert roundtrip_ok or args.no_verify, "inflate(deflate(x)) != x"  # --no-verify: kernel tuning experiments only

    nsw = None
    if world > 1 and args.headline and not args.no_extras:
        nsw = north_star_workloads(args, torch, dist, eng, world, rank, device)  # (every rank: its timing is a collective)
    result = None
    if rank == 0:
        algo_bytes = n_in + n_out  # read every input byte once, write every output byte once
        roofline = roofline_of(prof, args.steps, algo_bytes, lambda k: measured_traffic(args, n_in, k))
        cpu = cpu_all = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, data, job.off_np, job.out, job.out_off_np, lens)
            if not args.no_extras:
                cpu_all = cpu_baseline_all_cores(args, data, job.off_np)
        elif not args.no_verify:
            verify_sample(args, data, job.off_np, job.out, job.out_off_np, lens, 64)
        dd = max(dprof.items(), key=lambda kv: kv[1][0]) if dprof else ("none", (0.0, 1))
        mode_name = MODE_NAMES.get(args.mode, "level %d" % args.mode)
        wl = {"text": "synthetic enwik-like text", "zeros": "all-zero bytes", "silesia": "synthetic Silesia-like mix",
              "tar": "synthetic TAR-like buffer"}[args.workload] if not args.input_file else os.path.basename(args.input_file)
        result = {
            "metric": "MB/s uncompressed, deflate level 6 compress" if args.mode == 6 else
                      "MB/s uncompressed, deflate %s compress" % mode_name,
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if not args.input_file else "file",
            "config": {"workload": "%s deflate %s, %d MiB %s per GPU, %d-byte independent chunks (%d chunks), one "
                                   "MI355X per rank (BASELINE.json configs[%d])" %
                                   (["raw", "gzip", "zlib"][args.container], mode_name, n_in >> 20, wl, args.chunk,
                                    job.n_chunks, args.config - 1),
                       "container": ["raw", "gzip", "zlib"][args.container], "chunk_bytes": args.chunk,
                       "bytes_per_gpu": n_in, "ratio": round(n_out / max(n_in, 1), 4),
                       "gather": gather.form if gather is not None else "none"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "decompress": None if args.no_decompress else {
                "value": round(world * n_in * dsteps / ddt / 1e6, 2), "unit": "MB/s",
                "ms_per_step": round(ddt / dsteps * 1e3, 3), "kernel": dd[0], "roundtrip_equal": roundtrip_ok,
                "roofline_frac": round((n_in + n_out) / (ddt / dsteps) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if args.headline and world == 1 and not args.no_extras:
            result["e2e_host"] = e2e_host(torch, eng, data, job)
            result["other_workloads"] = other_workloads(args, torch, eng, device)
        elif nsw is not None:
            result["other_workloads"] = nsw  # N > 1: the Silesia-like and all-zero buffers of the north star, whole-job rates
        if args.out_file or args.to_stdout:
            first = job.out[: int(lens[0])].cpu().numpy().tobytes()
            if args.out_file:
                with open(args.out_file, "wb") as f:
                    f.write(first)
            else:
                args.stdout_stream = first  # -c: the stream owns stdout, the JSON line goes to stderr
    if gather is not None:
        # the reassembled shard of this rank must be the packed streams themselves
        gather.run(job.out, job.out_off, job.out_len)
        torch.cuda.synchronize()
        if rank == 0:
            sizes = gather.sizes_host()
            assert sizes[rank] == n_out and torch.equal(gather.shard(rank, sizes), comp[:n_out]), "gather mismatch"
    return result


def run_config5(args, torch, dist, eng, world, rank, device):
    """BASELINE.json configs[4]: batched gunzip of 1 MiB gzip members.  The job's members (128 per
    rank) are sharded over the ranks by ISIZE; every rank builds and inflates its
    own range, no collective in the data path."""
    import numpy as np
    from flate_amd import sharded, synth
    per, sz = 128, args.chunk
    total = per * world
    lo, hi = sharded.shard_ranges([sz] * total, world)[rank]
    m = hi - lo
    raw = torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA + 1 + rank, m * sz)).to(device)
    mk = CompressJob(torch, eng, raw, sz, 1, 6)  # the members: gzip level 6 of each slice (this engine, == oracle)
    mk.step()
    torch.cuda.synchronize()
    lens = mk.results()
    comp, comp_off, n_comp = mk.packed(lens)
    inf = InflateJob(torch, eng, comp, comp_off, m, mk.in_off, m * sz, 1)
    dt, prof = timed(torch, dist, eng, inf.step, args.steps, args.warmup, world)
    ok = int(inf.dec_st.abs().sum().item()) == 0 and bool(torch.equal(inf.dec[: m * sz], raw))
    assert ok, "gunzip output != input"
    if rank != 0:
        return None
    if not args.no_verify:  # the members are what the reference would have written
        O = _oracle()
        for i in (0, m // 2, m - 1):
            a = int(comp_off[i].item())
            got = comp[a:a + int(lens[i])].cpu().numpy().tobytes()
            assert got == O.compress(raw[i * sz:(i + 1) * sz].cpu().numpy().tobytes(), O.GZIP, 6)
    n_out = m * sz
    return {"metric": "MB/s uncompressed, batched gunzip (inflate)", "value": round(world * n_out * args.steps / dt / 1e6, 2),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "batched gunzip of %d x %d-byte gzip level-6 members of a Silesia-like buffer per GPU, "
                                   "sharded by ISIZE (BASELINE.json configs[4])" % (m, sz),
                       "members_per_gpu": m, "ratio": round(n_comp / n_out, 4), "outputs_equal_inputs": ok},
            "roofline": roofline_of(prof, args.steps, n_comp + n_out), "cpu_baseline": None}


def measured_traffic(args, n_in, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, two separate
    rocprofv3 --pmc runs of this very command; profiles/r0N_hbm_traffic_1gib.json), when the workload matches;
    PMC counters cannot be read from inside an un-profiled run, so otherwise null."""
    for name in ("r06_hbm_traffic_1gib.json", "r05_hbm_traffic_1gib.json", "r04_hbm_traffic_1gib.json", "r03_hbm_traffic_1gib.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
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
        if int(off_np[i + 1] - off_np[i]) > (8 << 20):
            continue  # (a single huge stream: the -m gpu tests compare those with the oracle)
        src = data[int(off_np[i]):int(off_np[i + 1])].cpu().numpy().tobytes()
        got = out[int(out_off_np[i]):int(out_off_np[i]) + int(lens[i])].cpu().numpy().tobytes()
        assert got == O.compress(src, args.container, args.mode), "chunk %d differs from the oracle" % i


def cpu_baseline(args, data, off_np, out, out_off_np, lens):
    """The CPU oracle (a C port of the reference algorithm; the Zig reference itself cannot be
    built here) on a bounded sample of the same chunks, one host thread.  Doubles as the
    parity check of those chunks."""
    O = _oracle()
    n_chunks = len(off_np) - 1
    # ~10-20 s of CPU work: the slower the level, the fewer chunks
    k = min(n_chunks, args.cpu_sample_chunks if args.mode <= 6 else max(args.cpu_sample_chunks // 4, 1))
    while k > 1 and int(off_np[k]) > (512 << 20):
        k //= 2
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
            "sample": "first %d chunks (%d MiB) of the same input, same mode, oracle/flate_oracle.c -O3 "
                      "-march=native; every sampled chunk byte-identical to the GPU output" % (k, hi >> 20)}


def _cpu_worker(job):
    blob, offs, container, mode = job
    O = _oracle()
    t0 = time.perf_counter()
    n = 0
    for i in range(len(offs) - 1):
        n += len(O.compress(blob[offs[i]:offs[i + 1]], container, mode))
    return time.perf_counter() - t0, n


def cpu_baseline_all_cores(args, data, off_np):
    """The same oracle over the same independent chunks with one worker process per host core
    (SURVEY.md 8d (ii)): the aggregate a multi-threaded caller of the reference would see."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(cores, 64))
    per = 256 if args.mode <= 6 else 64  # chunks per worker (~0.4 s each at level 6)
    n_chunks = len(off_np) - 1
    workers = min(workers, n_chunks)
    per = max(1, min(per, n_chunks // workers))
    k = per * workers
    hi = int(off_np[k])
    host = data[:hi].cpu().numpy().tobytes()
    jobs = []
    for w in range(workers):
        a, b = int(off_np[w * per]), int(off_np[(w + 1) * per])
        jobs.append((host[a:b], [int(off_np[w * per + i]) - a for i in range(per + 1)], args.container, args.mode))
    ctx = mp.get_context("fork")
    import resource as _res
    _r = _res.getrusage(_res.RUSAGE_CHILDREN)
    _ru0 = _r.ru_utime + _r.ru_stime
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        pool.map(_cpu_worker, jobs)
    dt = time.perf_counter() - t0
    # what the box really gives: its CPU quota (cgroup), and the cores the workers kept busy (CPU seconds / wall)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
            quota = None if a == "max" else round(int(a) / int(b), 1)
    except (OSError, ValueError):
        pass
    import resource
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    busy = round((ru.ru_utime + ru.ru_stime - _ru0) / dt, 1) if dt > 0 else None
    return {"value": round(hi / dt / 1e6, 2), "unit": "MB/s", "cores": workers, "host_cores_visible": cores,
            "cpu_quota_cores": quota, "cores_kept_busy": busy, "kind": "port",
            "sample": "%d chunks (%d MiB) of the same input over %d worker processes, wall time incl. process start; "
                      "`cores` = worker processes, `cores_kept_busy` = their CPU seconds per second of wall time"
                      % (k, hi >> 20, workers)}


def e2e_host(torch, eng, data, job):
    """Host buffers in and out (FLATE_HIP_MEM_HOST): H2D + kernels + D2H over PCIe, 256 MiB of the same
    input, straight through the C ABI.  Never `value`."""
    import numpy as np
    from flate_amd import _capi
    L = _capi.lib()
    n = min(data.numel(), 256 << 20)
    k = int(np.searchsorted(job.off_np, n, side="right")) - 1
    hi = int(job.off_np[k])
    host = data[:hi].cpu().numpy()
    in_off = job.off_np[:k + 1].astype(np.uint64)
    out_off = job.out_off_np[:k + 1].astype(np.uint64)
    out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
    out_len = np.zeros(k, dtype=np.uint64)
    status = np.zeros(k, dtype=np.int32)

    def comp():
        rc = L.flate_hip_compress_batch(eng._h, host.ctypes.data, in_off.ctypes.data, k, job.container, job.mode,
                                        out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data,
                                        _capi.MEM_HOST)
        assert rc == 0 and not status.any()

    def best(f, warm=3, reps=5):
        """Three untimed calls, then the MEDIAN of five timed ones (and their minimum beside it).  Round 5: the best of a few
        calls used to be the second call of the process -- the only one in which the DMA engine's copies of the old pinned
        path overlapped (profiles/r05_host_path.txt); a library is used in its steady state."""
        for _ in range(warm):
            f()
        ts = []
        for _ in range(reps):
            t_ = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t_)
        ts.sort()
        mins.append(ts[0])
        return ts[len(ts) // 2]

    mins = []
    dt = best(comp)
    # inflate: the streams packed back to back, outputs into the original layout
    lens = out_len.astype(np.int64)
    c_off = np.zeros(k + 1, dtype=np.uint64)
    np.cumsum(lens, out=c_off[1:].view(np.int64))
    packed = np.concatenate([out[int(out_off[i]):int(out_off[i]) + int(lens[i])] for i in range(k)])
    dec = np.zeros(hi + 8, dtype=np.uint8)
    dlen = np.zeros(k, dtype=np.uint64)

    def decomp():
        rc = L.flate_hip_decompress_batch(eng._h, packed.ctypes.data, c_off.ctypes.data, k, job.container, 0,
                                          dec.ctypes.data, in_off.ctypes.data, dlen.ctypes.data, status.ctypes.data,
                                          None, _capi.MEM_HOST)
        assert rc == 0 and not status.any()

    dt2 = best(decomp)
    assert np.array_equal(dec[:hi], host)
    # the same call with pinned buffers: sub-batches, H2D / kernels / D2H overlapped on three streams
    p_in = torch.from_numpy(host).pin_memory()
    p_out = torch.zeros(out.size, dtype=torch.uint8).pin_memory()

    def comp_pinned():
        rc = L.flate_hip_compress_batch(eng._h, p_in.data_ptr(), in_off.ctypes.data, k, job.container, job.mode,
                                        p_out.data_ptr(), out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data,
                                        _capi.MEM_HOST)
        assert rc == 0 and not status.any()

    dt3 = best(comp_pinned)
    po = p_out.numpy()
    for i in (0, k // 2, k - 1):
        a, b = int(out_off[i]), int(out_off[i]) + int(out_len[i])
        assert np.array_equal(po[a:b], out[a:b])
    pk = torch.from_numpy(packed).pin_memory()
    pd = torch.zeros(dec.size, dtype=torch.uint8).pin_memory()

    def decomp_pinned():
        rc = L.flate_hip_decompress_batch(eng._h, pk.data_ptr(), c_off.ctypes.data, k, job.container, 0,
                                          pd.data_ptr(), in_off.ctypes.data, dlen.ctypes.data, status.ctypes.data,
                                          None, _capi.MEM_HOST)
        assert rc == 0 and not status.any()

    dt4 = best(decomp_pinned)
    assert np.array_equal(pd.numpy()[:hi], host)
    return {"compress_MBps": round(hi / dt / 1e6, 1), "decompress_MBps": round(hi / dt2 / 1e6, 1),
            "compress_pinned_overlapped_MBps": round(hi / dt3 / 1e6, 1),
            "decompress_pinned_overlapped_MBps": round(hi / dt4 / 1e6, 1), "bytes": hi,
            "method": "median of 5 calls after 3 warm-up calls",
            "best_call_MBps": {"compress": round(hi / mins[0] / 1e6, 1), "decompress": round(hi / mins[1] / 1e6, 1),
                               "compress_pinned": round(hi / mins[2] / 1e6, 1), "decompress_pinned": round(hi / mins[3] / 1e6, 1)},
            "note": "flate_hip_*_batch(MEM_HOST) over PCIe (57 GB/s one way; both ways at once 48 GB/s each in some processes, "
                    "28.6 in others: profiles/r05_host_path.txt).  Pinned: sub-batches of 1024 chunks, input and tables on a copy "
                    "stream beside the kernels, the produced bytes of every slot home by a copy kernel into the caller's pinned "
                    "memory (no DMA copy that waits for a sub-batch's kernels: it would sit in the engine's queue in front of the "
                    "next sub-batch's input).  Pageable: pinned mirrors, filled and emptied by host threads a sub-batch at a "
                    "time beside the GPU's work; inflate stages once each way (pinned: sub-batches of at least 3072 streams: a "
                    "sub-batch has to fill the chip)"}


def other_workloads(args, torch, eng, device):
    """The north star's other inputs, smaller buffers, same run: level 6 raw."""
    import numpy as np
    from flate_amd import synth
    res = {}
    cases = [("zeros_256MiB_64KiB_chunks", lambda: torch.zeros(256 << 20, dtype=torch.uint8, device=device), CHUNK),
             ("silesia_like_128MiB_64KiB_chunks",
              lambda: torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA, 128 << 20)).to(device), CHUNK),
             ("text_256MiB_1MiB_streams", lambda: synth.text_torch(synth.SEED_TEXT, 256 << 20, device=device), 1 << 20)]
    for name, mk, chunk in cases:
        d = mk()
        job = CompressJob(torch, eng, d, chunk, 0, 6)
        dt, prof = timed(torch, None, eng, job.step, 2, 1, 1)
        lens = job.results()
        ok = None
        if not args.no_verify:
            a = argparse.Namespace(container=0, mode=6)
            verify_sample(a, d, job.off_np, job.out, job.out_off_np, lens, 8 if chunk == CHUNK else 2)
            ok = True
        res[name] = {"MBps": round(d.numel() * 2 / dt / 1e6, 1), "ratio": round(float(lens.sum()) / d.numel(), 4),
                     "pipeline_frac": _frac(d.numel() + float(lens.sum()), dt / 2 * 1e3), "sampled_chunks_equal_oracle": ok}
        if not args.no_verify:
            # every stream back through GPU inflate: status 0 <=> the chunk comes back; status 102 = the reference's own Q1
            # streams (its bytes, replicated: they lose or repeat a match's bytes); with FLATE_HIP_DEFLATE_REPAIR_Q1 every
            # stream comes back
            rt, nq1 = job.roundtrip(lens)
            assert rt, "a stream of status 0 does not inflate to its chunk (or a status-102 stream does)"
            res[name].update({"roundtrip_equal": rt, "reference_q1_streams": nq1})
            if nq1:
                eng.set_flags(1)
                try:
                    job.step()
                    lens2 = job.results()
                    rt2, nq2 = job.roundtrip(lens2)
                finally:
                    eng.set_flags(0)
                assert rt2 and nq2 == 0, "FLATE_HIP_DEFLATE_REPAIR_Q1: a stream does not inflate to its chunk"
                res[name]["roundtrip_equal_with_repair_q1"] = True
        del job, d
    res.update(one_stream_inflate(args, torch, eng, device))
    res.update(baseline_configs(args, torch, eng, device))
    return res


def north_star_workloads(args, torch, dist, eng, world, rank, device, make_job=None, timer=None, make_inflate=None):
    """N > 1: the north star's other inputs -- a Silesia-like mix and an all-zero buffer per rank (level 6, raw, 64 KiB chunks) --
    and their inflate, in the same timed loop as the headline: barrier, K steps, barrier, MAX over the ranks; the values are
    whole-job rates (every rank's bytes / the slowest rank's time).  Called by EVERY rank (the timing is a collective); only
    rank 0's dict goes into the line.  `make_job` / `timer` / `make_inflate` are the seams of tests/test_sharded_gloo.py, which
    runs this with two CPU ranks over gloo."""
    from flate_amd import synth
    make_job = make_job or (lambda d: CompressJob(torch, eng, d, CHUNK, 0, 6))
    timer = timer or (lambda step, steps, warmup: timed(torch, dist, eng, step, steps, warmup, world))
    res = {}
    cases = [("zeros_256MiB_64KiB_chunks", lambda: torch.zeros(256 << 20, dtype=torch.uint8, device=device)),
             ("silesia_like_128MiB_64KiB_chunks",
              lambda: torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA + 16 * rank, 128 << 20)).to(device))]
    for name, mk in cases:
        d = mk()
        n = d.numel()
        job = make_job(d)
        dt, prof = timer(job.step, 2, 1)
        lens = job.results()
        n_out = int(lens.sum())
        k, kms = _dominant(prof, 2)
        entry = {"MBps": round(world * n * 2 / dt / 1e6, 1), "n_gpus": world, "bytes_per_gpu": n, "ratio_rank0": round(n_out / n, 4),
                 "kernel": k, "roofline_frac": _frac(n + n_out, kms)}
        if not args.no_decompress:
            if make_inflate is not None:
                inf = make_inflate(job, lens, d)
            else:
                comp, comp_off, _ = job.packed(lens)
                inf = InflateJob(torch, eng, comp, comp_off, job.n_chunks, job.in_off, n, 0)
            ddt, dprof = timer(inf.step, 2, 1)
            entry["decompress_MBps"] = round(world * n * 2 / ddt / 1e6, 1)
            if make_inflate is None:
                ok = int(inf.dec_st.abs().sum().item()) == 0 and bool(torch.equal(inf.dec[:n], d))
                assert ok or args.no_verify, "%s: inflate(deflate(x)) != x on rank %d" % (name, rank)
                entry["roundtrip_equal_rank0"] = ok
            del inf
        res[name] = entry
        del job, d
    return res


def _frac(n_bytes, ms):
    return round(n_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None


def _dominant(prof, steps):
    if not prof:
        return "none", 0.0
    k, v = max(prof.items(), key=lambda kv: kv[1][0])
    return k, v[0] / steps


def baseline_configs(args, torch, eng, device):
    """BASELINE.json configs[2], [3], [4] at their single-GPU sizes, and level 9 on the run-heavy inputs, in the same
    run as the headline: each with its time, its dominant kernel and that kernel's fraction of the HBM roofline
    (algorithmic bytes = bytes in + bytes out)."""
    import numpy as np
    from flate_amd import synth
    res = {}

    def compress_case(name, data, chunk, container, mode, steps=2, sample=2, roundtrip=False):
        job = CompressJob(torch, eng, data, chunk, container, mode)
        dt, prof = timed(torch, None, eng, job.step, steps, 2 if steps >= 10 else 1, 1)
        method = "%d steps" % steps
        best_ms = None
        if steps >= 10:
            # a step of about a millisecond, timed right after seconds of host-side preparation: the GPU's clocks are still on
            # their way up during the first run (measured: every kernel 10-15 % slower than in the runs that follow).  Three
            # runs: the MEDIAN run is the entry's figure (as everywhere else in this line), the fastest stands beside it.
            runs = [(dt, prof)] + [timed(torch, None, eng, job.step, steps, 1, 1) for _ in range(2)]
            runs.sort(key=lambda r: r[0])
            best_ms = round(runs[0][0] / steps * 1e3, 3)
            dt, prof = runs[1]
            method = "median of three runs of %d steps (best_ms: the fastest run)" % steps
        lens = job.results()
        n, n_out = data.numel(), int(lens.sum())
        ok = None
        if not args.no_verify and sample:
            verify_sample(argparse.Namespace(container=container, mode=mode), data, job.off_np, job.out, job.out_off_np, lens, sample)
            ok = True
        ms = dt / steps * 1e3
        k, kms = _dominant(prof, steps)
        # roofline_frac: the dominant KERNEL's (its time alone); pipeline_frac: the same bytes over the whole step
        res[name] = {"MBps": round(n / ms / 1e3, 1), "ms": round(ms, 3), "ratio": round(n_out / n, 4), "kernel": k,
                     "kernel_ms": round(kms, 3), "roofline_frac": _frac(n + n_out, kms), "pipeline_frac": _frac(n + n_out, ms),
                     "kernels_ms": {kk: round(v[0] / steps, 3) for kk, v in sorted(prof.items())},
                     "sampled_chunks_equal_oracle": ok, "method": method}
        if best_ms is not None:
            res[name]["best_ms"] = best_ms
        if roundtrip and not args.no_verify:
            rt, nq1 = job.roundtrip(lens)  # (GPU inflate checks CRC-32 and ISIZE of a gzip stream as well)
            assert rt, name + ": inflate(stream) != input"
            res[name].update({"roundtrip_equal": rt, "reference_q1_streams": nq1})
        return job, lens

    # configs[2]: gzip level 9 of the TAR-like buffer (177,244,160 bytes: the size of the reference's ziglang.tar), 65535-byte members
    tar = torch.from_numpy(synth.tar_like(synth.SEED_TAR, synth.TAR_BYTES)).to(device)
    compress_case("config3_gzip_l9_tar", tar, CHUNK, 1, 9)
    # ... and as ONE gzip -9 stream, the shape bin/gzip.zig produces (the whole-stream path; no oracle sample: the CPU port takes
    # a quarter of a minute for it -- the stream goes back through GPU inflate, which checks CRC-32 and ISIZE, instead)
    compress_case("config3_gzip_l9_tar_one_stream", tar, synth.TAR_BYTES, 1, 9, steps=1, sample=0, roundtrip=True)
    del tar
    # configs[3]: huffman-only, one 128 MiB buffer = one stream (per GPU), and its inflate
    sil = torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA, 128 << 20)).to(device)
    job, lens = compress_case("config4_huffman_only_128MiB_stream", sil, 128 << 20, 1, 1, steps=10, sample=1)  # (a step is about 1 ms: ten of them, not three of which the first is cold)
    del job
    # configs[4]: gunzip of 128 members of 1 MiB (per GPU; the members of `--config 5`: same seed)
    del sil
    sil = torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA + 1, 128 << 20)).to(device)
    mk = CompressJob(torch, eng, sil, 1 << 20, 1, 6)
    mk.step()
    torch.cuda.synchronize()
    lens = mk.results()
    comp, comp_off, n_comp = mk.packed(lens)
    inf = InflateJob(torch, eng, comp, comp_off, mk.n_chunks, mk.in_off, sil.numel(), 1)
    dt, prof = timed(torch, None, eng, inf.step, 3, 1, 1)
    ok = int(inf.dec_st.abs().sum().item()) == 0 and bool(torch.equal(inf.dec[:sil.numel()], sil))
    assert ok or args.no_verify, "config 5: gunzip output != input"
    ms = dt / 3 * 1e3
    k, kms = _dominant(prof, 3)
    res["config5_gunzip_128x1MiB_members"] = {"MBps": round(sil.numel() / ms / 1e3, 1), "ms": round(ms, 3), "kernel": k,
                                              "kernel_ms": round(kms, 3), "roofline_frac": _frac(sil.numel() + n_comp, kms),
                                              "pipeline_frac": _frac(sil.numel() + n_comp, ms),
                                              "kernels_ms": {kk: round(v[0] / 3, 3) for kk, v in sorted(prof.items())},
                                              "output_equals_input": ok}
    del mk, inf, comp, sil
    # level 9 on the inputs whose chains are long: binary records, sparse zeros (64 MiB each, 65535-byte chunks)
    n = 64 << 20
    z = np.zeros(n, dtype=np.uint8)
    k_ = n // 97 + 1
    where = (synth.splitmix64(4242, k_) % np.uint64(n)).astype(np.int64)
    z[where] = (synth.splitmix64(4243, k_) & np.uint64(0xFF)).astype(np.uint8)
    for name, arr in (("level9_records_64MiB", synth._records(4242, n)), ("level9_sparse_zeros_64MiB", z)):
        d = torch.from_numpy(arr).to(device)
        compress_case(name, d, CHUNK, 0, 9, steps=1, sample=2)
        del d
    return res


def one_stream_inflate(args, torch, eng, device):
    """Inflate of ONE long stream, device-resident (the reference's own decompress benchmark is one 177 MB file,
    readme.md:47 / bin/inflate_bench.zig:14): a gzip level-6 stream of 177,244,160 bytes of text and the
    huffman-only stream of BASELINE.json configs[3] (128 MiB of the Silesia-like mix), both made by this engine."""
    from flate_amd import synth
    res = {}
    cases = [("gunzip_one_177MB_level6_text_stream", lambda: synth.text_torch(synth.SEED_TEXT + 7, 177_244_160, device=device), 6),
             ("gunzip_one_128MiB_huffman_only_stream",
              lambda: torch.from_numpy(synth.silesia_like(synth.SEED_SILESIA, 128 << 20)).to(device), 1)]
    for name, mk, mode in cases:
        d = mk()
        n = d.numel()
        job = CompressJob(torch, eng, d, n, 1, mode)
        job.step()
        torch.cuda.synchronize()
        lens = job.results()
        comp, comp_off, n_comp = job.packed(lens)
        inf = InflateJob(torch, eng, comp, comp_off, 1, job.in_off, n, 1)
        dt, prof = timed(torch, None, eng, inf.step, 3, 1, 1)
        ok = int(inf.dec_st.abs().sum().item()) == 0 and bool(torch.equal(inf.dec[:n], d))
        assert ok or args.no_verify, "%s: inflate(deflate(x)) != x" % name
        res[name] = {"MBps": round(n * 3 / dt / 1e6, 1), "ms": round(dt / 3 * 1e3, 2), "compressed_bytes": int(n_comp),
                     "pipeline_frac": _frac(n + int(n_comp), dt / 3 * 1e3),
                     "kernels_ms": {k: round(v[0] / 3, 2) for k, v in sorted(prof.items())}, "output_equals_input": ok}
        del job, inf, d, comp
    return res


if __name__ == "__main__":
    main()
