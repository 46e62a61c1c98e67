"""World-size-2 test of the multi-GPU protocol on CPU with the gloo
backend: contiguous byte-balanced shards, no data-path collective, reassembly of the
compressed shards on every rank.  The per-rank "compress" here is the CPU oracle (test
infrastructure) -- the protocol under test is flate_amd/sharded.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
from flate_amd import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.text(synth.SEED_TEXT, 9 * 65535 + 321).tobytes()
        chunks = [data[i:i + 65535] for i in range(0, len(data), 65535)]
        ranges = sharded.shard_ranges([len(c) for c in chunks], world)
        lo, hi = ranges[rank]
        mine = chunks[lo:hi]
        # slot layout exactly like the C ABI: n+1 slot starts, produced lengths
        caps = [(O.lib().fo_compress_bound(len(c)) + 7) & ~7 for c in mine]
        out_off = torch.zeros(len(mine) + 1, dtype=torch.int64)
        out_off[1:] = torch.cumsum(torch.tensor(caps, dtype=torch.int64), 0)
        out = torch.zeros(int(out_off[-1]) + 8, dtype=torch.uint8)
        out_len = torch.zeros(len(mine), dtype=torch.int64)
        for i, c in enumerate(mine):
            z = O.compress(c, O.GZIP, 6)
            out[int(out_off[i]):int(out_off[i]) + len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8)
            out_len[i] = len(z)
        want = b"".join(O.compress(c, O.GZIP, 6) for c in chunks)
        ok = True
        for algo in ("all_gather", "p2p"):
            # exact mode: sizes read on the host every step
            g = sharded.OutputGather(world, rank, torch.device("cpu"), int(out_off[-1]), algo=algo)
            sizes = g.run(out, out_off, out_len)
            whole = b"".join(g.shard(r, sizes).numpy().tobytes() for r in range(world))
            ok = ok and whole == want
            # calibrated mode: one host sync up front, then none; the sizes stay in a tensor
            g2 = sharded.OutputGather(world, rank, torch.device("cpu"), int(out_off[-1]), algo=algo)
            g2.calibrate(out, out_off, out_len)
            assert g2.run(out, out_off, out_len) is None and not g2.overflowed()
            s2 = g2.sizes_host()
            ok = ok and s2 == sizes and b"".join(g2.shard(r, s2).numpy().tobytes() for r in range(world)) == want
        # ranks with different slot capacities agree on one buffer size
        g3 = sharded.OutputGather(world, rank, torch.device("cpu"), int(out_off[-1]) + 1000 * rank)
        ok = ok and g3.local_cap >= int(out_off[-1]) + 1000 * rank
        q.put((rank, ok, sizes, ranges, g3.local_cap))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_are_contiguous_and_balanced():
    sizes = [65535] * 100 + [10]
    for world in (1, 2, 3, 8):
        r = sharded.shard_ranges(sizes, world)
        assert r[0][0] == 0 and r[-1][1] == len(sizes)
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        per = [sum(sizes[a:b]) for a, b in r]
        assert max(per) - min(per) <= 2 * 65535
    assert sharded.shard_ranges([5], 4) == [(0, 0), (0, 0), (0, 0), (0, 1)] or \
        sum(b - a for a, b in sharded.shard_ranges([5], 4)) == 1


def test_two_rank_gather_reassembles_every_shard_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sizes, ranges, cap in res:
        assert ok, rank
        assert len(sizes) == world and all(s > 0 for s in sizes)
    assert len({r[4] for r in res}) == 1  # one buffer size on every rank


def _inflate(member):
    r = O.decompress(member, O.GZIP)
    assert r[0] == "Ok"
    return r[1]


def test_inflate_shards_by_isize():
    """Config #5's partitioning: gzip members go to ranks by ISIZE (the trailer's
    uncompressed size), contiguous and balanced; together the ranges cover every member once."""
    rng = np.random.default_rng(5)
    data = synth.text(synth.SEED_TEXT, 1 << 20).tobytes()
    cuts = np.sort(rng.choice(np.arange(1, len(data)), 40, replace=False))
    pieces = [data[a:b] for a, b in zip(np.r_[0, cuts], np.r_[cuts, len(data)])]
    members = [O.compress(p, O.GZIP, 6) for p in pieces]
    isize = [int.from_bytes(m[-4:], "little") for m in members]
    assert isize == [len(p) for p in pieces]
    for world in (2, 3, 8):
        ranges = sharded.shard_ranges(isize, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(members)
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        per = [sum(isize[a:b]) for a, b in ranges]
        assert max(per) - min(per) <= 2 * max(isize)
        back = b"".join(_inflate(m) for a, b in ranges for m in members[a:b])
        assert back == data


def test_bench_spawns_the_ranks_itself():
    """`bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N
    ranks (checked without a GPU: the command line it would exec)."""
    import importlib.util
    import sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    old_execv, old_argv, old_ws = os.execv, sys.argv, os.environ.pop("WORLD_SIZE", None)
    try:
        os.execv = lambda exe, cmd: seen.setdefault("cmd", cmd)
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
        args = bench.parse()
        bench.maybe_spawn(args)
    finally:
        os.execv, sys.argv = old_execv, old_argv
        if old_ws is not None:
            os.environ["WORLD_SIZE"] = old_ws
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # under a launcher nothing is spawned
    os.environ["WORLD_SIZE"] = "4"
    try:
        seen.clear()
        os.execv = lambda exe, cmd: seen.setdefault("cmd", cmd)
        bench.maybe_spawn(args)
        assert not seen
    finally:
        os.execv = old_execv
        if old_ws is None:
            os.environ.pop("WORLD_SIZE", None)
        else:
            os.environ["WORLD_SIZE"] = old_ws


def _nsw_worker(rank, world, port, q):
    """bench.py's N > 1 workloads with two CPU ranks: the compress job, the inflate job and the timer are stand-ins (the oracle
    on a few chunks; barrier + MAX over the ranks on CPU tensors) -- what is under test is that every rank takes part, that
    the values are whole-job rates and that the keys the north star names are in the dict."""
    import argparse
    import importlib.util
    import time
    from conftest import ROOT
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)

        class Job:
            def __init__(self, d):
                self.d = d.numpy()[: 3 * 65535]
                self.lens = None

            def step(self):
                b = self.d.tobytes()
                self.lens = np.array([len(O.compress(b[i:i + 65535], O.RAW, 6)) for i in range(0, len(b), 65535)])

            def results(self):
                return self.lens

        class Inf:
            def step(self):
                pass

        def timer(step, steps, warmup):
            for _ in range(warmup):
                step()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0 + 0.001 * rank], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()), {"k_lz_parse": (1.0, steps)}

        # (small stand-ins for the 256 / 128 MiB buffers: the sizes in the keys are the bench's, the tensors here are not)
        real_zeros, real_sil = torch.zeros, synth.silesia_like
        torch.zeros = lambda n, **kw: real_zeros(min(n, 4 * 65535), dtype=torch.uint8)
        synth.silesia_like = lambda seed, n: real_sil(seed, min(n, 4 * 65535))
        try:
            args = argparse.Namespace(no_decompress=False, no_verify=True)
            res = bench.north_star_workloads(args, torch, dist, None, world, rank, torch.device("cpu"), make_job=Job, timer=timer,
                                             make_inflate=lambda job, lens, d: Inf())
        finally:
            torch.zeros, synth.silesia_like = real_zeros, real_sil
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_bench_line_carries_the_north_stars_workloads_at_world_2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nsw_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        res = got[r]
        assert set(res) == {"zeros_256MiB_64KiB_chunks", "silesia_like_128MiB_64KiB_chunks"}
        for v in res.values():
            assert v["n_gpus"] == world and v["MBps"] > 0 and v["decompress_MBps"] > 0 and "roofline_frac" in v and "ratio_rank0" in v
    # whole-job rates from the slowest rank's time: every rank reports the same value
    assert got[0]["zeros_256MiB_64KiB_chunks"]["MBps"] == got[1]["zeros_256MiB_64KiB_chunks"]["MBps"]
