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
        g = sharded.OutputGather(world, rank, torch.device("cpu"), int(out_off[-1]))
        sizes = g.run(out, out_off, out_len)
        whole = b"".join(g.shard(r, sizes).numpy().tobytes() for r in range(world))
        want = b"".join(O.compress(c, O.GZIP, 6) for c in chunks)
        q.put((rank, whole == want, sizes, ranges))
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
    for rank, ok, sizes, ranges in res:
        assert ok, rank
        assert len(sizes) == world and all(s > 0 for s in sizes)
