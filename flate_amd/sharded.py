"""Multi-GPU sharding of a batch: one process per GPU, chunks are
independent, so each rank runs the single-GPU path on its own contiguous range of
chunks with no data-path collective; the only exchange is the reassembly of the
output: ranks all-gather their packed sizes (tiny) and then every rank's packed
shard goes to each peer as one grouped send/recv batch over RCCL (xGMI is point to
point: one shard per link, no ring), see OutputGather.

torch.distributed is plumbing here (backend "nccl" = RCCL on ROCm; "gloo" in the
CPU tests of the protocol).
"""
import numpy as np


def shard_ranges(sizes, world):
    """Contiguous chunk ranges balanced by input bytes: list of (lo, hi) per rank."""
    sizes = np.asarray(sizes, dtype=np.int64)
    if world <= 0:
        raise ValueError("world must be positive")
    total = int(sizes.sum())
    cs = np.concatenate(([0], np.cumsum(sizes)))
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(cs, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), len(sizes)))
    bounds.append(len(sizes))
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def pack_streams(engine, out, out_off, out_len, dst, dst_off):
    """Pack variable-length streams back to back.  CUDA tensors go through the HIP kernel
    of libflate_hip.so; CPU tensors (only used by the gloo protocol tests) take a torch
    index path."""
    import torch
    n = out_len.numel()
    if out.is_cuda:
        engine.gather_streams_device(out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), n, dst.data_ptr(),
                                     dst_off.data_ptr())
        return
    lens = out_len.to(torch.int64)
    dst_off[0] = 0
    dst_off[1:] = torch.cumsum(lens, 0)
    for i in range(n):
        a, l, b = int(out_off[i]), int(lens[i]), int(dst_off[i])
        dst[b:b + l] = out[a:a + l]


def compact(out, out_slot_start, out_len, dst, dst_off, engine=None):
    """bench helper: pack using explicit slot starts (n entries)."""
    from .engine import default_engine
    import torch
    eng = engine or default_engine()
    tmp_off = torch.empty(out_len.numel() + 1, dtype=torch.int64, device=out.device)
    pack_streams(eng, out, out_slot_start, out_len, dst, tmp_off)
    dst_off.copy_(tmp_off)


class OutputGather:
    """Reassemble the compressed output of all ranks on every rank.

    Every rank packs its streams back to back (HIP kernel of libflate_hip.so), the packed sizes are
    all-gathered (one int64 per rank, they stay on the device) and every rank's shard travels to
    every peer as one slice of a common width.  Two exchange forms:
      "p2p"         one grouped batch of send/recv pairs (RCCL ncclGroupStart .. ncclSend / ncclRecv ..
                    ncclGroupEnd): every GPU has a direct xGMI link to every peer, so the 7 shards of a
                    rank move on 7 links at once -- the form SURVEY.md 8e prefers (default on nccl)
      "all_gather"  one all_gather_into_tensor of the padded slices (the library picks the algorithm)
    Width: `calibrate()` (once, outside a timed loop) fixes it from the first batch with some slack, after
    which `run()` never touches the host; a shard that outgrows it raises a device-side flag
    (`overflowed()`).  Without calibration `run()` reads the sizes on the host every time (exact width).
    `gathered[r * width : r * width + sizes[r]]` is rank r's packed output afterwards.

    Stream contract: the engine must run on torch's CURRENT stream (Engine.set_stream(
    torch.cuda.current_stream().cuda_stream) with a non-default stream current): the pack kernel and the next
    step's kernels are ordered by that one stream.

    Overlap (CUDA tensors, calibrated, FLATE_GATHER_OVERLAP != 0): the exchange of step k runs on a stream of its
    own, behind an event recorded after step k's pack kernel, and reads one of two pack buffers -- so it moves
    over xGMI while the kernels of step k + 1 compute; the pack of step k + 2 waits for it.  `sizes_host()`,
    `overflowed()` and `shard()` wait for the exchange first; a device-wide synchronize does too.
    """

    def __init__(self, world, rank, device, local_cap, engine=None, algo=None):
        import torch
        import torch.distributed as dist
        from .engine import default_engine
        self.world, self.rank, self.device = world, rank, device
        self.engine = engine if engine is not None else (default_engine() if device.type == "cuda" else None)
        # per-rank capacities differ when the shards are uneven: everybody allocates the largest
        cap = torch.tensor([int(local_cap)], dtype=torch.int64, device=device)
        if dist.is_initialized() and world > 1:
            dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        self.local_cap = (int(cap.item()) + 15) & ~15
        assert int(local_cap) <= self.local_cap
        self.packed = torch.empty(self.local_cap + 16, dtype=torch.uint8, device=device)
        self.sizes = torch.zeros(world, dtype=torch.int64, device=device)
        self.gathered = torch.empty(world * self.local_cap, dtype=torch.uint8, device=device)
        self.over = torch.zeros(1, dtype=torch.int64, device=device)
        self.width = 0
        self.calibrated = False
        self.dst_off = None
        if algo is None:
            algo = "p2p" if (dist.is_initialized() and dist.get_backend() == "nccl") else "all_gather"
        self.algo = algo
        import os
        self.overlap = device.type == "cuda" and os.environ.get("FLATE_GATHER_OVERLAP", "1") != "0"
        self.k = 0
        if self.overlap:
            self.comm_stream = torch.cuda.Stream(device=device)
            self.packed_bufs = [self.packed, torch.empty_like(self.packed)]
            self.dst_offs = [None, None]
            self.ev_pack = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_done = [torch.cuda.Event(), torch.cuda.Event()]

    def _pack(self, out, out_off, out_len):
        import torch
        n = out_len.numel()
        if self.dst_off is None or self.dst_off.numel() != n + 1:
            self.dst_off = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        pack_streams(self.engine, out, out_off, out_len, self.packed, self.dst_off)
        return n

    def _exchange(self, width, packed=None):
        import torch.distributed as dist
        g = self.gathered
        packed = self.packed if packed is None else packed
        if self.algo == "all_gather" or self.world == 1:
            dist.all_gather_into_tensor(g[: self.world * width], packed[:width])
            return
        ops = []
        for d in range(1, self.world):
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            ops.append(dist.P2POp(dist.isend, packed[:width], to))
            ops.append(dist.P2POp(dist.irecv, g[frm * width:(frm + 1) * width], frm))
        g[self.rank * width:(self.rank + 1) * width].copy_(packed[:width])
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def calibrate(self, out, out_off, out_len, slack=1.05):
        """Fix the slice width from this batch (one host sync): max packed size over the ranks * slack."""
        import torch.distributed as dist
        self._wait_exchange()  # (an exchange of an earlier run() may still be adding to `over` / writing `sizes`)
        n = self._pack(out, out_off, out_len)
        dist.all_gather_into_tensor(self.sizes, self.dst_off[n:n + 1])
        sizes = [int(x) for x in self.sizes.cpu().tolist()]
        assert max(sizes) <= self.local_cap, "a packed shard is larger than the gather buffer"
        self.width = min(self.local_cap, (int(max(sizes) * slack) + 4096 + 15) & ~15)
        self.calibrated = True
        self.over.zero_()
        return sizes

    def run(self, out, out_off, out_len):
        """out_off: n+1 slot starts; out_len: n produced lengths.  Calibrated: no host access, returns
        None (sizes stay in `self.sizes` on the device).  Otherwise returns the per-rank packed sizes."""
        import torch
        import torch.distributed as dist
        if self.calibrated and self.overlap:
            # (no fallback to the serial form from here: one rank that changed form alone would issue other
            # collectives than its peers and hang the job -- a failure of the exchange is raised to the caller)
            return self._run_overlapped(out, out_off, out_len)
        n = self._pack(out, out_off, out_len)
        dist.all_gather_into_tensor(self.sizes, self.dst_off[n:n + 1])
        if self.calibrated:
            self.over += (self.sizes.max() > self.width).to(torch.int64)
            self._exchange(self.width)
            return None
        sizes = [int(x) for x in self.sizes.cpu().tolist()]
        assert max(sizes) <= self.local_cap, "a packed shard is larger than the gather buffer"
        self.width = (max(sizes) + 15) & ~15  # same slice width on every rank
        self._exchange(self.width)
        return sizes

    def _run_overlapped(self, out, out_off, out_len):
        import torch
        import torch.distributed as dist
        if True:
            # step k's exchange on its own stream, behind the pack; the kernels of step k + 1 do not wait for it
            i = self.k & 1
            self.k += 1
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.ev_done[i])  # (the exchange that read this pack buffer last is over)
            n = out_len.numel()
            if self.dst_offs[i] is None or self.dst_offs[i].numel() != n + 1:
                self.dst_offs[i] = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            pack_streams(self.engine, out, out_off, out_len, self.packed_bufs[i], self.dst_offs[i])
            self.ev_pack[i].record(cur)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.ev_pack[i])
                dist.all_gather_into_tensor(self.sizes, self.dst_offs[i][n:n + 1])
                self.over += (self.sizes.max() > self.width).to(torch.int64)
                self._exchange(self.width, self.packed_bufs[i])
                self.ev_done[i].record(self.comm_stream)
            return None

    @property
    def form(self):
        """What run() does once calibrated: "p2p-overlapped" / "all_gather-overlapped" (exchange on its own stream beside
        the next step's kernels) or "p2p-serial" / "all_gather-serial"."""
        return "%s-%s" % (self.algo, "overlapped" if (self.calibrated and self.overlap) else "serial")

    def wait(self):
        """Make torch's current stream wait for the exchange in flight: `gathered`, `sizes` and `over` are what the last
        run() produced from here on (sizes_host(), overflowed(), shard() and calibrate() do this themselves)."""
        self._wait_exchange()

    def _wait_exchange(self):
        if getattr(self, "overlap", False):
            import torch
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def overflowed(self):
        self._wait_exchange()
        return bool(int(self.over.item()))

    def sizes_host(self):
        self._wait_exchange()
        return [int(x) for x in self.sizes.cpu().tolist()]

    def shard(self, r, sizes):
        self._wait_exchange()
        return self.gathered[r * self.width: r * self.width + sizes[r]]
