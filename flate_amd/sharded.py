"""Multi-GPU sharding of a batch: one process per GPU, chunks are
independent, so each rank runs the single-GPU path on its own contiguous range of
chunks with no data-path collective; the only exchange is the reassembly of the
output: ranks all-gather their packed sizes (tiny) and then every rank's packed
shard is broadcast to its peers over RCCL (xGMI is point to point: one shard per
link, no ring).

torch.distributed is plumbing here (backend "nccl" = RCCL on ROCm; "gloo" in the
CPU tests of the protocol).
"""
import numpy as np


def shard_ranges(sizes, world):
    """Contiguous chunk ranges balanced by input bytes: list of (lo, hi) per rank."""
    sizes = np.asarray(sizes, dtype=np.int64)
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

    Every rank packs its streams back to back, the packed sizes are all-gathered (one int64
    per rank), and the shards are exchanged with ONE all-gather of equal-sized slices (each
    padded to the largest shard): on xGMI every GPU has a direct link to every peer, so all
    shards move concurrently.  `gathered[r, :sizes[r]]` is rank r's packed output afterwards.
    """

    def __init__(self, world, rank, device, local_cap, engine=None):
        import torch
        from .engine import default_engine
        self.world, self.rank, self.device = world, rank, device
        self.engine = engine if engine is not None else (default_engine() if device.type == "cuda" else None)
        self.local_cap = (int(local_cap) + 15) & ~15
        self.packed = torch.empty(self.local_cap + 16, dtype=torch.uint8, device=device)
        self.sizes = torch.zeros(world, dtype=torch.int64, device=device)
        # one flat buffer: a step uses its first world * width bytes as `world` equal slices, which
        # is what all_gather_into_tensor wants (no per-rank tensor list, no staging copy)
        self.gathered = torch.empty(world * self.local_cap, dtype=torch.uint8, device=device)
        self.width = 0
        self.dst_off = None

    def run(self, out, out_off, out_len):
        """out_off: n+1 slot starts; out_len: n produced lengths.  Returns the per-rank packed sizes."""
        import torch
        import torch.distributed as dist
        n = out_len.numel()
        if self.dst_off is None or self.dst_off.numel() != n + 1:
            self.dst_off = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        pack_streams(self.engine, out, out_off, out_len, self.packed, self.dst_off)
        dist.all_gather_into_tensor(self.sizes, self.dst_off[n:n + 1])
        sizes = [int(x) for x in self.sizes.cpu().tolist()]
        self.width = (max(sizes) + 15) & ~15  # same slice width on every rank
        dist.all_gather_into_tensor(self.gathered[: self.world * self.width], self.packed[: self.width])
        return sizes

    def shard(self, r, sizes):
        return self.gathered[r * self.width: r * self.width + sizes[r]]
