"""GPU parity tests in the shapes BASELINE.json's configurations have on one rank: the
reduced shapes live in test_gpu_compress.py / test_gpu_inflate.py, these are the real ones at sizes the oracle
still finishes in seconds.

  configs[3]  huffman-only gzip of ONE long Silesia-like stream, through flate_hip_compress_batch_sharded
              (the C ABI the multi-GPU run uses, here with an RCCL communicator of one rank)
  configs[4]  gunzip of a batch of 1 MiB gzip level-6 members made by the oracle from Silesia-like slices
              (few long streams: k_inflate_par), outputs + consumed counts + the container's CRC-32 / ISIZE
  and one stream of several MiB through the same inflater.
"""
import ctypes as C
import os
import zlib as pyzlib

import numpy as np
import pytest

import _oracle as O
from gpu_util import engine

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _one_rank_comm():
    import torch
    import torch.distributed  # noqa: F401  (loads the RCCL that torch ships)
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return rccl, comm


def test_config4_one_long_huffman_only_gzip_stream_through_the_sharded_entry_point():
    import torch
    from flate_amd import _capi, synth
    eng = engine()
    data = synth.silesia_like(synth.SEED_SILESIA + 4, 8 * 1024 * 1024 + 12345).tobytes()
    want = O.compress(data, O.GZIP, O.HUFFMAN)
    rccl, comm = _one_rank_comm()
    try:
        dev = torch.device("cuda", 0)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        in_off = torch.tensor([0, len(data)], dtype=torch.int64, device=dev)
        cap = (eng.compress_bound(len(data), O.GZIP, O.HUFFMAN) + 15) & ~15
        out_off = torch.tensor([0, cap], dtype=torch.int64, device=dev)
        out = torch.zeros(cap + 8, dtype=torch.uint8, device=dev)
        out_len = torch.zeros(1, dtype=torch.int64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        gathered = torch.zeros(cap, dtype=torch.uint8, device=dev)
        sizes = torch.zeros(1, dtype=torch.int64, device=dev)
        dst_off = torch.zeros(2, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L = _capi.lib()
        rc = L.flate_hip_compress_batch_sharded(eng._h, comm, 0, 1, d_in.data_ptr(), in_off.data_ptr(), 1, O.GZIP, O.HUFFMAN,
                                                out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(),
                                                gathered.data_ptr(), cap, sizes.data_ptr(), dst_off.data_ptr())
        assert rc == 0, eng._L.flate_hip_last_error(eng._h)
        torch.cuda.synchronize()
        assert int(status[0]) == 0 and int(sizes[0]) == len(want) and int(out_len[0]) == len(want)
        assert gathered[:len(want)].cpu().numpy().tobytes() == want
        assert pyzlib.decompress(want, 31) == data  # (and a third implementation reads it)
    finally:
        rccl.ncclCommDestroy(comm)


def test_config5_gunzip_of_oracle_made_1mib_level6_members():
    from flate_amd import synth
    eng = engine()
    n, size = 16, 1 << 20
    data = synth.silesia_like(synth.SEED_SILESIA + 5, n * size).tobytes()
    parts = [data[i * size:(i + 1) * size] for i in range(n)]
    members = [O.compress(p, O.GZIP, 6) for p in parts]
    outs, st, used = eng.decompress_many(members, O.GZIP, caps=[size] * n)  # (slots of exactly ISIZE bytes)
    assert st == [0] * n
    assert outs == parts
    assert used == [len(m) for m in members]
    # the footer is checked: a member whose CRC-32 or ISIZE is off by one bit is refused with the reference's names
    bad_crc = bytearray(members[3])
    bad_crc[-5] ^= 1
    bad_size = bytearray(members[4])
    bad_size[-1] ^= 0x40
    outs, st, used = eng.decompress_many([bytes(bad_crc), bytes(bad_size), members[5]], O.GZIP, caps=[size + 8] * 3)
    assert [O.STATUS[s] for s in st] == ["WrongGzipChecksum", "WrongGzipSize", "Ok"]
    assert outs[2] == parts[5]
    # two members back to back in one input: `consumed` stops behind the first (inflate.zig:301-309)
    outs, st, used = eng.decompress_many([members[0] + members[1]], O.GZIP, caps=[size + 8])
    assert st == [0] and outs[0] == parts[0] and used == [len(members[0])]


@pytest.mark.parametrize("mode", [6, O.HUFFMAN, 0])
def test_one_stream_of_several_mib(mode):
    from flate_amd import synth
    eng = engine()
    data = synth.silesia_like(synth.SEED_SILESIA + 6, 5 * 1024 * 1024 + 777).tobytes()
    comp = O.compress(data, O.ZLIB, mode)
    outs, st, used = eng.decompress_many([comp], O.ZLIB, caps=[len(data)])
    assert st == [0] and used == [len(comp)]
    assert outs[0] == data
    # truncated in the middle / one bit flipped in the last quarter: the oracle's verdict
    cut = comp[:len(comp) // 2]
    flip = bytearray(comp)
    flip[(len(comp) * 7) // 8] ^= 0x04
    for bad in (cut, bytes(flip)):
        outs, st, used = eng.decompress_many([bad], O.ZLIB, caps=[len(data) + 8])
        name, want, wused = O.decompress(bad, O.ZLIB, 0, cap=len(data) + 8)
        assert O.STATUS[st[0]] == name


def test_output_gather_overlapped_three_steps_one_rank():
    # The exchange of step k on its own stream beside the kernels of step k + 1 (sharded.OutputGather, two pack
    # buffers, events both ways): three steps with different inputs, every step's reassembled shard compared with
    # that step's packed streams.  One rank (RCCL through torch.distributed); the 8-GPU run is the driver's.
    import torch
    import torch.distributed as dist
    from flate_amd import sharded, synth
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    import sys
    # its own process: a process group is process-wide state
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
import _oracle as O
from flate_amd import Engine, sharded, synth
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", FLATE_GATHER_OVERLAP="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng = Engine(0); eng.set_stream(stream.cuda_stream); eng.set_sync(False)
n_chunks, chunk = 300, 40000
def batch(seed):
    d = torch.from_numpy(synth.text(seed, n_chunks * chunk)).to(dev)
    off = np.arange(n_chunks + 1, dtype=np.int64) * chunk
    caps = np.array([(eng.compress_bound(chunk, 0, 6) + 7) & ~7] * n_chunks, dtype=np.int64)
    oo = np.zeros(n_chunks + 1, dtype=np.int64); np.cumsum(caps, out=oo[1:])
    return d, torch.from_numpy(off).to(dev), torch.from_numpy(oo).to(dev), oo
datas = [batch(s) for s in (11, 12, 13)]
out = [torch.empty(int(datas[0][3][-1]) + 8, dtype=torch.uint8, device=dev) for _ in range(3)]
lens = [torch.zeros(n_chunks, dtype=torch.int64, device=dev) for _ in range(3)]
st = [torch.zeros(n_chunks, dtype=torch.int32, device=dev) for _ in range(3)]
def compress(i):
    d, io, oo, _ = datas[i]
    eng.compress_device(d.data_ptr(), io.data_ptr(), n_chunks, 0, 6, out[i].data_ptr(), oo.data_ptr(), lens[i].data_ptr(), st[i].data_ptr())
compress(0)
g = sharded.OutputGather(1, 0, dev, int(datas[0][3][-1]), engine=eng)
g.calibrate(out[0], datas[0][2], lens[0])
assert g.form.endswith("-overlapped"), g.form
for i in range(3):
    compress(i)
    g.run(out[i], datas[i][2], lens[i])   # returns at once; the exchange of step i runs beside step i + 1
    if i < 2:
        compress(i + 1)                   # (the next step's kernels, enqueued while the exchange is in flight)
    sizes = g.sizes_host()
    got = g.shard(0, sizes).cpu().numpy().tobytes()
    l = lens[i].cpu().numpy(); oo = datas[i][3]
    o = out[i].cpu().numpy()
    want = b"".join(o[int(oo[k]):int(oo[k]) + int(l[k])].tobytes() for k in range(n_chunks))
    assert int(st[i].abs().sum().item()) == 0 and got == want, "step %%d: reassembled shard differs" %% i
    data = datas[i][0].cpu().numpy().tobytes()
    assert want[:int(l[0])] == O.compress(data[:chunk], O.RAW, 6)
assert not g.overflowed()
dist.destroy_process_group()
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_config2_at_its_real_size_sampled_parity_and_round_trip():
    # BASELINE.json configs[1] at full size: raw deflate level 6 of 1 GiB of the benchmark text in 16385 chunks of 65535
    # bytes -- 256 chunks spread over the batch byte for byte against the oracle, every stream through the GPU inflater
    # back to the input (the size-independent property), lengths and statuses of all of them.
    import torch
    from flate_amd import synth
    eng = engine()
    dev = torch.device("cuda", 0)
    n, chunk = 1 << 30, 65535
    data = synth.text_torch(synth.SEED_TEXT, n, device=dev)
    off = synth.split_offsets(n, chunk)
    k = len(off) - 1
    assert k == 16385
    caps = np.array([(eng.compress_bound(int(off[i + 1] - off[i]), 0, 6) + 7) & ~7 for i in range(k)], dtype=np.uint64)
    oo = np.zeros(k + 1, dtype=np.uint64)
    np.cumsum(caps, out=oo[1:])
    io = torch.from_numpy(off.astype(np.int64)).to(dev)
    ot = torch.from_numpy(oo.astype(np.int64)).to(dev)
    out = torch.empty(int(oo[-1]) + 8, dtype=torch.uint8, device=dev)
    ol = torch.zeros(k, dtype=torch.int64, device=dev)
    st = torch.zeros(k, dtype=torch.int32, device=dev)
    eng.compress_device(data.data_ptr(), io.data_ptr(), k, 0, 6, out.data_ptr(), ot.data_ptr(), ol.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0
    lens = ol.cpu().numpy()
    for i in list(range(0, k, 65)) + [k - 2, k - 1]:
        a, b = int(off[i]), int(off[i + 1])
        got = out[int(oo[i]):int(oo[i]) + int(lens[i])].cpu().numpy().tobytes()
        assert got == O.compress(data[a:b].cpu().numpy().tobytes(), O.RAW, 6), i
    # every stream back through the inflater (the streams packed back to back first): slots = the input's own layout
    from flate_amd import sharded
    comp_off_np = np.zeros(k + 1, dtype=np.int64)
    np.cumsum(lens, out=comp_off_np[1:])
    comp = torch.empty(int(comp_off_np[-1]) + 8, dtype=torch.uint8, device=dev)
    comp_off = torch.from_numpy(comp_off_np).to(dev)
    sharded.compact(out, torch.from_numpy(oo[:-1].astype(np.int64)).to(dev), ol, comp, comp_off, engine=eng)
    out, ot = comp, comp_off
    dec = torch.empty(n + 8, dtype=torch.uint8, device=dev)
    dl = torch.zeros(k, dtype=torch.int64, device=dev)
    ds = torch.zeros(k, dtype=torch.int32, device=dev)
    eng.decompress_device(out.data_ptr(), ot.data_ptr(), k, 0, 0, dec.data_ptr(), io.data_ptr(), dl.data_ptr(), ds.data_ptr())
    torch.cuda.synchronize()
    assert int(ds.abs().sum().item()) == 0
    assert bool(torch.equal(dec[:n], data))
