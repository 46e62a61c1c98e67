"""CPU check of the serial block planner (flate_amd/csrc/flate_common.h) -- the exact
source that one GPU lane per block executes -- against the oracle and the reference's
golden block vectors (block_writer.zig:599-706).  No GPU."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN, ROOT, golden

SHIM_DIR = os.path.join(ROOT, "tests", "cpu_shim")
SHIM_SO = os.path.join(SHIM_DIR, "libplanner_shim.so")
NO_INPUT = 0xFFFFFFFF


class Plan(C.Structure):
    _fields_ = [("type", C.c_uint32), ("size_bits", C.c_uint32), ("hdr_nbits", C.c_uint32),
                ("final_block", C.c_uint32), ("in_start", C.c_uint32), ("in_len", C.c_uint32),
                ("tok_start", C.c_uint32), ("tok_count", C.c_uint32), ("valid", C.c_uint32),
                ("no_input", C.c_uint32), ("q1_gap", C.c_uint32), ("pad_", C.c_uint32), ("bit_off", C.c_uint64), ("hdr", C.c_uint8 * 640),
                ("lit", C.c_uint16 * (2 * 286)), ("dist", C.c_uint16 * (2 * 30))]


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(SHIM_DIR, "planner_shim.cpp")
    deps = [src] + [os.path.join(ROOT, "flate_amd", "csrc", h) for h in ("flate_common.h", "flate_layout.h", "stream_tables.h")]
    if not os.path.exists(SHIM_SO) or os.path.getmtime(SHIM_SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-Wall", "-fsanitize=undefined", "-fno-sanitize-recover",
                        "-fPIC", "-shared", "-o", SHIM_SO, src], check=True)
    lib = C.CDLL(SHIM_SO)
    lib.shim_plan_block.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.shim_huff_generate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.shim_tables.argtypes = [C.c_void_p] * 6
    lib.shim_set_pm.argtypes = [C.c_int]
    assert lib.shim_plan_sizeof() == C.sizeof(Plan)
    return lib


class BitSink:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nb):
        self.acc |= int(v) << self.n
        self.n += int(nb)
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def align(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
        self.acc, self.n = 0, 0


def tables(lib):
    li = np.zeros(256, np.uint8); le = np.zeros(29, np.uint8); lb = np.zeros(29, np.uint8)
    dc = np.zeros(32768, np.uint8); de = np.zeros(30, np.uint8); db = np.zeros(30, np.uint16)
    lib.shim_tables(li.ctypes.data, le.ctypes.data, lb.ctypes.data, dc.ctypes.data, de.ctypes.data, db.ctypes.data)
    return li, le, lb, dc, de, db


def encode_block(lib, mode, tokens, input_bytes, eof, dyn=False):
    """Assemble the block bytes the way the encode kernel does: planner output + codes."""
    li, le, lb, dc, de, db = tables(lib)
    lit = np.zeros(286, np.uint16)
    dist = np.zeros(30, np.uint16)
    if mode == 0:
        for t in tokens:
            t = int(t)
            if (t >> 23) & 1:
                lit[257 + int(li[(t >> 15) & 0xFF])] += 1
                dist[dc[t & 0x7FFF]] += 1
            else:
                lit[(t >> 15) & 0xFF] += 1
    else:
        h = np.bincount(np.frombuffer(input_bytes, np.uint8), minlength=256)
        lit[:256] = h
    in_len = NO_INPUT if input_bytes is None else len(input_bytes)
    plan = Plan()
    lib.shim_plan_block(2 if dyn else mode, lit.ctypes.data, dist.ctypes.data, in_len, int(eof), C.addressof(plan))
    s = BitSink()
    if plan.type == 0:  # stored
        s.put(1 if eof else 0, 3)
        s.align()
        s.put(len(input_bytes), 16)
        s.put((~len(input_bytes)) & 0xFFFF, 16)
        s.out += input_bytes
        return bytes(s.out), plan
    hdr = bytes(plan.hdr)
    for i in range(plan.hdr_nbits // 8):
        s.put(hdr[i], 8)
    if plan.hdr_nbits % 8:
        s.put(hdr[plan.hdr_nbits // 8] & ((1 << (plan.hdr_nbits % 8)) - 1), plan.hdr_nbits % 8)
    lc = np.array(plan.lit, np.uint16).reshape(286, 2)
    dcodes = np.array(plan.dist, np.uint16).reshape(30, 2)
    if mode == 0:
        for t in tokens:
            t = int(t)
            if (t >> 23) & 1:
                ll = (t >> 15) & 0xFF
                idx = int(li[ll])
                s.put(lc[257 + idx][0], lc[257 + idx][1])
                if le[idx]:
                    s.put(ll - lb[idx], le[idx])
                d = t & 0x7FFF
                c = int(dc[d])
                s.put(dcodes[c][0], dcodes[c][1])
                if de[c]:
                    s.put(d - db[c], de[c])
            else:
                b = (t >> 15) & 0xFF
                s.put(lc[b][0], lc[b][1])
    else:
        for b in input_bytes:
            s.put(lc[b][0], lc[b][1])
    s.put(lc[256][0], lc[256][1])
    nbits = len(s.out) * 8 + s.n
    assert nbits == plan.size_bits, (nbits, plan.size_bits)
    s.align()
    return bytes(s.out), plan


def test_tables_match_oracle(shim):
    li, le, lb, dc, de, db = tables(shim)
    for v in range(256):
        assert 257 + int(li[v]) == O.lib().fo_length_code(v)
    for d in range(32768):
        assert dc[d] == O.lib().fo_distance_code(d)
    for i in range(29):
        assert le[i] == O.lib().fo_length_extra_bits(257 + i)
    for i in range(30):
        assert de[i] == O.lib().fo_distance_extra_bits(i)
    assert list(lb) == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112,
                        128, 160, 192, 224, 255]
    assert list(db[:6]) == [0, 1, 2, 3, 4, 6] and db[29] == 0x6000 and db[28] == 0x4000


@pytest.fixture(params=[0, 1], ids=["lazy-loop", "package-merge"])
def pm(request, shim):
    """Both forms of the Huffman bit counts: the reference's lazy loop and the package-merge form the GPU runs."""
    shim.shim_set_pm(request.param)
    yield request.param
    shim.shim_set_pm(0)


def test_package_merge_equals_the_lazy_loop(shim):
    """huffman_encoder.zig:122-247 as package-merge (flate_common.h fl_huff_bit_counts_pm) against the loop
    itself, on frequency sets that stress ties, the length limit and totals of exactly 65536 (Q3's 65535-leaves)."""
    rng = np.random.default_rng(23)
    n_cases = 0
    for it in range(1600):
        kind = it % 8
        n = int(rng.integers(3, 287))
        f = np.zeros(286, np.int64)
        if kind == 0:
            f = rng.integers(0, 50, 286)
        elif kind == 1:
            f = (rng.pareto(1.0, 286) * 20).clip(0, 60000).astype(np.int64)
        elif kind == 2:
            a, b = 1, 1
            for i in range(min(n, 23)):
                f[i] = a
                a, b = b, a + b
        elif kind == 3:
            w = rng.dirichlet(np.ones(n) * 0.3)
            f[:n] = np.floor(w * 65536).astype(np.int64)
            f[0] += 65536 - f.sum()
        elif kind == 4:
            f[:n] = rng.integers(1, 4, n)
        elif kind == 5:
            f[:n] = (2.0 ** rng.integers(0, 12, n)).astype(np.int64)
        elif kind == 6:
            f[:n] = 1
            f[rng.integers(0, n)] = 65535 - n
        else:
            f = rng.integers(0, 2000, 286) * (rng.random(286) < 0.4)
        f = np.clip(f, 0, 65535)
        while f.sum() > 65536:
            f = (f + 1) // 2 * (f > 0)
        rng.shuffle(f)
        for sub, mb in ((f, 15), (f[:30], 15), (f[:19], 7)):
            sub = np.ascontiguousarray(sub, dtype=np.uint16)
            res = []
            for form in (0, 1):
                shim.shim_set_pm(form)
                cs = np.zeros(sub.size, np.uint16)
                ls = np.zeros(sub.size, np.uint16)
                shim.shim_huff_generate(sub.ctypes.data, sub.size, mb, cs.ctypes.data, ls.ctypes.data)
                res.append((cs, ls))
            shim.shim_set_pm(0)
            assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
            n_cases += 1
    assert n_cases == 4800


def test_huffman_generate_matches_oracle(shim, pm):
    rng = np.random.default_rng(7)
    cases = []
    for n, mb in ((286, 15), (30, 15), (19, 7)):
        for trial in range(150):
            kind = trial % 6
            if kind == 0:
                f = rng.integers(0, 50, n)
            elif kind == 1:
                f = (rng.pareto(0.6, n) * 3).astype(np.int64)
            elif kind == 2:  # fibonacci-like: forces the length limit
                f = np.zeros(n, np.int64)
                a, b = 1, 1
                for i in range(min(n, 22)):
                    f[rng.integers(0, n)] = a
                    a, b = b, a + b
            elif kind == 3:
                f = rng.integers(0, 3, n)
            elif kind == 4:
                f = np.zeros(n, np.int64)
                k = int(rng.integers(0, 5))
                f[rng.choice(n, k, replace=False)] = rng.integers(1, 1000, k)
            else:
                f = rng.integers(0, 65535, n) * (rng.random(n) < 0.3)
            f = np.clip(f, 0, 65535)
            # keep the total within what a block can hold (<= 65536 symbols)
            while f.sum() > 65536:
                f = f // 2
            cases.append((f.astype(np.uint16), mb))
    for f, mb in cases:
        co, lo = O.huffman_generate(f, mb)
        cs = np.zeros(f.size, np.uint16)
        ls = np.zeros(f.size, np.uint16)
        shim.shim_huff_generate(f.ctypes.data, f.size, mb, cs.ctypes.data, ls.ctypes.data)
        assert (lo == ls).all() and (co == cs).all()


def _cases():
    with open(os.path.join(GOLDEN, "block_writer_tokens.json")) as f:
        cases = json.load(f)
    for c in cases:
        c["tok"] = np.array([O.tok_lit(t[0]) if len(t) == 1 else O.tok_match(t[0], t[1])
                             for t in c["tokens"]], dtype=np.uint32)
    return cases


@pytest.mark.parametrize("fn", ["wb", "dyn"])
def test_planner_reproduces_block_writer_goldens(shim, pm, fn):
    n = 0
    for c in _cases():
        for with_input in (True, False):
            if with_input and not (c["input"] and c["want"]):
                continue
            inp = golden("block_writer", c["input"]) if with_input else None
            name = (c["want"] if with_input else c["want_no_input"]).replace("{s}", fn)
            want = golden("block_writer", name)
            got, plan = encode_block(shim, 0, c["tok"], inp, False, fn == "dyn")
            assert got == want, name
            got, plan = encode_block(shim, 0, c["tok"], inp, True, fn == "dyn")
            assert got[0] & 1 and bytes([got[0] & 0xFE]) + got[1:] == want
            n += 1
    assert n == 17


def test_planner_reproduces_huffman_block_goldens(shim, pm):
    names = [c["input"] for c in _cases() if c["input"]] + ["huffman-rand-max.input"]
    for name in names:
        inp = golden("block_writer", name)
        want = golden("block_writer", name.replace(".input", ".huff.expect"))
        got, plan = encode_block(shim, 1, None, inp, False)
        assert got == want, name


def test_planner_matches_oracle_on_random_blocks(shim, pm):
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(0, 3000))
        kind = trial % 4
        if kind == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = rng.integers(97, 101, n, dtype=np.uint8).tobytes()
        elif kind == 2:
            data = (b"the quick brown fox jumps over the lazy dog " * (n // 40 + 1))[:n]
        else:
            data = bytes(n)
        toks = O.tokenize(data, 6)
        for eof in (False, True):
            want = O.block_write("wb", toks, eof, data)
            got, _ = encode_block(shim, 0, toks, data, eof)
            assert got == want
            want = O.block_write("huff", toks, eof, data)
            got, _ = encode_block(shim, 1, None, data, eof)
            assert got == want
