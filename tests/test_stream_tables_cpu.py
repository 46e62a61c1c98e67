"""CPU tests of the host logic of the whole-stream path (flate_amd/csrc/stream_tables.h): the
tile / piece / segment tables and, above all, the slide table, checked against a step-by-step
model of the reference's window bookkeeping (deflate.zig:154-211, 291-321, 335-347;
SlidingWindow.zig:36-60) run with arbitrary token steps."""
import ctypes as C
import os

import numpy as np
import pytest

from test_planner_cpu import shim  # noqa: F401  (fixture: builds tests/cpu_shim with g++)

HIST, WIN, MINLA = 32768, 65536, 262


def tables(lib, n, flushes, finish=True):
    fp = np.array(flushes, dtype=np.uint64)
    cap = n // HIST + 8
    zones = np.zeros(cap, dtype=np.uint32)
    tiles = np.zeros(3 * cap, dtype=np.uint32)
    pcap = len(flushes) + 2
    pieces = np.zeros(7 * pcap, dtype=np.uint32)
    scap = cap + 2 * pcap
    segs = np.zeros(2 * scap, dtype=np.uint32)
    nb, ns, nt, npc, nsg = (C.c_uint32() for _ in range(5))
    lib.shim_stream_tables.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_void_p]
    lib.shim_stream_tables(n, fp.ctypes.data if len(fp) else None, len(fp), int(finish), C.byref(nb), C.byref(ns),
                           zones.ctypes.data, cap, tiles.ctypes.data, cap, C.byref(nt), pieces.ctypes.data, pcap,
                           C.byref(npc), segs.ctypes.data, scap, C.byref(nsg))
    return dict(n_blocks=nb.value, zones=zones[:ns.value].tolist(), tiles=tiles[:3 * nt.value].reshape(-1, 3).tolist(),
                pieces=pieces[:7 * npc.value].reshape(-1, 7).tolist(), segs=segs[:2 * nsg.value].reshape(-1, 2).tolist())


def model_bases(n, flushes, finish, rng):
    """base[p] = window start (multiple of 32768) when the reference visits position p, for the
    positions it does visit with a random parse.  Follows the control flow of Deflate.write /
    flush / finish literally; token steps are random (1..258, never beyond the lookahead)."""
    base, wp, rp = 0, 0, 0   # absolute
    seen = {}

    def tokenize(flush):
        nonlocal rp
        while True:
            la = wp - rp
            if not (la > (0 if flush else MINLA)):
                break
            seen[rp] = base
            rp += min(la, int(rng.integers(1, 259)) if rng.random() < 0.5 else 1)

    def write(k):
        nonlocal base, wp
        off = 0
        while True:
            room = base + WIN - wp
            if room == 0:
                tokenize(False)
                base += HIST          # slide (deflate.zig:291-294)
                continue
            c = min(k - off, room)
            wp += c
            off += c
            tokenize(False)
            if c < room:
                break

    prev = 0
    for f in flushes:
        write(f - prev)
        tokenize(True)
        prev = f
    if finish:
        write(n - prev)
        tokenize(True)
    return seen


def formula_base(p, zones):
    return HIST * sum(1 for z in zones if p >= z)


CASES = [
    (70000, []), (65536, []), (65535 + 32768, []), (200000, []), (98304, []), (98303, []),
    (70000, [65400]), (70000, [65535]), (70000, [65536]), (70000, [65537]), (140000, [65300, 98000, 98100]),
    (140000, [65274, 65275, 98042]), (300000, [1, 2, 3]), (300000, list(range(0, 300001, 30011))),
    (131072, [131072]), (131072, [65536 + 32768]),
]


def test_slide_table_matches_a_model_of_the_reference_window(shim):  # noqa: F811
    rng = np.random.default_rng(3)
    cases = list(CASES)
    for _ in range(40):
        n = int(rng.integers(65536, 400000))
        marks = [65274, 65536, 98042, 98304, 130810, 131072]
        fl = sorted({int(min(n, max(0, rng.choice(marks) + rng.integers(-400, 400)))) for _ in range(int(rng.integers(0, 5)))})
        cases.append((n, fl))
    for n, fl in cases:
        t = tables(shim, n, fl)
        for _ in range(2):
            seen = model_bases(n, fl, True, rng)
            bad = [(p, b, formula_base(p, t["zones"])) for p, b in seen.items() if b != formula_base(p, t["zones"])]
            assert not bad, (n, fl, t["zones"], bad[:3])


def test_tables_cover_the_stream(shim):  # noqa: F811
    for n, fl in CASES + [(10, [3, 3, 10]), (0, [0]), (5, [])]:
        for finish in (True, False):
            fls = fl if finish else sorted(set(fl) | {n})
            t = tables(shim, n, fls, finish)
            # pieces: consecutive, end at n, one marker per flush, final flag only on the last
            pos = 0
            blocks = 0
            for i, (start, end, fb, nb, seg0, nseg, flags) in enumerate(t["pieces"]):
                assert start == pos and end >= start and fb == blocks
                assert nb == (end - start) // 32768 + 1 + (1 if flags & 2 else 0)
                segs = t["segs"][seg0:seg0 + nseg]
                assert all(s[0] == i for s in segs)
                cover = [(max(h0, start), min(h0 + 32768, end)) for _, h0 in segs]
                assert all(a < b for a, b in cover)
                assert sum(b - a for a, b in cover) == end - start
                assert all(h0 % 32768 == 0 for _, h0 in segs)
                pos, blocks = end, blocks + nb
            assert pos == (n if finish else fls[-1]) and blocks == t["n_blocks"]
            assert [p[6] & 1 for p in t["pieces"]] == [0] * (len(t["pieces"]) - 1) + [1 if finish else 0]
            assert len(t["pieces"]) == len(fls) + (1 if finish else 0)
            # tiles: targets partition [0, n)
            got = []
            for w0, tgt0, zone in t["tiles"]:
                got.append((w0 + tgt0, min(w0 + 65536, max(n, 0))))
                assert 65274 <= zone <= 65536
            flat = sorted(got)
            assert flat[0][0] == 0 and all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1))
            assert flat[-1][1] >= n
