#!/usr/bin/env python3
"""k_inflate_par against zlib-made streams: which streams it finishes itself, bytes equal, time."""
import os, sys, time, zlib
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import Engine, synth

eng = Engine(0)
rng = np.random.default_rng(5)
cases = []
def add(name, data, level=6, wbits=-15, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    cases.append((name, bytes(data), co.compress(bytes(data)) + co.flush(), {-15: 0, 31: 1, 15: 2}[wbits]))
text = synth.text(synth.SEED_TEXT, 4 << 20).tobytes()
sil = synth.silesia_like(synth.SEED_SILESIA, 8 << 20).tobytes()
add("text1M-l6", text[:1 << 20]); add("text1M-l1", text[:1 << 20], 1); add("text1M-l9", text[1 << 20:2 << 20], 9)
add("text300K-gz", text[:300000], 6, 31); add("text2M-zlib", text[:2 << 20], 6, 15)
add("sil1M-a", sil[:1 << 20]); add("sil1M-b", sil[3 << 20:4 << 20]); add("sil4M-gz", sil[:4 << 20], 6, 31)
add("zeros4M-gz", bytes(4 << 20), 6, 31); add("zeros100K", bytes(100000))
add("rand1M", rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes())
add("huff-only", text[:1 << 20], 6, -15, zlib.Z_HUFFMAN_ONLY); add("rle", text[:1 << 20], 6, -15, zlib.Z_RLE)
add("fixed", text[:200000], 6, -15, zlib.Z_FIXED)
add("rec1M", synth._records(4242, 1 << 20).tobytes()); _rec = synth._records(4242, 1 << 20).tobytes()
cases.append(("rec1M-ours", _rec, eng.compress_many([_rec], 0, 6)[0][0], 0))
add("small", text[:20000]); add("abab", (b"ab" * 300000)); add("period7", (b"abcdefg" * 100000), 9)
bad = 0
for cont in (0, 1, 2):
    grp = [c for c in cases if c[3] == cont]
    if not grp:
        continue
    eng.profile_enable(True); eng.profile_reset()
    t0 = time.perf_counter()
    outs, st, _cons = eng.decompress_many([c[2] for c in grp], cont, 0, caps=[len(c[1]) + 64 for c in grp])
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    for c, o, s in zip(grp, outs, st):
        ok = s == 0 and o == c[1]
        bad += not ok
        print("%-14s cont %d in %8d out %8d status %3d %s" % (c[0], cont, len(c[2]), len(c[1]), s, "ok" if ok else "MISMATCH"))
    print("   kernels:", {k: round(v[0], 3) for k, v in prof.items()}, "wall %.1f ms" % (dt * 1e3))
print("bad", bad)
if os.environ.get("FL_PAR_EACH"):
    for c in cases:
        eng.profile_reset()
        eng.decompress_many([c[2]], c[3], 0, caps=[len(c[1]) + 64])
        prof = eng.profile_read()
        print("%-14s" % c[0], {k: round(v[0], 3) for k, v in prof.items()}, "redo reason", int(eng.phase_cycles()[60]))
if os.environ.get("FL_PAR_PROF"):
    only = [c for c in cases if c[0] == os.environ.get("FL_PAR_CASE", "text1M-l6")]
    tz = eng.phase_cycles().astype(np.int64)
    eng.decompress_many([only[0][2]], 0, 0, caps=[len(only[0][1]) + 64])
    t = eng.phase_cycles().astype(np.int64) - tz
    for k, nm in {20: "hdr: counts + precode lens", 21: "hdr: precode tables", 22: "hdr: code lengths", 23: "hdr: generate x2", 24: "hdr: luts", 32: "loop top", 33: "block header", 34: "stage", 41: "decode (wave 0)", 35: "join + wait", 36: "stitch", 37: "layout", 38: "fill", 39: "resolve",
                  40: "flush+update", 48: "# rounds", 49: "# valid waves", 50: "# out bytes", 51: "# resolve rounds", 52: "end normal", 53: "end eob", 54: "end bail", 55: "end full", 56: "end cut", 57: "nojoin", 61: "decode passes (wave 0)", 62: "... with a long lit code", 63: "... with a long dist code", 58: "nojoin: sum xkind", 59: "nojoin: sum xpos"}.items():
        print("%-16s %12d" % (nm, t[k]))
