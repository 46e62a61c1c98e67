"""Directed DEFLATE streams that fuzzing rarely builds (test infrastructure): fixed-block symbols 286 / 287 and distance codes 30 / 31,
a distance equal to / one more than what has been written, lengths and distances at the decoders' ring edges, stored blocks of 0 and
65535 bytes, a wrong NLEN, a truncated stored block, zlib-made dynamic blocks with long code-length runs (repeats that cross from the
literal into the distance table) and with one / no distance code.  `CASES`: {name: raw deflate stream}."""
import zlib

import numpy as np


class BW:
    def __init__(self): self.b = bytearray(); self.acc = 0; self.n = 0
    def bits(self, v, k):  # LSB first
        self.acc |= (v & ((1 << k) - 1)) << self.n; self.n += k
        while self.n >= 8: self.b.append(self.acc & 255); self.acc >>= 8; self.n -= 8
    def code(self, c, k):  # a Huffman code: MSB first
        for i in range(k - 1, -1, -1): self.bits((c >> i) & 1, 1)
    def done(self):
        if self.n: self.b.append(self.acc & 255); self.acc = 0; self.n = 0
        return bytes(self.b)


def fixed_lit(w, s):
    if s < 144: w.code(0x30 + s, 8)
    elif s < 256: w.code(0x190 + s - 144, 9)
    elif s < 280: w.code(s - 256, 7)
    else: w.code(0xC0 + s - 280, 8)


LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]


def fixed_match(w, length, dist):
    li = max(i for i in range(29) if LBASE[i] <= length)
    if length == 258: li = 28
    fixed_lit(w, 257 + li); w.bits(length - LBASE[li], LEXT[li])
    di = max(i for i in range(30) if DBASE[i] <= dist)
    w.code(di, 5); w.bits(dist - DBASE[di], DEXT[di])


def fixed_block(ops, final=1):
    w = BW(); w.bits(final, 1); w.bits(1, 2)
    for op in ops:
        if op[0] == "L": fixed_lit(w, op[1])
        elif op[0] == "M": fixed_match(w, op[1], op[2])
        elif op[0] == "RAWLIT": fixed_lit(w, op[1])          # any symbol number, 286 / 287 too
        elif op[0] == "RAWDIST":                              # a length symbol, then a raw 5-bit distance code
            fixed_lit(w, 257); w.code(op[1], 5); w.bits(0, 13)
    fixed_lit(w, 256)
    return w.done()


CASES = cases = {}
lits = [("L", 65 + i % 26) for i in range(40)]
cases["sym286"] = fixed_block(lits + [("RAWLIT", 286)])
cases["sym287"] = fixed_block(lits + [("RAWLIT", 287)])
cases["dist30"] = fixed_block(lits + [("RAWDIST", 30)])
cases["dist31"] = fixed_block(lits + [("RAWDIST", 31)])
cases["dist_eq_written"] = fixed_block(lits + [("M", 10, 40)])
cases["dist_written_plus_1"] = fixed_block(lits + [("M", 10, 41)])
cases["dist_first_byte"] = fixed_block([("M", 3, 1)])
for dist in (2047, 2048, 2049, 4096, 32767, 32768):
    body = [("L", (i * 7) & 255) for i in range(dist)]
    for ln in (3, 258, 257):
        cases["ring_d%d_l%d" % (dist, ln)] = fixed_block(body + [("M", ln, dist), ("M", 258, 1), ("M", ln, dist)])
    cases["ring_d%d_plus1" % dist] = fixed_block(body + [("M", 3, dist + 1)])
# stored blocks
def stored(data, final):
    return bytes([final]) + len(data).to_bytes(2, "little") + (len(data) ^ 0xffff).to_bytes(2, "little") + data
cases["stored0_chain"] = b"".join(stored(b"", 0) for _ in range(300)) + stored(b"xyz", 1)
cases["stored65535"] = stored(bytes(range(256)) * 255 + bytes(255), 0) + stored(b"tail", 1)
cases["stored_bad_nlen"] = b"\x01\x05\x00\xfa\xfe" + b"hello"
cases["stored_truncated"] = stored(b"hello world", 1)[:-3]
# zlib-made dynamic streams with long code-length runs (repeats cross the literal / distance boundary) and no / one distance code
rng = np.random.default_rng(3)
for k, d in enumerate([bytes(70000), b"a" * 300 + bytes(range(256)) * 3, rng.integers(0, 4, 5000, dtype=np.uint8).tobytes(), b"abcdefgh" * 4000]):
    for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, strat)
        cases["zlib%d_%d" % (k, strat)] = co.compress(d) + co.flush()

