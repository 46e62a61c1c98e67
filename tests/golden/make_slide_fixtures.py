#!/usr/bin/env python3
"""An independent, slow, pure-Python model of the reference's level 4-9 compressor front end -- deflate.zig's
`compress` / `tokenize` / `findMatch` / `flushTokens` with SlidingWindow.zig and Lookup.zig -- written from the Zig
sources (file:line in the comments), NOT from oracle/flate_oracle.c.  It exists to give the oracle's window-slide path
(inputs beyond 64 KiB: the window slides by 32 KiB, chain entries below the new start become null, the raw slice of a
block is gone after a slide) a second opinion: no vector the reference itself holds is longer than 65 535 bytes.

Run in the build container (`python tests/golden/make_slide_fixtures.py`); writes tests/golden/slide/<name>.bin (the
input) and tests/golden/slide/fixtures.json: per (input, level) the number of tokens, the SHA-256 of the token list
(uint32 little endian, the encoding of tests/golden/block_writer_tokens.json: literal = byte << 15, match = 1 << 23 |
(length - 3) << 15 | (distance - 1)) and, for every flushTokens, (tokens, final, has raw input, slice start in the
stream, slice length).  tests/test_oracle_slide_pins.py holds the oracle to them.
"""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HIST = 32768                 # consts.zig history.len
BUF = 2 * HIST               # SlidingWindow.zig:12
MIN_LOOKAHEAD = 4 + 258      # SlidingWindow.zig:13 (match.min_length + match.max_length)
MAX_RP = BUF - MIN_LOOKAHEAD  # SlidingWindow.zig:14
MAX_TOKENS = 1 << 15         # consts.zig deflate.tokens
LEVELS = {4: (4, 4, 16, 16), 5: (8, 16, 32, 32), 6: (8, 16, 128, 128), 7: (8, 32, 128, 256),
          8: (32, 128, 258, 1024), 9: (32, 258, 258, 4096)}  # good, lazy, nice, chain: deflate.zig:41-52


def hashu(v):  # Lookup.zig:82-84
    return ((v * 0x9E3779B1) & 0xFFFFFFFF) >> 17


class Model:
    def __init__(self, level):
        self.good, self.lazy, self.nice, self.chain_max = LEVELS[level]
        self.buf = bytearray(BUF)
        self.wp = self.rp = 0
        self.fp = 0
        self.head = [0] * 32768
        self.chain = [0] * BUF
        self.tokens = []          # the block in progress
        self.all_tokens = []
        self.blocks = []
        self.prev_match = None    # (distance, length)
        self.prev_literal = None
        self.slid = 0             # bytes the window has slid in total

    # ---- Lookup.zig
    def lk_set(self, h, pos):  # :35-40
        p = self.head[h]
        self.head[h] = pos
        self.chain[pos] = p
        return p

    def lk_add(self, lh_len, pos):  # :23-27 (data = buffer[pos:pos + lh_len])
        if lh_len < 4:
            return 0
        b = self.buf
        return self.lk_set(hashu(b[pos] << 24 | b[pos + 1] << 16 | b[pos + 2] << 8 | b[pos + 3]), pos)

    def lk_bulk_add(self, start, data_len, count, pos):  # :55-72: data = buffer[start:start + data_len]
        if count == 0 or data_len < 4:
            return
        b = self.buf
        hb = b[start] << 24 | b[start + 1] << 16 | b[start + 2] << 8 | b[start + 3]
        self.lk_set(hashu(hb), pos)
        i = pos
        for j in range(4, min(count + 3, data_len)):
            hb = ((hb << 8) | b[start + j]) & 0xFFFFFFFF
            i += 1
            self.lk_set(hashu(hb), i)

    def lk_slide(self, n):  # :43-51 (saturating subtraction)
        self.head = [v - n if v > n else 0 for v in self.head]
        for i in range(n):
            v = self.chain[i + n]
            self.chain[i] = v - n if v > n else 0

    # ---- SlidingWindow.zig
    def win_match(self, prev_pos, curr_pos, min_len):  # :81-104
        b = self.buf
        max_len = min(self.wp - curr_pos, 258)
        i = min_len
        if i > 0:
            if max_len <= i:
                return 0
            while True:
                if b[prev_pos + i] != b[curr_pos + i]:
                    return 0
                if i == 0:
                    break
                i -= 1
            i = min_len
        while i < max_len and b[prev_pos + i] == b[curr_pos + i]:
            i += 1
        return i if i >= 4 else 0

    # ---- deflate.zig
    def find_match(self, pos, lh_len, min_len):  # :233-266
        length = min_len
        prev_pos = self.lk_add(lh_len, pos)
        match = None
        chain = self.chain_max
        if length >= self.good:
            chain >>= 2
        while prev_pos > 0 and chain > 0:
            distance = pos - prev_pos
            if distance > 32768:
                break
            new_len = self.win_match(prev_pos, pos, length)
            if new_len > length:
                match = (distance, new_len)
                if new_len >= self.nice:
                    return match
                length = new_len
            prev_pos = self.chain[prev_pos]
            chain -= 1
        return match

    def flush_tokens(self, final):  # :268-288 (what the block writer is handed; it is pinned by its own goldens)
        has = self.fp >= 0  # SlidingWindow.zig:119-123
        self.blocks.append((len(self.tokens), 1 if final else 0, 1 if has else 0, self.slid + self.fp if has else 0,
                            self.rp - self.fp if has else 0))
        self.all_tokens += self.tokens
        self.tokens = []
        self.fp = self.rp  # SlidingWindow.zig:113-115

    def add_token(self, t):  # :227-230
        self.tokens.append(t)
        if len(self.tokens) == MAX_TOKENS:
            self.flush_tokens(False)

    def add_prev_literal(self):  # :214-216
        if self.prev_literal is not None:
            self.add_token(self.prev_literal << 15)  # Token.zig: kind 0, len_lit = the byte

    def add_match(self, m):  # :220-225
        self.add_token(1 << 23 | (m[1] - 3) << 15 | (m[0] - 1))
        self.prev_literal = None
        self.prev_match = None
        return m[1]

    def tokenize(self, flush):  # :154-205; flush: None (deflate.zig .none) | "flush" | "final"
        should_flush = flush is not None
        while self.wp - self.rp > (0 if should_flush else MIN_LOOKAHEAD):  # SlidingWindow.zig:58-62
            lh_len = self.wp - self.rp
            step = 1
            pos = self.rp
            literal = self.buf[pos]
            min_len = self.prev_match[1] if self.prev_match else 0
            m = self.find_match(pos, lh_len, min_len)
            if m:
                self.add_prev_literal()
                if m[1] >= self.lazy:
                    step = self.add_match(m)
                else:
                    self.prev_literal = literal
                    self.prev_match = m
            else:
                if self.prev_match:
                    step = self.add_match(self.prev_match) - 1
                else:
                    self.add_prev_literal()
                    self.prev_literal = literal
            # windowAdvance :207-211: lookup.bulkAdd(lh[1..], step - 1, pos + 1); win.advance(step)
            self.lk_bulk_add(pos + 1, lh_len - 1, step - 1, pos + 1)
            self.rp += step
        if should_flush:
            assert self.prev_match is None
            self.add_prev_literal()
            self.prev_literal = None
            self.flush_tokens(flush == "final")

    def slide(self):  # :291-294, SlidingWindow.zig:36-44
        assert self.rp >= MAX_RP and self.wp >= self.rp
        n = self.wp - HIST
        self.buf[0:n] = self.buf[HIST:self.wp]
        self.rp -= HIST
        self.wp -= HIST
        self.fp -= HIST
        self.slid += HIST
        self.lk_slide(n)

    def compress(self, data):  # :304-321 with a reader over `data`
        off = 0
        while True:
            room = BUF - self.wp
            if room == 0:
                self.tokenize(None)
                self.slide()
                continue
            n = min(room, len(data) - off)
            self.buf[self.wp:self.wp + n] = data[off:off + n]
            off += n
            self.wp += n
            self.tokenize(None)
            if n < room:
                break


def splitmix(seed, n):
    out = bytearray()
    x = seed & 0xFFFFFFFFFFFFFFFF
    while len(out) < n:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        out += struct.pack("<Q", z ^ (z >> 31))
    return bytes(out[:n])


def slide_edge(seed, steps, a=65273, total=140000, base=34000):
    """The window's LAST target before a slide (position a < 65274 = 65536 - min_lookahead) starts a lazy chain of `steps` improving
    matches -- 4 bytes at a, 5 at a + 1, ... -- that ends in a 258-byte match at a + steps >= 65279: tokenize leaves the loop at a
    (SlidingWindow.zig:56-60 gives no lookahead any more), the window slides, and the calls at a + 1 ... are made with the refilled
    window: the match at a + steps has all its 258 bytes, not the 65536 - (a + steps) the window held before the slide."""
    r = splitmix(seed, total + 4000)
    junk = bytes(b | 0x80 for b in r[:total])          # never part of T
    T = bytes(b & 0x7F for b in r[total:total + steps + 258 + 8])
    pieces = [T[k:2 * k + 4] for k in range(steps)] + [T[steps:steps + 258]]
    d = bytearray(junk)
    pos = base
    for pc in pieces:
        d[pos:pos + len(pc)] = pc
        pos += len(pc) + 40
    d[a:a + len(T)] = T
    return bytes(d)


def inputs():
    """Inputs of 100-300 KB whose matches reach across the slide boundaries."""
    r = splitmix(1, 400000)
    words = [bytes(97 + (r[i + k] % 26) for k in range(2 + r[i] % 9)) for i in range(0, 6000, 12)]
    text = bytearray()
    i = 6000
    while len(text) < 300000:
        text += words[(r[i] | r[i + 1] << 8) % len(words)] + (b" " if r[i + 2] % 11 else b".\n")
        i = 6000 + (i - 6000 + 3) % 390000
    text = bytes(text)
    block = splitmix(2, 7000)
    recs = b"".join(struct.pack("<IHH", 1000 + k, k % 7, 0) + block[(k * 37) % 6000:(k * 37) % 6000 + 24] for k in range(9000))
    far = splitmix(3, 30000)  # a block that comes back 32768 +- a few bytes later: candidates on both sides of the distance limit
    return {
        "text250k": text[:250003],
        "text_tail": text[:65536 + 32768 * 3 + 263],  # ends a few bytes after a slide's refill
        "records": recs[:200000],
        "far_copies": (far + splitmix(4, 2768 - 5) + far + splitmix(5, 2768 + 3) + far + splitmix(6, 40000) + far)[:180000],
        "zeros_then_text": bytes(70000) + text[:60000] + bytes(33000) + text[1000:20000],
        "noise_with_runs": b"".join(splitmix(10 + k, 900) + bytes([k % 3]) * (300 + 17 * (k % 40)) for k in range(110))[:150000],
        "slide_edge6": slide_edge(21, 6),      # round 6: the defect of k_lz_parse<true> (a match of 257 instead of 258 bytes at 65279)
        "slide_edge30": slide_edge(22, 30, total=100000 + 65536),
    }


def main():
    out_dir = os.path.join(HERE, "slide")
    os.makedirs(out_dir, exist_ok=True)
    fixtures = {}
    for name, data in inputs().items():
        with open(os.path.join(out_dir, name + ".bin"), "wb") as f:
            f.write(data)
        for level in (4, 6, 9):
            m = Model(level)
            m.compress(data)
            m.tokenize("flush")  # tests/_oracle.tokenize: write, then flush
            toks = m.all_tokens
            sha = hashlib.sha256(struct.pack("<%dI" % len(toks), *toks)).hexdigest()
            fixtures["%s@%d" % (name, level)] = {"bytes": len(data), "tokens": len(toks), "sha256": sha, "blocks": m.blocks}
            sys.stderr.write("%s level %d: %d bytes, %d tokens, %d blocks\n" % (name, level, len(data), len(toks), len(m.blocks)))
    with open(os.path.join(out_dir, "fixtures.json"), "w") as f:
        json.dump(fixtures, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
