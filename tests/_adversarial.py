"""Inputs made to sit ON the tokenizer's and the Huffman builder's thresholds (test infrastructure; tools/threshold_sweep.py and
tools/depth_sweep.py run them in bulk): candidates exactly at, before and behind the chain budget and its quarter, matches of exactly
`good` / `lazy` / `nice` bytes and one less / more with longer ones behind them and better ones at the next positions; symbol
frequencies that grow like Fibonacci numbers or powers of two (trees deeper than 15 / 7 bits).  Every function draws from the
numpy Generator it is given."""
import numpy as np

LV = {4: (4, 4, 16, 16), 5: (8, 16, 32, 32), 6: (8, 16, 128, 128), 7: (8, 32, 128, 256), 8: (32, 128, 258, 1024), 9: (32, 258, 258, 4096)}  # good, lazy, nice, chain  # good, lazy, nice, chain (deflate.zig:44-49)


def junk(rng, n):
    return rng.integers(128, 256, n, dtype=np.uint8).tobytes()


def near(rng, v):
    return max(3, int(v) + int(rng.integers(-2, 3)))


def threshold_input(rng, level, total):
    good, lazy, nice, chain = LV[level]
    S = rng.integers(0, 64, 600, dtype=np.uint8).tobytes()
    copies = []  # farthest first
    # a long candidate, far
    copies.append(S[:near(rng, rng.choice([nice + 4, 258, lazy + 3, 40]))])
    # fillers that share the first 4 (or good + 1) bytes: around the budget
    share = int(rng.choice([4, good, good + 1, 6]))
    budget = int(rng.choice([chain, chain // 4, chain // 4 + 1, chain // 2]))
    k = max(0, budget + int(rng.integers(-4, 3)))
    k = min(k, (30000 - 2000) // (share + 3))
    fill = [S[:share] + junk(rng, 3)[: 1 + int(rng.integers(0, 3))] for _ in range(k)]
    # candidates at the thresholds, nearest
    nearc = [S[:near(rng, rng.choice([good, lazy, nice, good - 1, lazy - 1, nice - 1, 5, 7]))] for _ in range(int(rng.integers(0, 4)))]
    # better matches at the next positions (lazy evaluation)
    nextc = [S[o:o + near(rng, rng.choice([good, lazy, nice, 9, 33, 258]))] for o in (1, 2, 3) if rng.random() < 0.5]
    parts = copies + fill + nearc
    order = list(range(len(nextc)))
    body = bytearray()
    for pc in nextc:
        body += pc + junk(rng, 5)
    for pc in parts:
        body += pc + junk(rng, 2 + int(rng.integers(0, 3)))
    body = bytes(body)
    lead = junk(rng, max(0, total - len(body) - len(S) - 200))
    return lead + body + S + junk(rng, 200 - int(rng.integers(0, 150)))


def skewed(rng, total, growth, nsym):
    """bytes with symbol k about growth^k times (capped by total), shuffled"""
    w = np.array([growth ** k for k in range(nsym)], dtype=np.float64)
    cnt = np.maximum(1, np.floor(w / w.sum() * total)).astype(np.int64)
    syms = rng.permutation(256)[:nsym]
    a = np.repeat(syms.astype(np.uint8), cnt)
    rng.shuffle(a)
    return a[:total].tobytes()


def match_skew(rng, total, growth):
    """match lengths / distances with skewed frequencies: copies of earlier stretches at chosen lengths and distances"""
    base = rng.integers(0, 256, 4000, dtype=np.uint8).tobytes()
    out = bytearray(base)
    lens = [3 + k for k in range(0, 255, 9)]
    w = np.array([growth ** k for k in range(len(lens))]); w /= w.sum()
    while len(out) < total:
        L = int(rng.choice(lens, p=w))
        dist = int(2 ** rng.integers(2, 15)) + int(rng.integers(0, 3))
        dist = min(dist, len(out))
        s = len(out) - dist
        for i in range(L):
            out.append(out[s + i])
        out += rng.integers(0, 256, int(rng.integers(1, 3)), dtype=np.uint8).tobytes()
    return bytes(out[:total])
