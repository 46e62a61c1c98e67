"""Deterministic synthetic corpora.  numpy only; PRNG = splitmix64
(counter based, so every piece is reproducible from (seed, index) alone).

  zeros(n)             Z(n): n zero bytes
  text(seed, n)        T: "enwik-like" -- Zipf draws from a 4096-word vocabulary with
                       punctuation and wiki/XML-ish markup tokens
  silesia_like(seed,n) S: mixed segments (text, binary records, tag-heavy XML, random, sparse zeros)
  tar_like(seed, n)    TAR: ustar headers + text bodies + zero padding (stand-in for ziglang.tar)
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
SEED_TEXT = 0x5EED0002
SEED_SILESIA = 0x5EED0005
SEED_TAR = 0x5EED0003


def splitmix64(seed, n, start=0):
    """n 64-bit draws: the splitmix64 stream of `seed`, positions start .. start+n-1."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed, n, start=0):
    return (splitmix64(seed, n, start) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def zeros(n):
    return np.zeros(n, dtype=np.uint8)


_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_W = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0,
                      1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_MARKUP = [b"<page>", b"</page>", b"<title>", b"</title>", b"<id>", b"</id>", b"[[", b"]]", b"<text>", b"</text>",
           b"&quot;", b"&amp;", b"{{", b"}}", b"==", b"''"]
_vocab_cache = {}


def _vocab(seed, size=4096, maxlen=12):
    key = (seed, size)
    if key in _vocab_cache:
        return _vocab_cache[key]
    u = _uniform(seed ^ 0xA5A5, size * (maxlen + 1))
    # word lengths 2..12, mode around 5; the most frequent ranks are biased short
    # (function words), like natural text
    base = 2 + np.minimum(((u[:size] + u[:size][::-1] * 0.7) / 1.7 * 11).astype(np.int64), maxlen - 2)
    rank = np.arange(size)
    lens = np.where(rank < 64, 2 + (base % 3), base)
    cdf = np.cumsum(_LETTER_W) / _LETTER_W.sum()
    chars = _LETTERS[np.searchsorted(cdf, u[size:].reshape(size, maxlen)).clip(0, 25)]
    for i, m in enumerate(_MARKUP):  # sprinkle markup tokens over the mid ranks
        r = 40 + 37 * i
        lens[r] = len(m)
        chars[r, :len(m)] = np.frombuffer(m, dtype=np.uint8)
    w = 1.0 / np.arange(1, size + 1) ** 1.05
    zcdf = np.cumsum(w) / w.sum()
    _vocab_cache[key] = (lens, chars, zcdf)
    return _vocab_cache[key]


_SEPS = np.array([[32, 0], [44, 32], [46, 32], [10, 0]], dtype=np.uint8)
_SEP_LEN = np.array([1, 2, 2, 1])
_SEP_CDF = np.array([0.85, 0.92, 0.97, 1.0])


def _text_piece(seed, piece, n):
    lens, chars, zcdf = _vocab(seed)
    nw = n // 4 + 64
    u = _uniform(seed + 1000003 * (piece + 1), 2 * nw)
    idx = np.searchsorted(zcdf, u[:nw]).clip(0, len(lens) - 1)
    sep = np.searchsorted(_SEP_CDF, u[nw:]).clip(0, 3)
    wl = lens[idx]
    tot = wl + _SEP_LEN[sep]
    start = np.concatenate(([0], np.cumsum(tot)[:-1]))
    keep = start + tot <= n + 16
    idx, sep, wl, start = idx[keep], sep[keep], wl[keep], start[keep]
    out = np.full(n + 32, 32, dtype=np.uint8)
    for k in range(int(wl.max())):
        m = wl > k
        out[start[m] + k] = chars[idx[m], k]
    out[start + wl] = _SEPS[sep, 0]
    two = _SEP_LEN[sep] == 2
    out[start[two] + wl[two] + 1] = _SEPS[sep[two], 1]
    return out[:n]


def text(seed, n, piece_bytes=1 << 22):
    """T(seed, n): n bytes of enwik-like text."""
    out = np.empty(n, dtype=np.uint8)
    for p, off in enumerate(range(0, n, piece_bytes)):
        m = min(piece_bytes, n - off)
        out[off:off + m] = _text_piece(seed, p, m)
    return out


def _records(seed, n):
    k = n // 16 + 1
    rec = np.zeros((k, 16), dtype=np.uint8)
    cnt = np.arange(k, dtype=np.uint32)
    rec[:, 0:4] = cnt.view(np.uint8).reshape(k, 4)
    r = splitmix64(seed, k)
    rec[:, 4] = (r & np.uint64(3)).astype(np.uint8)
    rec[:, 8:12] = np.frombuffer(b"\x10\x00\x20\x00", dtype=np.uint8)
    rec[:, 12] = ((r >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8) * (cnt % 7 == 0)
    return rec.reshape(-1)[:n]


def _xml(seed, n):
    t = text(seed, n)
    tags = [b"<row id=\"", b"\" name=\"", b"\" value=\"", b"\"/>\n  "]
    out = t.copy()
    pos = 0
    i = 0
    while pos + 16 < n:
        tg = np.frombuffer(tags[i % 4], dtype=np.uint8)
        out[pos:pos + len(tg)] = tg[: max(0, min(len(tg), n - pos))]
        pos += len(tg) + 6 + (i * 7) % 9
        i += 1
    return out


def silesia_like(seed, n):
    """S(seed, n): concatenated segments of 64 KiB - 4 MiB: 40% text, 20% binary records,
    15% tag-heavy XML, 15% uniform random, 10% sparse zeros."""
    out = np.empty(n, dtype=np.uint8)
    pos, seg = 0, 0
    r = splitmix64(seed ^ 0x51, 4096)
    while pos < n:
        kind_u = float(r[(2 * seg) % 4096] >> np.uint64(11)) / (1 << 53)
        size = (1 << 16) + int(r[(2 * seg + 1) % 4096] % np.uint64((4 << 20) - (1 << 16)))
        size = min(size, n - pos)
        s = seed + 7919 * (seg + 1)
        if kind_u < 0.40:
            out[pos:pos + size] = text(s, size)
        elif kind_u < 0.60:
            out[pos:pos + size] = _records(s, size)
        elif kind_u < 0.75:
            out[pos:pos + size] = _xml(s, size)
        elif kind_u < 0.90:
            out[pos:pos + size] = (splitmix64(s, (size + 7) // 8).view(np.uint8))[:size]
        else:
            z = np.zeros(size, dtype=np.uint8)
            k = size // 97 + 1
            where = (splitmix64(s, k) % np.uint64(max(size, 1))).astype(np.int64)
            z[where] = (splitmix64(s + 1, k) & np.uint64(0xFF)).astype(np.uint8)
            out[pos:pos + size] = z
        pos += size
        seg += 1
    return out


TAR_BYTES = 177_244_160  # ziglang.tar of the reference's benchmarks (inflate_bench.zig:14)


def tar_like(seed=SEED_TAR, n=TAR_BYTES):
    """TAR(seed): stand-in for ziglang.tar (not obtainable offline): text bodies, every 24 KiB a
    512-byte ustar header, each entry closed by 4 KiB of zero padding."""
    body = text(seed, n)
    hdr = np.zeros(512, dtype=np.uint8)
    hdr[:100] = np.frombuffer(b"src/flate/deflate.zig".ljust(100, b"\0"), dtype=np.uint8)
    hdr[257:263] = np.frombuffer(b"ustar\0", dtype=np.uint8)
    for off in range(0, n - 8192, 24576):
        body[off:off + 512] = hdr
        body[off + 20480:off + 24576] = 0
    return body


def split_offsets(n, chunk):
    """Offsets (n_chunks + 1, uint64) cutting n bytes into `chunk`-byte chunks."""
    k = (n + chunk - 1) // chunk if n else 1
    off = np.minimum(np.arange(k + 1, dtype=np.uint64) * np.uint64(chunk), np.uint64(n))
    return off


# ---- torch twin of text(): bit-identical, runs on the GPU for the 1 GiB bench input ----
def _splitmix64_torch(torch, seed, n, start, device):
    def srl(z, k):  # logical shift right on int64
        return (z >> k) & ((1 << (64 - k)) - 1)

    def wrap(c):  # python int -> two's complement int64 constant
        c &= 0xFFFFFFFFFFFFFFFF
        return c - (1 << 64) if c >= (1 << 63) else c

    i = torch.arange(start + 1, start + n + 1, dtype=torch.int64, device=device)
    z = i * wrap(0x9E3779B97F4A7C15) + wrap(int(seed))
    z = (z ^ srl(z, 30)) * wrap(0xBF58476D1CE4E5B9)
    z = (z ^ srl(z, 27)) * wrap(0x94D049BB133111EB)
    z = z ^ srl(z, 31)
    return z, srl


def _uniform_torch(torch, seed, n, start, device):
    z, srl = _splitmix64_torch(torch, seed, n, start, device)
    return srl(z, 11).to(torch.float64) * (1.0 / (1 << 53))


def text_torch(seed, n, device="cuda", piece_bytes=1 << 22):
    """Same bytes as text(seed, n, piece_bytes), produced with torch ops on `device`."""
    import torch

    lens_np, chars_np, zcdf_np = _vocab(seed)
    lens = torch.from_numpy(lens_np.astype(np.int64)).to(device)
    chars = torch.from_numpy(chars_np.copy()).to(device)
    zcdf = torch.from_numpy(zcdf_np).to(device)
    seps = torch.from_numpy(_SEPS.copy()).to(device)
    sep_len = torch.from_numpy(_SEP_LEN.astype(np.int64)).to(device)
    sep_cdf = torch.from_numpy(_SEP_CDF).to(device)
    out_all = torch.empty(n, dtype=torch.uint8, device=device)
    for piece, off in enumerate(range(0, n, piece_bytes)):
        m_bytes = min(piece_bytes, n - off)
        nw = m_bytes // 4 + 64
        u = _uniform_torch(torch, seed + 1000003 * (piece + 1), 2 * nw, 0, device)
        idx = torch.searchsorted(zcdf, u[:nw]).clamp(0, lens.numel() - 1)
        sep = torch.searchsorted(sep_cdf, u[nw:]).clamp(0, 3)
        wl = lens[idx]
        tot = wl + sep_len[sep]
        cs = torch.cumsum(tot, 0)
        start = cs - tot
        keep = start + tot <= m_bytes + 16
        idx, sep, wl, start = idx[keep], sep[keep], wl[keep], start[keep]
        out = torch.full((m_bytes + 32,), 32, dtype=torch.uint8, device=device)
        for k in range(int(wl.max().item())):
            msk = wl > k
            out[start[msk] + k] = chars[idx[msk], k]
        out[start + wl] = seps[sep, 0]
        two = sep_len[sep] == 2
        out[start[two] + wl[two] + 1] = seps[sep[two], 1]
        out_all[off:off + m_bytes] = out[:m_bytes]
    return out_all
