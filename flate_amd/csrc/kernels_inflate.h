// kernels_inflate.h -- batched inflate: one wavefront per independent stream.
//
// Reference path: Inflate.step / dynamicBlockHeader / dynamicBlock / fixedBlock /
// storedBlock (inflate.zig:89-280), HuffmanDecoder (huffman_decoder.zig:71-175),
// BitReader (bit_reader.zig:46-217), CircularBuffer.writeMatch (CircularBuffer.zig:44-75),
// container parse (container.zig:111-166).  The status returned for a bad stream is
// the error name the reference returns (pinned by its 40-case table, inflate.zig:487-527).
//
// The symbol decode of one stream is inherently serial: every lane of the wave runs
// it in lock step (wave-uniform control flow), lane 0 owns the literal stores, and
// the LZ77 copies and the checksum are spread over the 64 lanes.  Throughput comes
// from many streams in flight (one per wave, several waves per CU).  Output goes
// straight to the caller's buffer -- no 64 KiB ring as in the reference.
#pragma once
#include "kernels_common.h"

struct fl_hdec {
    uint16_t count[16];
    uint16_t symbol[288];
};

// LDS pointers are typed with their address space: with generic pointers the decoder's table
// and ring accesses became FLAT instructions (PMC: 323 k FLAT vs 30 k LDS per wave), i.e. LDS
// traffic through the vector-memory pipe.
#define FL_LDS __attribute__((address_space(3)))

#define FL_INF_LIT_BITS 10
#define FL_INF_DST_BITS 9
#define FL_INF_RING 2048u                       // recent output kept in LDS (power of two)
#define FL_INF_NEAR (FL_INF_RING - 258u - 2u)   // matches up to this distance are served from it
#define FL_INF_FENCE 512u                       // output bytes between two "stores are visible" fences

struct fl_inflate_ws {
    fl_hdec lit, dst, cl;
    uint16_t lit_lut[1u << FL_INF_LIT_BITS];  // symbol | code_bits << 9, 0 = not in the table
    uint16_t dst_lut[1u << FL_INF_DST_BITS];
    uint8_t lens[320];
    uint8_t cl_lens[20];
    uint16_t offs[18];
    uint8_t ring[FL_INF_RING];
};

#define FL_INF_INRING 1024u  // compressed bytes staged in LDS (two 512-byte halves)

// Bit reader (bit_reader.zig:18-219) in position-free form: `left` = bits of the stream not yet
// consumed.  fill(nice) fails only when no bit at all is left (bit_reader.zig:59-67), shift(n)
// when n exceeds what is left (:159-163); peeks beyond the end see zero bits.
struct fl_bitr {
    const uint8_t* data;
    FL_LDS uint32_t* inring;  // stream bytes [in_loaded - 1024, in_loaded), index = offset & 1023
    int64_t left;             // unconsumed bits
    uint64_t buf;             // the next `have` bits, zero beyond the end of the stream
    uint32_t nbytes;
    uint32_t next_byte;  // stream offset of the first byte not yet in `buf`
    uint32_t in_loaded;  // stream bytes staged in LDS so far (a multiple of 512)
    uint32_t pf0, pf1;   // this lane's 8 bytes of the next half, already requested from memory
    uint32_t have;
    uint32_t lane;
};

// this lane's 8 bytes at stream offset `off` (zero beyond the stream)
__device__ __forceinline__ void fl_br_fetch8(const fl_bitr& r, uint32_t off, uint32_t& a, uint32_t& b) {
    a = 0;
    b = 0;
    if ((uint64_t)off + 12 <= r.nbytes) {
        const uint8_t* p = r.data + off;
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
        const uint32_t* w = (const uint32_t*)(p - sh);
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
        a = __builtin_amdgcn_alignbyte(d1, d0, sh);
        b = __builtin_amdgcn_alignbyte(d2, d1, sh);
    } else {
        for (uint32_t k = 0; k < 4; k++) {
            if ((uint64_t)off + k < r.nbytes) a |= (uint32_t)r.data[off + k] << (8 * k);
            if ((uint64_t)off + 4 + k < r.nbytes) b |= (uint32_t)r.data[off + 4 + k] << (8 * k);
        }
    }
}
// stage the 512-byte half that `pf` holds and request the one after it
__device__ __forceinline__ void fl_br_commit_half(fl_bitr& r) {
    const uint32_t slot = ((r.in_loaded & (FL_INF_INRING - 1)) >> 2) + 2 * r.lane;
    r.inring[slot] = r.pf0;
    r.inring[slot + 1] = r.pf1;
    r.in_loaded += 512;
    fl_br_fetch8(r, r.in_loaded + 8 * r.lane, r.pf0, r.pf1);
    fl_lds_order();
}
// (re)start reading at stream byte `byte` with an empty bit buffer
__device__ __forceinline__ void fl_br_seek(fl_bitr& r, uint32_t byte) {
    fl_lds_order();
    r.buf = 0;
    r.have = 0;
    r.next_byte = byte;
    r.in_loaded = byte & ~511u;
    fl_br_fetch8(r, r.in_loaded + 8 * r.lane, r.pf0, r.pf1);
    fl_br_commit_half(r);
}

// Keep at least 33 valid bits in the buffer (the most one decode step consumes between two
// refills, bit_reader.zig:46-68 fills for 5 + 15 + 13).  The bytes come from the LDS staging
// ring; the next half is always already on its way.
__device__ __forceinline__ void fl_br_refill(fl_bitr& r) {
    if (r.have <= 32) {
        if (r.next_byte + 8 > r.in_loaded) fl_br_commit_half(r);
        const uint32_t i = (r.next_byte & (FL_INF_INRING - 1)) >> 2;
        const uint32_t lo = r.inring[i], hi = r.inring[(i + 1) & (FL_INF_INRING / 4 - 1)];
        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, r.next_byte & 3);
        r.buf |= (uint64_t)w << r.have;
        r.have += 32;
        r.next_byte += 4;
    }
}
__device__ __forceinline__ int fl_br_fill(const fl_bitr& r, uint32_t nice) {
    return (nice > 0 && r.left <= 0) ? 1 : 0;  // EndOfStream
}
__device__ __forceinline__ uint32_t fl_br_peek(fl_bitr& r, uint32_t n) {  // n <= 32
    fl_br_refill(r);
    return (uint32_t)(r.buf & ((1ull << n) - 1));
}
__device__ __forceinline__ int fl_br_shift(fl_bitr& r, uint32_t n) {
    if ((int64_t)n > r.left) return 1;
    fl_br_refill(r);
    r.left -= n;
    r.buf >>= n;
    r.have -= n;
    return 0;
}
__device__ __forceinline__ int fl_br_read(fl_bitr& r, uint32_t n, uint32_t& v) {  // readF(U, 0)
    if (fl_br_fill(r, n)) return 1;
    v = fl_br_peek(r, n);
    return fl_br_shift(r, n);
}
__device__ __forceinline__ void fl_br_align(fl_bitr& r) {  // bit_reader.zig:189-192
    const uint32_t k = (uint32_t)r.left & 7;  // the stream is a whole number of bytes
    if (k) {
        fl_br_refill(r);
        r.left -= k;
        r.buf >>= k;
        r.have -= k;
    }
}
// bytes consumed so far, rounded up
__device__ __forceinline__ uint64_t fl_br_consumed(const fl_bitr& r) {
    const uint64_t pos = (uint64_t)r.nbytes * 8 - (uint64_t)r.left;
    return (pos + 7) >> 3;
}

// huffman_decoder.zig:71-153 (checkCompletnes + canonical symbol order).  Runs on all
// lanes redundantly except the LDS writes (lane 0).
__device__ __forceinline__ int fl_hdec_generate(FL_LDS fl_hdec* d, const FL_LDS uint8_t* lens, FL_LDS uint16_t* offs,
                                                int n, int alphabet, int max_code_bits, uint32_t lane) {
    if (alphabet == 286 && lens[256] == 0) return 10;  // MissingEndOfBlockCode
    uint32_t cnt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cnt[i] = 0;
    int mx = 0;
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (l == 0) continue;
        if (l > mx) mx = l;
#pragma unroll
        for (int k = 1; k < 16; k++)
            if (k == l) cnt[k]++;
    }
    if (mx != 0) {
        int left = 1;
        for (int len = 1; len <= max_code_bits; len++) {
            left <<= 1;
            uint32_t cl = 0;
#pragma unroll
            for (int k = 1; k < 16; k++)
                if (k == len) cl = cnt[k];
            if ((int)cl > left) return 8;  // OversubscribedHuffmanTree
            left -= (int)cl;
        }
        if (left > 0) {
            if (!(max_code_bits > 7 && mx == (int)cnt[1])) return 9;  // IncompleteHuffmanTree
        }
    }
    fl_wave_lds_sync();
    if (lane == 0) {
        offs[1] = 0;
        d->count[0] = 0;
        for (int len = 1; len < 16; len++) {
            uint32_t cl = 0;
#pragma unroll
            for (int k = 1; k < 16; k++)
                if (k == len) cl = cnt[k];
            d->count[len] = (uint16_t)cl;
            offs[len + 1] = (uint16_t)(offs[len] + cl);
        }
        for (int i = 0; i < n; i++)
            if (lens[i] != 0) d->symbol[offs[lens[i]]++] = (uint16_t)i;
    }
    fl_wave_lds_sync();
    return 0;
}

// huffman_decoder.zig:156-175: the symbol whose code is a prefix of `peek`
// (stream bit order), or InvalidCode.
__device__ __forceinline__ int fl_hdec_find(const FL_LDS fl_hdec* d, uint32_t peek, int max_code_bits, uint32_t& sym,
                                            uint32_t& code_bits) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= max_code_bits; len++) {
        code |= (int)(peek & 1);
        peek >>= 1;
        const int count = d->count[len];
        if (code - count < first) {
            sym = d->symbol[index + (code - first)];
            code_bits = (uint32_t)len;
            return 0;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return 7;  // InvalidCode
}

// Fill a 2^bits-entry table: entry[i] = the symbol whose code is a prefix of i (stream bit
// order) when that code has at most `bits` bits, else 0.  Every lane decodes its share of the
// indices with the same walk as fl_hdec_find, so table and walk cannot disagree.
__device__ __forceinline__ void fl_hdec_build_lut(const FL_LDS fl_hdec* d, FL_LDS uint16_t* lut, int bits,
                                                  uint32_t lane) {
    for (uint32_t i = lane; i < (1u << bits); i += 64) {
        uint32_t sym, cb;
        uint16_t e = 0;
        if (fl_hdec_find(d, i, bits, sym, cb) == 0) e = (uint16_t)(sym | (cb << 9));
        lut[i] = e;
    }
    fl_wave_lds_sync();
}

__device__ __forceinline__ uint32_t fl_rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32 - n); }

struct fl_inf_out {
    uint8_t* out;
    FL_LDS uint8_t* ring;  // LDS copy of the last FL_INF_RING output bytes
    uint64_t cap;
    uint64_t wp;
    uint64_t fenced;  // every output byte below this offset is visible to this wave's loads
};

// Called when wp has moved: once per FL_INF_FENCE bytes wait for the outstanding stores, so that
// far matches may read the output buffer without waiting.
__device__ __forceinline__ void fl_inf_advance(fl_inf_out& o) {
    if (o.wp - o.fenced >= 2 * FL_INF_FENCE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        o.fenced = o.wp;
    }
}

// CircularBuffer.zig:44-75, spread over the wave.  Near matches copy out of the LDS ring; far
// ones read the output buffer, whose bytes that far back are already fenced.
__device__ __forceinline__ int fl_inf_match(fl_inf_out& o, uint32_t length, uint32_t distance, uint32_t lane) {
    if (o.wp < distance || length < 3 || length > 258 || distance < 1 || distance > 32768) return 11;
    if (o.wp + length > o.cap) return 100;
    const uint32_t wp = (uint32_t)o.wp;
    uint8_t* to = o.out + o.wp;
    const bool near_ = distance <= FL_INF_NEAR;
    if (!near_ && o.wp - distance + length > o.fenced) {  // the source is younger than the last fence
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        o.fenced = o.wp;
    }
    const uint8_t* from = o.out + o.wp - distance;  // far matches: distance > length, no overlap
    for (uint32_t i0 = 0; i0 < length; i0 += 64) {  // one trip for lengths up to 64
        const uint32_t i = i0 + lane;
        uint32_t byte = 0;
        if (i < length) {
            if (near_) {
                // the source bytes repeat with period `distance` when the match overlaps itself
                const uint32_t si = distance >= length ? i : (i % distance);
                byte = o.ring[(wp - distance + si) & (FL_INF_RING - 1)];
            } else {
                byte = from[i];
            }
        }
        fl_lds_order();
        if (i < length) {
            to[i] = (uint8_t)byte;
            o.ring[(wp + i) & (FL_INF_RING - 1)] = (uint8_t)byte;
        }
        fl_lds_order();
    }
    o.wp += length;
    fl_inf_advance(o);
    return 0;
}
__device__ __forceinline__ int fl_inf_literal(fl_inf_out& o, uint32_t byte, uint32_t lane) {
    if (o.wp >= o.cap) return 100;
    // every lane stores the same byte to the same address: one transaction, no exec juggling
    o.out[o.wp] = (uint8_t)byte;
    o.ring[(uint32_t)o.wp & (FL_INF_RING - 1)] = (uint8_t)byte;
    o.wp++;
    return 0;
}

#define FL_TRY(expr)            \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

// inflate.zig:123-140 (the caller has already filled)
__device__ __forceinline__ int fl_inf_length(fl_bitr& r, uint32_t code, uint32_t& length) {
    if (code > 28) return 7;
    const uint32_t eb = fl_len_extra_bits(code);
    length = fl_len_base_scaled(code) + 3;
    if (eb) {
        length += fl_br_peek(r, eb);
        return fl_br_shift(r, eb);
    }
    return 0;
}
__device__ __forceinline__ int fl_inf_distance(fl_bitr& r, uint32_t code, uint32_t& distance) {
    if (code > 29) return 7;
    const uint32_t eb = fl_dist_extra_bits(code);
    distance = fl_dist_base_scaled(code) + 1;
    if (eb) {
        distance += fl_br_peek(r, eb);
        return fl_br_shift(r, eb);
    }
    return 0;
}

// inflate.zig:89-102
__device__ __forceinline__ int fl_inf_stored(fl_bitr& r, fl_inf_out& o, uint32_t lane) {
    fl_br_align(r);
    uint32_t len, nlen;
    FL_TRY(fl_br_read(r, 16, len));
    FL_TRY(fl_br_read(r, 16, nlen));
    if (len != ((~nlen) & 0xffff)) return 13;
    if ((int64_t)len * 8 > r.left) return 1;
    if (o.wp + len > o.cap) return 100;
    const uint32_t src_off = (uint32_t)fl_br_consumed(r);  // byte aligned here
    const uint8_t* s = r.data + src_off;
    for (uint32_t i = lane; i < len; i += 64) o.out[o.wp + i] = s[i];
    // the ring mirrors the last bytes of the output
    {
        const uint32_t tail = len < FL_INF_RING ? len : FL_INF_RING;
        fl_lds_order();
        for (uint32_t i = lane; i < tail; i += 64) {
            const uint64_t off = o.wp + len - tail + i;
            o.ring[(uint32_t)off & (FL_INF_RING - 1)] = s[len - tail + i];
        }
        fl_lds_order();
    }
    o.wp += len;
    fl_inf_advance(o);
    r.left -= (int64_t)len * 8;
    fl_br_seek(r, src_off + len);
    return 0;
}

// bit_reader.zig:205-217 + inflate.zig:104-121
__device__ __forceinline__ int fl_inf_fixed(fl_bitr& r, fl_inf_out& o, uint32_t lane) {
    for (;;) {
        FL_TRY(fl_br_fill(r, 9));
        const uint32_t code7 = fl_rev_bits(fl_br_peek(r, 7), 7);
        FL_TRY(fl_br_shift(r, 7));
        uint32_t code;
        if (code7 <= 0x17) {
            code = code7 + 256;
        } else if (code7 <= 0x5f) {
            const uint32_t e = fl_br_peek(r, 1);
            FL_TRY(fl_br_shift(r, 1));
            code = (code7 << 1) + e - 0x30;
        } else if (code7 <= 0x63) {
            const uint32_t e = fl_br_peek(r, 1);
            FL_TRY(fl_br_shift(r, 1));
            code = ((code7 - 0x60) << 1) + e + 280;
        } else {
            const uint32_t e = fl_rev_bits(fl_br_peek(r, 2), 2);
            FL_TRY(fl_br_shift(r, 2));
            code = ((code7 - 0x64) << 2) + e + 144;
        }
        if (code <= 255) {
            FL_TRY(fl_inf_literal(o, code, lane));
        } else if (code == 256) {
            return 0;
        } else if (code <= 285) {
            FL_TRY(fl_br_fill(r, 5 + 5 + 13));
            uint32_t length, distance;
            FL_TRY(fl_inf_length(r, code - 257, length));
            const uint32_t dcode = fl_rev_bits(fl_br_peek(r, 5), 5);
            FL_TRY(fl_br_shift(r, 5));
            FL_TRY(fl_inf_distance(r, dcode, distance));
            FL_TRY(fl_inf_match(o, length, distance, lane));
        } else {
            return 7;
        }
    }
}

// inflate.zig:188-216 + the read loops of :161-180
__device__ __forceinline__ int fl_inf_read_lens(fl_bitr& r, FL_LDS fl_inflate_ws* ws, uint32_t base, uint32_t lens_len, uint32_t want,
                                uint32_t boundary, bool& crossed, uint32_t lane) {
    uint32_t pos = 0;
    FL_LDS uint8_t* lens = ws->lens + base;
    while (pos < want) {
        FL_TRY(fl_br_fill(r, 7));
        uint32_t sym, cb;
        FL_TRY(fl_hdec_find(&ws->cl, fl_br_peek(r, 7), 7, sym, cb));
        FL_TRY(fl_br_shift(r, cb));
        if (boundary && sym == 16 && pos == boundary) crossed = true;
        if (pos >= lens_len) return 14;
        uint32_t adv, v;
        if (sym == 16) {
            FL_TRY(fl_br_read(r, 2, v));
            adv = v + 3;
            if (pos == 0 || pos + adv > lens_len) return 14;
            fl_wave_lds_sync();
            const uint8_t prev = lens[pos - 1];
            fl_wave_lds_sync();
            if (lane == 0)
                for (uint32_t i = 0; i < adv; i++) lens[pos + i] = prev;
        } else if (sym == 17) {
            FL_TRY(fl_br_read(r, 3, v));
            adv = v + 3;
        } else if (sym == 18) {
            FL_TRY(fl_br_read(r, 7, v));
            adv = v + 11;
        } else {
            if (lane == 0) lens[pos] = (uint8_t)sym;
            adv = 1;
        }
        if (boundary && pos < boundary && pos + adv > boundary) crossed = true;
        pos += adv;
    }
    if (pos > want) return 14;
    return 0;
}

// inflate.zig:144-184.  flags bit0: reference-strict Q6 (two separate length lists).
__device__ __forceinline__ int fl_inf_dynamic_header(fl_bitr& r, FL_LDS fl_inflate_ws* ws, int flags, uint32_t lane) {
    uint32_t v;
    FL_TRY(fl_br_read(r, 5, v));
    const uint32_t hlit = v + 257;
    FL_TRY(fl_br_read(r, 5, v));
    const uint32_t hdist = v + 1;
    FL_TRY(fl_br_read(r, 4, v));
    const uint32_t hclen = v + 4;
    if (hlit > 286 || hdist > 30) return 14;
    fl_wave_lds_sync();
    if (lane < 20) ws->cl_lens[lane] = 0;
    for (uint32_t i = lane; i < 320; i += 64) ws->lens[i] = 0;
    fl_wave_lds_sync();
    for (uint32_t i = 0; i < hclen; i++) {
        FL_TRY(fl_br_read(r, 3, v));
        if (lane == 0) ws->cl_lens[fl_codegen_order(i)] = (uint8_t)v;
    }
    fl_wave_lds_sync();
    FL_TRY(fl_hdec_generate(&ws->cl, ws->cl_lens, ws->offs, 19, 19, 7, lane));
    bool crossed = false;
    int rc;
    if (flags & 1) {
        // literal lengths live at lens[0..286), distance lengths at lens[288..318)
        FL_TRY(fl_inf_read_lens(r, ws, 0, 286, hlit, 0, crossed, lane));
        FL_TRY(fl_inf_read_lens(r, ws, 288, 30, hdist, 0, crossed, lane));
        fl_wave_lds_sync();
        FL_TRY(fl_hdec_generate(&ws->lit, ws->lens, ws->offs, 286, 286, 15, lane));
        FL_TRY(fl_hdec_generate(&ws->dst, ws->lens + 288, ws->offs, 30, 30, 15, lane));
        fl_hdec_build_lut(&ws->lit, ws->lit_lut, FL_INF_LIT_BITS, lane);
        fl_hdec_build_lut(&ws->dst, ws->dst_lut, FL_INF_DST_BITS, lane);
        return 0;
    }
    rc = fl_inf_read_lens(r, ws, 0, hlit + hdist, hlit + hdist, hlit, crossed, lane);
    if (rc) return crossed ? 14 : rc;
    fl_wave_lds_sync();
    // split the single list: the literal decoder must see zeros in [hlit, 286)
    uint8_t dl = 0;
    if (lane < 30) dl = lane < hdist ? ws->lens[hlit + lane] : 0;
    fl_wave_lds_sync();
    for (uint32_t i = hlit + lane; i < 320; i += 64) ws->lens[i] = 0;
    fl_wave_lds_sync();
    if (lane < 30) ws->lens[288 + lane] = dl;
    fl_wave_lds_sync();
    rc = fl_hdec_generate(&ws->lit, ws->lens, ws->offs, 286, 286, 15, lane);
    if (rc) return crossed ? 14 : rc;
    rc = fl_hdec_generate(&ws->dst, ws->lens + 288, ws->offs, 30, 30, 15, lane);
    if (rc) return crossed ? 14 : rc;
    fl_hdec_build_lut(&ws->lit, ws->lit_lut, FL_INF_LIT_BITS, lane);
    fl_hdec_build_lut(&ws->dst, ws->dst_lut, FL_INF_DST_BITS, lane);
    return 0;
}

// inflate.zig:220-249.  Codes of up to 10 / 9 bits come out of the LDS tables built after the
// block header; longer ones (and invalid ones) take the canonical walk, which also keeps the
// reference's order of errors: a miss in the table of the decoder is InvalidCode before the
// bits are consumed (huffman_decoder.zig:156-175), running out of input is EndOfStream at the
// shift (bit_reader.zig:159-163).
__device__ __forceinline__ int fl_inf_dynamic(fl_bitr& r, FL_LDS fl_inflate_ws* ws, fl_inf_out& o, uint32_t lane) {
    for (;;) {
        FL_TRY(fl_br_fill(r, 15));
        uint32_t sym, cb;
        {
            const uint32_t pk = fl_br_peek(r, 15);
            const uint32_t e = ws->lit_lut[pk & ((1u << FL_INF_LIT_BITS) - 1)];
            if (e) {
                sym = e & 0x1ff;
                cb = e >> 9;
            } else {
                FL_TRY(fl_hdec_find(&ws->lit, pk, 15, sym, cb));
            }
        }
        FL_TRY(fl_br_shift(r, cb));
        if (sym < 256) {
            FL_TRY(fl_inf_literal(o, sym, lane));
        } else if (sym == 256) {
            return 0;
        } else {
            FL_TRY(fl_br_fill(r, 5 + 15 + 13));
            uint32_t length, distance, dsym;
            FL_TRY(fl_inf_length(r, sym - 257, length));
            {
                const uint32_t pk = fl_br_peek(r, 15);
                const uint32_t e = ws->dst_lut[pk & ((1u << FL_INF_DST_BITS) - 1)];
                if (e) {
                    dsym = e & 0x1ff;
                    cb = e >> 9;
                } else {
                    FL_TRY(fl_hdec_find(&ws->dst, pk, 15, dsym, cb));
                }
            }
            FL_TRY(fl_br_shift(r, cb));
            FL_TRY(fl_inf_distance(r, dsym, distance));
            FL_TRY(fl_inf_match(o, length, distance, lane));
        }
    }
}

// container.zig:119-152
__device__ __forceinline__ int fl_inf_header(fl_bitr& r, int container) {
    uint32_t v;
    if (container == 1) {
        uint32_t m1, m2, method, flags;
        FL_TRY(fl_br_read(r, 8, m1));
        FL_TRY(fl_br_read(r, 8, m2));
        FL_TRY(fl_br_read(r, 8, method));
        FL_TRY(fl_br_read(r, 8, flags));
        for (int i = 0; i < 6; i++) FL_TRY(fl_br_read(r, 8, v));
        if (m1 != 0x1f || m2 != 0x8b || method != 0x08) return 2;
        if (flags & 0x04) {
            uint32_t xl;
            FL_TRY(fl_br_read(r, 16, xl));
            for (uint32_t i = 0; i < xl; i++) FL_TRY(fl_br_read(r, 8, v));
        }
        if (flags & 0x08) {
            do {
                FL_TRY(fl_br_read(r, 8, v));
            } while (v != 0);
        }
        if (flags & 0x10) {
            do {
                FL_TRY(fl_br_read(r, 8, v));
            } while (v != 0);
        }
        if (flags & 0x02) {
            FL_TRY(fl_br_read(r, 8, v));
            FL_TRY(fl_br_read(r, 8, v));
        }
    } else if (container == 2) {
        uint32_t cm, cinfo;
        FL_TRY(fl_br_read(r, 4, cm));
        FL_TRY(fl_br_read(r, 4, cinfo));
        FL_TRY(fl_br_read(r, 8, v));
        if (cm != 8 || cinfo > 7) return 3;
    }
    return 0;
}

// CRC-32 of out[0..n) by the whole wave (container.zig:170, inflate.zig:330)
__device__ __forceinline__ uint32_t fl_wave_crc32(const uint8_t* p, uint64_t n, const fl_crc_consts& cc,
                                                  FL_LDS uint32_t* tab /*256*/, uint32_t lane) {
    for (uint32_t t = lane; t < 256; t += 64) {
        uint32_t c = t;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (FL_CRC_POLY ^ (c >> 1)) : (c >> 1);
        tab[t] = c;
    }
    fl_wave_lds_sync();
    const uint64_t per = (n + 63) / 64;
    const uint64_t lo = min(n, lane * per), hi = min(n, lo + per);
    uint32_t c = 0xffffffffu;
    for (uint64_t i = lo; i < hi; i++) c = tab[(c ^ p[i]) & 0xff] ^ (c >> 8);
    c = hi > lo ? ~c : 0u;
    c = fl_crc_mulmod(c, fl_crc_xpow8n(cc.xpow8, n - hi));
    return fl_wave_xor(c);
}
// Adler-32 of out[0..n) by the whole wave
__device__ __forceinline__ uint32_t fl_wave_adler32(const uint8_t* p, uint64_t n, uint32_t lane) {
    const uint64_t per = (n + 63) / 64;
    const uint64_t lo = min(n, lane * per), hi = min(n, lo + per);
    uint32_t A = 0, B = 0;  // a = b = 0 start
    uint64_t i = lo;
    while (i < hi) {
        const uint64_t e = min(hi, i + 5552);
        for (; i < e; i++) {
            A += p[i];
            B += A;
        }
        A %= 65521u;
        B %= 65521u;
    }
    const uint64_t after = (n - hi) % 65521u;
    uint32_t Bm = (uint32_t)((B + (uint64_t)A * after) % 65521u);
    const uint32_t Am = fl_wave_sum(A) % 65521u;
    Bm = fl_wave_sum(Bm) % 65521u;
    const uint32_t a = (1 + Am) % 65521u;
    const uint32_t b = (uint32_t)((n % 65521u + Bm) % 65521u);
    return a | (b << 16);
}

// One wave per stream.
__global__ __launch_bounds__(64) void k_inflate(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks,
                                                int container, int flags, fl_crc_consts cc,
                                                uint8_t* __restrict__ out, uint64_t* __restrict__ out_len,
                                                int32_t* __restrict__ status, uint64_t* __restrict__ consumed) {
    __shared__ fl_inflate_ws ws_mem;
    __shared__ uint32_t crc_tab_mem[256];
    __shared__ uint32_t inring_mem[FL_INF_INRING / 4];
    FL_LDS fl_inflate_ws* ws = (FL_LDS fl_inflate_ws*)&ws_mem;
    FL_LDS uint32_t* crc_tab = (FL_LDS uint32_t*)crc_tab_mem;
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    const uint32_t lane = threadIdx.x;
    if (ck.skip) return;
    fl_bitr r;
    r.data = in + ck.in_off;
    r.nbytes = ck.in_len;
    r.left = (int64_t)ck.in_len * 8;
    r.lane = lane;
    r.inring = (FL_LDS uint32_t*)inring_mem;
    fl_br_seek(r, 0);
    fl_inf_out o;
    o.out = out + ck.out_off;
    o.ring = ws->ring;
    o.cap = ck.out_cap;
    o.wp = 0;
    o.fenced = 0;

    int rc = fl_inf_header(r, container);
    while (rc == 0) {  // inflate.zig:251-280
        uint32_t bfinal, btype;
        if ((rc = fl_br_read(r, 1, bfinal))) break;
        if ((rc = fl_br_read(r, 2, btype))) break;
        if (btype == 2) {
            if ((rc = fl_inf_dynamic_header(r, ws, flags, lane))) break;
            rc = fl_inf_dynamic(r, ws, o, lane);
        } else if (btype == 0) {
            rc = fl_inf_stored(r, o, lane);
        } else if (btype == 1) {
            rc = fl_inf_fixed(r, o, lane);
        } else {
            rc = 12;  // InvalidBlockType
        }
        if (rc) break;
        if (bfinal) {
            fl_br_align(r);
            // container.zig:154-166
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t v;
            if (container == 1) {
                const uint32_t crc = fl_wave_crc32(o.out, o.wp, cc, crc_tab, lane);
                if ((rc = fl_br_read(r, 32, v))) break;
                if (v != crc) {
                    rc = 4;
                    break;
                }
                if ((rc = fl_br_read(r, 32, v))) break;
                if (v != (uint32_t)o.wp) rc = 5;
            } else if (container == 2) {
                const uint32_t ad = fl_wave_adler32(o.out, o.wp, lane);
                if ((rc = fl_br_read(r, 32, v))) break;
                if (v != __builtin_bswap32(ad)) rc = 6;
            }
            break;
        }
    }
    if (lane == 0) {
        out_len[c] = o.wp;
        status[c] = rc;
        if (consumed) consumed[c] = fl_br_consumed(r);
    }
}
