// kernels_inflate.h -- batched inflate: one wavefront per independent stream.
//
// Reference path: Inflate.step / dynamicBlockHeader / dynamicBlock / fixedBlock /
// storedBlock (inflate.zig:89-280), HuffmanDecoder (huffman_decoder.zig:71-175),
// BitReader (bit_reader.zig:46-217), CircularBuffer.writeMatch (CircularBuffer.zig:44-75),
// container parse (container.zig:111-166).  The status returned for a bad stream is
// the error name the reference returns (pinned by its 40-case table, inflate.zig:487-527).
//
// The symbol decode of one stream is serial, but a dynamic block is decoded in rounds (fl_inf_fast_round): every lane
// decodes the whole token that would start at "current bit + lane", the wave walks the chain of real token starts with
// one scalar read per token, output offsets come from a prefix sum and the tokens that read nothing of the round are
// copied together; anything unusual takes the symbol-at-a-time path, which keeps the reference's order of errors.
// Output goes to an LDS ring and leaves in coalesced 8-byte stores; the checksum is spread over the 64 lanes.
// Throughput comes from many streams in flight: 20 per CU with the 2 KiB ring (6.5 KB of LDS per stream).
#pragma once
#include "kernels_common.h"

// canonical decoder tables (huffman_decoder.zig:71-153); NSYM = alphabet capacity
template <int NSYM>
struct fl_hdec_t {
    uint16_t count[16];
    uint16_t symbol[NSYM];
};
typedef fl_hdec_t<288> fl_hdec;      // literal / length alphabet (286 used)
typedef fl_hdec_t<32> fl_hdec_small;  // distance (30) and code-length (19) alphabets

// LDS pointers are typed with their address space: with generic pointers the decoder's table
// and ring accesses became FLAT instructions (PMC: 323 k FLAT vs 30 k LDS per wave), i.e. LDS
// traffic through the vector-memory pipe.
#define FL_LDS __attribute__((address_space(3)))

#define FL_INF_LIT_BITS 10
#define FL_INF_DST_BITS 8
#ifndef FL_INF16_DST_BITS
#define FL_INF16_DST_BITS 9  // k_inflate's own distance table (fl_inflate_ws16): 7 KB of LDS per stream still hold 20 per CU; 8 bits 17.9 ms, 9: 17.6, 10: 19.1
#endif
// Recent output kept in LDS (power of two); matches up to ring - 260 back are served from it.
// Large batches run the small ring (20 streams per CU in flight); small batches, where the
// latency of one stream is what counts, the large one: every match is then an LDS copy.
#define FL_INF_RING_SMALL 2048u
#define FL_INF_RING_LARGE 32768u

struct fl_inflate_ws {
    enum { DST_BITS = FL_INF_DST_BITS };
    fl_hdec lit;
    fl_hdec_small dst, cl;
    // symbol | code_bits << 9 | extra_bits << 13 | value << 17, 0 = not in the table.  value: the
    // byte of a literal, base length of a length code (inflate.zig:123-131), base distance of a
    // distance code (:133-140); extra_bits = 15 marks a symbol that is not a valid code
    uint32_t lit_lut[1u << FL_INF_LIT_BITS];
    uint32_t dst_lut[1u << FL_INF_DST_BITS];
    uint8_t lens[320];
    uint8_t cl_lens[20];
    uint16_t offs[18];
    uint16_t cl_lut[128];  // code-length code: symbol | code bits << 8 | 0x8000 by the next 7 stream bits, 0 = no code
};

// k_inflate's own, smaller workspace: what a stream keeps in LDS decides how many streams a CU decodes at once, and
// the kernel's time is inversely proportional to that number (16 -> 12 streams per CU: 20.1 -> 26.8 ms, r04).  Table
// entries of 16 bits -- the base values follow from symbol and extra-bit count in a few VALU instructions -- and
// the structures that only the block header needs lie where the literal table is built afterwards.
//   lit_lut: symbol | code_bits << 9 | extra_bits << 13 (7 = not a valid length code), 0 = not in the table
//   dst_lut: symbol | code_bits << 5 | extra_bits << 9 (15 = not a valid distance code), 0 = not in the table
struct fl_inflate_ws16 {
    enum { DST_BITS = FL_INF16_DST_BITS };
    fl_hdec lit;
    fl_hdec_small dst;
    union {
        uint16_t lit_lut[1u << FL_INF_LIT_BITS];
        struct {  // dead once the two decoders are generated, before the tables are filled
            fl_hdec_small cl;
            uint8_t cl_lens[20];
            uint16_t offs[18];
            uint16_t cl_lut[128];
        };
    };
    uint16_t dst_lut[1u << FL_INF16_DST_BITS];
    uint8_t lens[320];
};
// table entry of a symbol with a code of cb bits, by table type
template <bool DIST>
__device__ __forceinline__ uint32_t fl_lut_entry(uint32_t sym, uint32_t cb, uint32_t /*tag*/) {
    uint32_t eb, val;
    if (DIST) {
        eb = sym <= 29 ? fl_dist_extra_bits(sym) : 15u;
        val = sym <= 29 ? fl_dist_base_scaled(sym) + 1 : 0u;
    } else if (sym < 256) {
        eb = 0;
        val = sym;
    } else if (sym == 256) {
        eb = 0;
        val = 0;
    } else {
        eb = sym <= 285 ? fl_len_extra_bits(sym - 257) : 15u;
        val = sym <= 285 ? fl_len_base_scaled(sym - 257) + 3 : 0u;
    }
    return sym | (cb << 9) | (eb << 13) | (val << 17);
}
template <bool DIST>
__device__ __forceinline__ uint16_t fl_lut_entry(uint32_t sym, uint32_t cb, uint16_t /*tag*/) {
    if (DIST) return (uint16_t)(sym | (cb << 5) | ((sym <= 29 ? fl_dist_extra_bits(sym) : 15u) << 9));
    const uint32_t eb = sym <= 256 ? 0u : sym <= 285 ? fl_len_extra_bits(sym - 257) : 7u;
    return (uint16_t)(sym | (cb << 9) | (eb << 13));
}
// symbol and code bits of a distance table entry (the slow path wants nothing else)
__device__ __forceinline__ void fl_dst_sym_cb(uint32_t e, uint32_t& sym, uint32_t& cb) {
    sym = e & 0x1ff;
    cb = (e >> 9) & 15;
}
__device__ __forceinline__ void fl_dst_sym_cb(uint16_t e, uint32_t& sym, uint32_t& cb) {
    sym = e & 31u;
    cb = (e >> 5) & 15;
}

#define FL_INF_INRING 1024u  // compressed bytes staged in LDS (two 512-byte halves)

// wave-uniform values the compiler cannot prove uniform (anything derived from an LDS load)
__device__ __forceinline__ uint32_t fl_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t fl_uni64(uint64_t v) {
    return (uint64_t)fl_uni((uint32_t)v) | ((uint64_t)fl_uni((uint32_t)(v >> 32)) << 32);
}

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
        // everything the reader's state is computed from is pinned to scalar registers: what comes out of LDS is
        // divergent as far as the compiler knows, and the stream position would live in vector registers with it
        const uint32_t w = fl_uni(__builtin_amdgcn_alignbyte(hi, lo, r.next_byte & 3));
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

// huffman_decoder.zig:71-153 (checkCompletnes + canonical symbol order), spread over the wave: lane l
// holds symbols l, l + 64, ...; per code length one ballot per 64 symbols counts the codes and ranks
// the symbols (a symbol's place among the codes of its length = the symbols of that length before it).
template <class H>
__device__ __forceinline__ int fl_hdec_generate(FL_LDS H* d, const FL_LDS uint8_t* lens, FL_LDS uint16_t* offs,
                                                int n, int alphabet, int max_code_bits, uint32_t lane) {
    if (alphabet == 286 && lens[256] == 0) return 10;  // MissingEndOfBlockCode
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t cnt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cnt[i] = 0;
    uint32_t ml[5], mr[5];  // this lane's symbols: length, rank among the symbols of that length
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const int i = c * 64 + (int)lane;
        ml[c] = (c * 64 < n && i < n) ? lens[i] : 0u;
        mr[c] = 0;
        if (c * 64 < n) {
#pragma unroll
            for (int k = 1; k < 16; k++) {
                const uint64_t m = __ballot(ml[c] == (uint32_t)k);
                if (ml[c] == (uint32_t)k) mr[c] = cnt[k] + (uint32_t)__popcll(m & lt_mask);
                cnt[k] += (uint32_t)__popcll(m);
            }
        }
    }
    int mx = 0;
#pragma unroll
    for (int k = 1; k < 16; k++)
        if (cnt[k]) mx = k;
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
        uint32_t run = 0;
#pragma unroll
        for (int len = 1; len < 16; len++) {
            d->count[len] = (uint16_t)cnt[len];
            offs[len] = (uint16_t)run;
            run += cnt[len];
        }
    }
    fl_wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 5; c++)
        if (ml[c] != 0) d->symbol[offs[ml[c]] + mr[c]] = (uint16_t)(c * 64 + (int)lane);
    fl_wave_lds_sync();
    return 0;
}

// huffman_decoder.zig:156-175: the symbol whose code is a prefix of `peek`
// (stream bit order), or InvalidCode.
template <class H>
__device__ __forceinline__ int fl_hdec_find(const FL_LDS H* d, uint32_t peek, int max_code_bits, uint32_t& sym,
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
template <bool DIST, class H, class E>
__device__ __forceinline__ void fl_hdec_build_lut(const FL_LDS H* d, FL_LDS E* lut, int bits, uint32_t lane) {
    // Every symbol with a code of at most `bits` bits writes the entries whose low bits are its code (first bit of the
    // code = lowest bit of the index): a lane per symbol in (length, symbol) order, which is the order of the codes
    // (huffman_decoder.zig:62-117).  What no code covers stays 0 (a longer code, or none: InvalidCode).
    uint32_t cnt[16];
    uint32_t upto = 0;  // symbols with a code of at most `bits` bits
#pragma unroll
    for (int len = 1; len < 16; len++) {
        cnt[len] = d->count[len];
        if (len <= bits) upto += cnt[len];
    }
    fl_wave_lds_sync();  // (the 16-bit literal table lies over structures of the block header: counts first)
    for (uint32_t i = lane; i < (1u << bits); i += 64) lut[i] = 0;
    fl_wave_lds_sync();
    for (uint32_t j = lane; j < upto; j += 64) {
        uint32_t code = 0, idx = 0, my_len = 0, my_code = 0;
#pragma unroll
        for (int len = 1; len < 16; len++) {
            if (j >= idx && j < idx + cnt[len]) {
                my_len = (uint32_t)len;
                my_code = code + (j - idx);
            }
            code = (code + cnt[len]) << 1;
            idx += cnt[len];
        }
        const E e = fl_lut_entry<DIST>((uint32_t)d->symbol[j], my_len, E());
        const uint32_t rev = __brev(my_code) >> (32 - my_len);
        for (uint32_t t = rev; t < (1u << bits); t += 1u << my_len) lut[t] = e;
    }
    fl_wave_lds_sync();
}

__device__ __forceinline__ uint32_t fl_rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32 - n); }

// Output of one stream.  Decoded bytes go to an LDS ring only; whenever 1 KiB has piled up the
// wave stores the finished 512-byte lines to the caller's buffer with 8-byte stores (one byte
// store per literal made the memory pipe, not the decoder, the limit).  Output byte k lives in
// ring[(k + bias) & (RING - 1)], bias = low 3 bits of the output address, so that an 8-byte
// aligned global address is an 8-byte aligned ring index.
struct fl_inf_out {
    uint8_t* out;
    FL_LDS uint8_t* ring;
    uint64_t cap;
    uint64_t wp;
    uint64_t flushed;  // output bytes [0, flushed) have been stored to `out`
    uint64_t fenced;   // ... and [0, fenced) are known to have landed: loads of this wave see them
    uint32_t bias;
    uint32_t rmask;     // ring size - 1
    uint32_t near_max;  // ring size - 260
};

// store output bytes [flushed, upto) from the ring
__device__ __forceinline__ void fl_inf_flush(fl_inf_out& o, uint64_t upto, uint32_t lane) {
    const uint64_t a = o.flushed;
    if (upto <= a) return;
    fl_lds_order();
    const uint64_t va = a + o.bias, vb = upto + o.bias;  // same low bits as the global addresses
    const uint64_t w0 = (va + 7) >> 3, w1 = vb >> 3;     // whole 8-byte words [w0, w1)
    uint8_t* base = o.out - o.bias;                      // 8-byte aligned
    if (w0 <= w1) {
        const uint32_t nh = (uint32_t)(w0 * 8 - va), nt = (uint32_t)(vb - w1 * 8);
        if (lane < nh) base[va + lane] = o.ring[(uint32_t)(va + lane) & o.rmask];
        for (uint64_t w = w0 + lane; w < w1; w += 64) {
            const FL_LDS uint32_t* rw = (const FL_LDS uint32_t*)(o.ring + ((uint32_t)(w * 8) & o.rmask));
            *(uint2*)(base + w * 8) = make_uint2(rw[0], rw[1]);
        }
        if (lane < nt) base[w1 * 8 + lane] = o.ring[(uint32_t)(w1 * 8 + lane) & o.rmask];
    } else if (lane < (uint32_t)(vb - va)) {  // inside one word
        base[va + lane] = o.ring[(uint32_t)(va + lane) & o.rmask];
    }
    fl_lds_order();
    o.flushed = upto;
}
// Called when wp has moved: keeps less than FL_INF_PILE + 258 bytes unflushed (the ring holds
// at least 2048 and near matches reach ring - 260 back: nothing live is ever overwritten).
#define FL_INF_PILE 1024u
__device__ __forceinline__ void fl_inf_advance(fl_inf_out& o, uint32_t lane) {
    if ((uint32_t)(o.wp - o.flushed) >= FL_INF_PILE)
        fl_inf_flush(o, ((o.wp + o.bias) & ~(uint64_t)511) - o.bias, lane);
}

// CircularBuffer.zig:44-75, spread over the wave.  Near matches copy inside the LDS ring; far
// ones read the output buffer: that far back everything has been flushed (see fl_inf_advance),
// the wave only has to wait for those stores if they are younger than the last wait.
__device__ __forceinline__ int fl_inf_match(fl_inf_out& o, uint32_t length, uint32_t distance, uint32_t lane) {
    if (o.wp < distance || length < 3 || length > 258 || distance < 1 || distance > 32768) return 11;
    if (o.wp + length > o.cap) return 100;
    const uint32_t vp = (uint32_t)o.wp + o.bias;  // ring positions only need the low bits
    if (distance <= o.near_max) {
        for (uint32_t i0 = 0; i0 < length; i0 += 64) {  // one trip for lengths up to 64
            const uint32_t i = i0 + lane;
            uint32_t byte = 0;
            if (i < length) {
                // the source bytes repeat with period `distance` when the match overlaps itself
                const uint32_t si = distance >= length ? i : (i % distance);
                byte = o.ring[(vp - distance + si) & o.rmask];
            }
            fl_lds_order();
            if (i < length) o.ring[(vp + i) & o.rmask] = (uint8_t)byte;
            fl_lds_order();
        }
    } else {
        if (o.wp - distance + length > o.fenced) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            o.fenced = o.flushed;
        }
        const uint8_t* from = o.out + o.wp - distance;  // distance > length: no overlap
        for (uint32_t i0 = 0; i0 < length; i0 += 64) {
            const uint32_t i = i0 + lane;
            if (i < length) o.ring[(vp + i) & o.rmask] = from[i];
        }
        fl_lds_order();
    }
    o.wp += length;
    fl_inf_advance(o, lane);
    return 0;
}
__device__ __forceinline__ int fl_inf_literal(fl_inf_out& o, uint32_t byte, uint32_t lane) {
    if (o.wp >= o.cap) return 100;
    // every lane stores the same byte to the same LDS address
    o.ring[((uint32_t)o.wp + o.bias) & o.rmask] = (uint8_t)byte;
    o.wp++;
    fl_inf_advance(o, lane);
    return 0;
}

// (the status is pinned to a scalar register: a status that depends on something read from LDS is divergent as far as
// the compiler knows, and every loop that ends on it would keep the stream position in vector registers)
#define FL_TRY(expr)                               \
    do {                                           \
        const int rc_ = (int)fl_uni((uint32_t)(expr)); \
        if (rc_) return rc_;                       \
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
    fl_inf_flush(o, o.wp, lane);  // what the ring still holds goes out first
    for (uint32_t i = lane; i < len; i += 64) o.out[o.wp + i] = s[i];
    // the ring mirrors the last bytes of the output
    {
        const uint32_t tail = len < o.rmask + 1 ? len : o.rmask + 1;
        fl_lds_order();
        for (uint32_t i = lane; i < tail; i += 64) {
            const uint64_t off = o.wp + len - tail + i;
            o.ring[((uint32_t)off + o.bias) & o.rmask] = s[len - tail + i];
        }
        fl_lds_order();
    }
    o.wp += len;
    o.flushed = o.wp;
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
template <class WS>
__device__ __forceinline__ int fl_inf_read_lens(fl_bitr& r, FL_LDS WS* ws, uint32_t base, uint32_t lens_len, uint32_t want,
                                uint32_t boundary, bool& crossed, uint32_t lane) {
    uint32_t pos = 0;
    FL_LDS uint8_t* lens = ws->lens + base;
    while (pos < want) {
        FL_TRY(fl_br_fill(r, 7));
        uint32_t sym, cb;
        {
            const uint32_t e = fl_uni(ws->cl_lut[fl_br_peek(r, 7)]);
            if (e == 0) return 7;  // InvalidCode (huffman_decoder.zig:156-175)
            sym = e & 0xff;
            cb = (e >> 8) & 15;
        }
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
#ifdef FL_PAR_PROF
#define FL_HDR_T(slot) do { const uint64_t n_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += n_ - th_; th_ = n_; } while (0)
#else
#define FL_HDR_T(slot)
#endif
template <class WS>
__device__ __forceinline__ int fl_inf_dynamic_header(fl_bitr& r, FL_LDS WS* ws, int flags, uint32_t lane) {
    uint32_t v;
#ifdef FL_PAR_PROF
    uint64_t th_ = __builtin_readcyclecounter();
#endif
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
    FL_HDR_T(20);
    FL_TRY(fl_hdec_generate(&ws->cl, ws->cl_lens, ws->offs, 19, 19, 7, lane));
    for (uint32_t i = lane; i < 128; i += 64) {  // every 7-bit window decoded once, by the same walk
        uint32_t sy, cb;
        ws->cl_lut[i] = fl_hdec_find(&ws->cl, i, 7, sy, cb) == 0 ? (uint16_t)(sy | (cb << 8) | 0x8000u) : (uint16_t)0;
    }
    fl_wave_lds_sync();
    FL_HDR_T(21);
    bool crossed = false;
    int rc;
    if (flags & 1) {
        // literal lengths live at lens[0..286), distance lengths at lens[288..318)
        FL_TRY(fl_inf_read_lens(r, ws, 0, 286, hlit, 0, crossed, lane));
        FL_TRY(fl_inf_read_lens(r, ws, 288, 30, hdist, 0, crossed, lane));
        fl_wave_lds_sync();
        FL_TRY(fl_hdec_generate(&ws->lit, ws->lens, ws->offs, 286, 286, 15, lane));
        FL_TRY(fl_hdec_generate(&ws->dst, ws->lens + 288, ws->offs, 30, 30, 15, lane));
        fl_hdec_build_lut<false>(&ws->lit, ws->lit_lut, FL_INF_LIT_BITS, lane);
        fl_hdec_build_lut<true>(&ws->dst, ws->dst_lut, WS::DST_BITS, lane);
        return 0;
    }
    rc = (int)fl_uni((uint32_t)fl_inf_read_lens(r, ws, 0, hlit + hdist, hlit + hdist, hlit, crossed, lane));
    if (rc) return crossed ? 14 : rc;
    fl_wave_lds_sync();
    FL_HDR_T(22);
    // split the single list: the literal decoder must see zeros in [hlit, 286)
    uint8_t dl = 0;
    if (lane < 30) dl = lane < hdist ? ws->lens[hlit + lane] : 0;
    fl_wave_lds_sync();
    for (uint32_t i = hlit + lane; i < 320; i += 64) ws->lens[i] = 0;
    fl_wave_lds_sync();
    if (lane < 30) ws->lens[288 + lane] = dl;
    fl_wave_lds_sync();
    rc = (int)fl_uni((uint32_t)fl_hdec_generate(&ws->lit, ws->lens, ws->offs, 286, 286, 15, lane));
    if (rc) return crossed ? 14 : rc;
    rc = (int)fl_uni((uint32_t)fl_hdec_generate(&ws->dst, ws->lens + 288, ws->offs, 30, 30, 15, lane));
    if (rc) return crossed ? 14 : rc;
    FL_HDR_T(23);
    fl_hdec_build_lut<false>(&ws->lit, ws->lit_lut, FL_INF_LIT_BITS, lane);
    fl_hdec_build_lut<true>(&ws->dst, ws->dst_lut, WS::DST_BITS, lane);
    FL_HDR_T(24);
    return 0;
}

// Restart the symbol-at-a-time reader at the stream position that `left` says we are at.
__device__ __forceinline__ void fl_br_resync(fl_bitr& r) {
    const uint64_t pos = (uint64_t)r.nbytes * 8 - (uint64_t)r.left;
    r.buf = 0;
    r.have = 0;
    r.next_byte = (uint32_t)(pos >> 3);
    const uint32_t k = (uint32_t)pos & 7;
    if (k) {
        fl_br_refill(r);
        r.buf >>= k;
        r.have -= k;
    }
}

// One fast round of a dynamic block: every token that starts in the next 64 bits of the stream.
//
// (1) Every lane decodes the WHOLE token that would start at "current bit + lane" from the LDS tables (literal / length
//     code + extra bits, and for a length the distance code + extra bits behind it: a 64-bit window per lane), giving
//     its size in bits `nb` and in output bytes `olen`.
// (2) The only serial step left is the chain of real token starts, p -> p + nb[p]: one v_readlane, one s_bitset and one
//     s_add per token (the PMC counters of round 3's walk said 51 scalar instructions per token, with the CU's scalar
//     issue saturated: SQ_INSTS_SALU + BRANCH = 0.93 per cycle and CU).  Its result is the mask S of lanes that are tokens.
// (3) Output offsets by a prefix sum over S; the checks that need them (room in the slot, distance against the history)
//     cut S in front of the first token that fails: that one is left to the symbol-at-a-time path, which owns the
//     reference's error order -- as is a code longer than the tables, an invalid symbol, the end of input in sight.
// (4) Tokens whose source lies wholly before this round's output (nearly all matches of text) are copied TOGETHER:
//     output byte b of the round belongs to lane b, which finds its token (owner lanes through a small LDS array and a
//     max-scan), takes the literal or reads ring / output buffer at b - distance and writes the ring.  From the first
//     token that reads this round's own output (or from 260 bytes on: the ring must not be lapped) the rest goes one
//     match at a time, as before.
// Returns 0 = go on, 1 = end of block, 2 = the next symbol needs the slow path.
#ifdef FL_INF_COUNT
#define FL_T0() const uint64_t t0_ = __builtin_readcyclecounter()
#define FL_TACC(slot)                                                                          \
    do {                                                                                       \
        if (blockIdx.x == 0 && lane == 0) g_fl_prof[slot] += __builtin_readcyclecounter() - t0_; \
    } while (0)
#else
#define FL_T0()
#define FL_TACC(slot)
#endif
#define FL_INF_FAST_MIN_BITS 192  // a round looks at 63 + 64 bits and consumes at most 63 + 36
#define FL_INF_PAR_MAX 260u       // bytes of one round that are copied together (ring - near_max: nothing live is lapped)
__device__ __forceinline__ uint32_t fl_wave_incl_max_dpp(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return v;
}
#define FL_BALLOT(c) __builtin_amdgcn_ballot_w64(c)
// A lane predicate from a mask held in scalar registers: no vector instruction at all (the mask is the select operand /
// the exec mask).  The round's predicates are kept as masks and combined on the scalar side: written as lane booleans
// the compiler turned every combination into a 0/1 vector register and compared it again.
#define FL_INV(m) __builtin_amdgcn_inverse_ballot_w64(m)
__device__ __forceinline__ int fl_inf_fast_round(fl_bitr& r, FL_LDS fl_inflate_ws16* ws, fl_inf_out& o, uint32_t lane) {
    FL_T0();
    const uint64_t left0 = fl_uni64((uint64_t)r.left);
    const uint64_t pos = (uint64_t)fl_uni(r.nbytes) * 8 - left0;
    const uint32_t byte0 = (uint32_t)(pos >> 3);
    if (__builtin_expect(byte0 + 24 > fl_uni(r.in_loaded), 0)) fl_br_commit_half(r);
    // ---- (1) the token that starts at pos + lane ----
    const uint32_t bp = ((byte0 & (FL_INF_INRING - 1)) << 3) + ((uint32_t)pos & 7) + lane;  // bit index in the ring
    const uint32_t di = bp >> 5;
    const uint32_t IM = FL_INF_INRING / 4 - 1;
    const uint32_t d0 = r.inring[di & IM], d1 = r.inring[(di + 1) & IM], d2 = r.inring[(di + 2) & IM];
    const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, bp & 31);  // stream bits [pos + lane, + 32)
    const uint32_t w1 = __builtin_amdgcn_alignbit(d2, d1, bp & 31);  // ... [+ 32, + 64)
    const uint32_t le = ws->lit_lut[w0 & ((1u << FL_INF_LIT_BITS) - 1)];
    const uint32_t lsym = le & 511, lcb = (le >> 9) & 15, leb = le >> 13;
    const uint64_t m_lok = FL_BALLOT(le != 0) & FL_BALLOT(leb != 7);
    const uint64_t m_lit = m_lok & FL_BALLOT(lsym < 256);
    const uint32_t lbits = lcb + leb;  // at most 10 + 5
    const uint32_t wd = __builtin_amdgcn_alignbit(w1, w0, lbits & 31);  // the 32 bits behind the length code
    const uint32_t de = ws->dst_lut[wd & ((1u << FL_INF16_DST_BITS) - 1)];
    const uint32_t dsym = de & 31, dcb = (de >> 5) & 15, deb = de >> 9;
    const uint64_t m_match = m_lok & FL_BALLOT(lsym > 256) & FL_BALLOT(de != 0) & FL_BALLOT(deb != 15);
    const uint64_t m_plain = m_lit | m_match;
    const bool is_lit = FL_INV(m_lit), is_match = FL_INV(m_match);
    // base values (inflate.zig:123-140) from the code and its extra-bit count
    const uint32_t lc = lsym - 257;
    // (selects, not branches: written with ?: inside ?: the compiler masked lanes in and out around three instructions)
    uint32_t lbase = ((4u | (lc & 3u)) << leb) + 3u;
    lbase = FL_INV(FL_BALLOT(leb != 0)) ? lbase : lc + 3u;
    lbase = FL_INV(FL_BALLOT(lc == 28u)) ? 258u : lbase;
    uint32_t dbase = ((2u | (dsym & 1u)) << deb) + 1u;
    dbase = FL_INV(FL_BALLOT(deb != 0)) ? dbase : dsym + 1u;
    const uint32_t length = lbase + ((w0 >> lcb) & ((1u << leb) - 1));
    const uint32_t dist = dbase + ((wd >> dcb) & ((1u << deb) - 1));
    // anything that is not a plain literal or match ends the chain: it is the last member of S
    const uint32_t nb = is_lit ? lcb : is_match ? lbits + dcb + deb : 64u;
    const uint32_t olen = is_lit ? 1u : is_match ? length : 0u;
    // The rest is wave-uniform; the compiler cannot see that for anything that came out of LDS, so the state is
    // pinned to scalar registers explicitly.
    const uint64_t wp0 = fl_uni64(o.wp);
    const uint64_t cap = fl_uni64(o.cap);
    // (32-bit halves: there is no 64-bit scalar compare, and the vector one costs a constant pair and the compare each time)
    const uint64_t roomq = cap - wp0;
    const uint32_t room0 = (uint32_t)(roomq >> 32) ? 0x40000000u : min((uint32_t)roomq, 0x40000000u);  // output bytes left (saturated)
    const uint32_t hist0 = (uint32_t)(wp0 >> 32) ? 0x100000u : min((uint32_t)wp0, 0x100000u);  // bytes a match may reach back (saturated)
    int32_t unfl0 = (int32_t)fl_uni((uint32_t)(wp0 - o.flushed));                // unflushed bytes = unfl0 + adv
    const uint32_t bias = fl_uni(o.bias), rmask = fl_uni(o.rmask), near_max = fl_uni(o.near_max);
    const uint32_t vp0 = (uint32_t)wp0 + bias;  // ring position of output byte wp0 (low bits)
#ifdef FL_INF_COUNT  // tuning build only (tools/inflate_probe.py): cycles of the table lookups vs the rest
    if (__builtin_amdgcn_readlane((int)nb, 0) == 0x7fffffff) return 5;  // wait for the lookups
    FL_TACC(44);
    const uint64_t t1_ = __builtin_readcyclecounter();
#endif
    // ---- (2) the chain of token starts ----
    // Four instructions per token, by hand (the compiler's loop has six: shift, or, readlane, add, compare, branch; with the
    // scalar pipe at 0.83 of its issue rate 17.5 -> 16.7 ms): the position is kept as p - 64 (mod 2^32), whose low six
    // bits -- all that s_bitset1 and the lane select of v_readlane look at -- are those of p, and whose sum with the
    // token's bits carries exactly when the next position is 64 or more.
    uint64_t S = 0;
    uint32_t p = 0u - 64u, pn;
    asm volatile(
        "1:\n\t"
        "s_bitset1_b64 %[S], %[p]\n\t"
        "v_readlane_b32 %[n], %[nb], %[p]\n\t"
        "s_add_u32 %[p], %[p], %[n]\n\t"
        "s_cbranch_scc0 1b\n\t"
        : [S] "+s"(S), [p] "+s"(p), [n] "=&s"(pn)
        : [nb] "v"(nb)
        : "scc");
    p += 64u;
    int rc = 0;
    uint32_t consumed = p;
    {
        const uint32_t top = 63u - (uint32_t)__builtin_clzll(S);
        if (__builtin_expect(!((m_plain >> top) & 1), 0)) {
            if (((m_lok & FL_BALLOT(lsym == 256)) >> top) & 1) {
                consumed = top + (uint32_t)__builtin_amdgcn_readlane((int)lcb, (int)top);
                rc = 1;
            } else {
                consumed = top;
                rc = 2;
            }
        }
    }
    // ---- (3) where every token's bytes go; the first token that does not fit or reaches too far back ends the round ----
    const uint32_t mylen = FL_INV(S) ? olen : 0u;
    const uint32_t incl = fl_wave_incl_scan_dpp(mylen);
    const uint32_t off = incl - mylen;
    uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (__builtin_expect(room0 < 64u * 258u || hist0 < 32768u, 0)) {  // (else everything fits and every distance has its history)
        const uint64_t fm = S & m_plain & (FL_BALLOT(off + mylen > room0) | (m_match & FL_BALLOT(dist > hist0 + off)));
        if (fm) {
            const uint32_t f = (uint32_t)__builtin_ctzll(fm);
            S &= (1ull << f) - 1;
            consumed = f;
            rc = 2;
            T = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)f);
        }
    }
    const uint64_t m_in = S & m_plain;  // the tokens of this round that write something
    // ---- (4a) the tokens that read nothing of this round, together ----
    const uint32_t oend = off + olen;
    const uint64_t qm = m_in & ((m_match & FL_BALLOT(dist < oend)) | FL_BALLOT(oend > FL_INF_PAR_MAX));
    const uint32_t q0 = qm ? (uint32_t)__builtin_ctzll(qm) : 64u;
    const uint32_t TP = qm ? (uint32_t)__builtin_amdgcn_readlane((int)off, (int)q0) : T;
#ifdef FL_INF_COUNT
    const uint64_t tm_ = __builtin_readcyclecounter();
#endif
    if (TP) {
        FL_LDS uint8_t* own = ws->lens;  // free between two block headers, all zero between two rounds
        if (FL_INV(qm ? m_in & ((1ull << q0) - 1) : m_in)) own[off] = (uint8_t)(lane + 1);
        fl_lds_order();
        const uint32_t tokinfo = is_lit ? ((lsym << 1) | 1u) : (dist << 1);
        uint32_t carry = 0;
        // bytes between the last fence and this round's output (saturated): what a far copy of pass b0 may read ends
        // at most b0 + 64 - near_max bytes behind wp0
        uint64_t gapq = wp0 - fl_uni64(o.fenced);
        uint32_t fgap = (uint32_t)(gapq >> 32) ? 0xfffffff0u - 512u : min((uint32_t)gapq, 0xfffffff0u - 512u);
        const uint8_t* far_base = o.out + ((int64_t)wp0 - 32768);  // (pointer arithmetic: stays a global address)
        for (uint32_t b0 = 0; b0 < TP; b0 += 64) {
            const uint32_t b = b0 + lane;
            const bool live = b < TP;
            uint32_t x = live ? (uint32_t)own[b] : 0u;
            if (live) own[b] = 0;
            x = max(fl_wave_incl_max_dpp(x), carry);  // owners come in rising order: the last one at or before b
            carry = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
            const uint32_t info = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((x - 1) << 2), (int)tokinfo);
            const uint32_t v = info >> 1;
            const uint64_t m_live = FL_BALLOT(live);
            const uint64_t m_copy = m_live & FL_BALLOT((info & 1) == 0);
            const uint64_t m_far = m_copy & FL_BALLOT(v > near_max);
            uint32_t byte = v;
            if (m_far) {
                // every far source of this pass ends at or before wp0 + b0 + 63 - near_max
                if (__builtin_expect(fgap + b0 + 64 > near_max, 0)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    o.fenced = o.flushed;
                    gapq = wp0 - fl_uni64(o.fenced);
                    fgap = (uint32_t)(gapq >> 32) ? 0xfffffff0u - 512u : min((uint32_t)gapq, 0xfffffff0u - 512u);
                }
                // a far match reaches back 32768 at most: uniform base, unsigned 32-bit lane offset
                if (FL_INV(m_far)) byte = far_base[b + 32768u - v];
            }
            if (FL_INV(m_copy & ~m_far)) byte = o.ring[(vp0 + b - v) & rmask];
            fl_lds_order();
            if (FL_INV(m_live)) o.ring[(vp0 + b) & rmask] = (uint8_t)byte;
            fl_lds_order();
        }
    }
    uint32_t adv = TP;  // output bytes produced in this round so far
    if (__builtin_expect(unfl0 + (int32_t)adv >= (int32_t)FL_INF_PILE, 0)) {
        o.wp = wp0 + adv;
        fl_inf_flush(o, ((o.wp + bias) & ~(uint64_t)511) - bias, lane);
        unfl0 = (int32_t)fl_uni((uint32_t)(o.wp - o.flushed)) - (int32_t)adv;
    }
#ifdef FL_INF_COUNT
    if (blockIdx.x == 0 && lane == 0) g_fl_prof[46] += __builtin_readcyclecounter() - tm_;
#endif
    // ---- (4b) from the first token that reads this round's output: one match at a time, literals by their lanes ----
    if (__builtin_expect(qm != 0, 0)) {
        const uint64_t rest = S & ~((1ull << q0) - 1);
        uint64_t lq = m_in & m_lit & rest;
        uint64_t mq = m_in & m_match & rest;
        while (mq) {
            const uint32_t m = (uint32_t)__builtin_ctzll(mq);
            mq &= mq - 1;
            const uint64_t lb = lq & ((1ull << m) - 1);
            if (lb) {
                if (FL_INV(lb)) o.ring[(vp0 + off) & rmask] = (uint8_t)lsym;
                lq &= ~lb;
                fl_lds_order();
            }
            const uint32_t length_m = (uint32_t)__builtin_amdgcn_readlane((int)olen, (int)m);
            const uint32_t distance = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
            const uint32_t moff = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)m);
            const uint32_t vp = vp0 + moff;
#ifdef FL_INF_COUNT
            if (blockIdx.x == 0 && lane == 0) g_fl_prof[47]++;
#endif
            if (distance <= near_max) {
                // A match that overlaps itself repeats its first `distance` bytes: every pass
                // copies as much as is already there (no per-lane modulo), so the period doubles.
                uint32_t have = distance, done = 0;
                do {
                    const uint32_t chunk = min(have, length_m - done);
                    const uint32_t src0 = vp + done - have, dst0 = vp + done;
                    for (uint32_t i0 = 0; i0 < chunk; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        uint32_t byte = 0;
                        if (i < chunk) byte = o.ring[(src0 + i) & rmask];
                        fl_lds_order();
                        if (i < chunk) o.ring[(dst0 + i) & rmask] = (uint8_t)byte;
                        fl_lds_order();
                    }
                    done += chunk;
                    have += chunk;
                } while (done < length_m);
            } else {
                const uint64_t wp = wp0 + moff;
                if (wp - distance + length_m > fl_uni64(o.fenced)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    o.fenced = o.flushed;
                }
                const uint8_t* from = o.out + wp - distance;
                for (uint32_t i0 = 0; i0 < length_m; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    if (i < length_m) o.ring[(vp + i) & rmask] = from[i];
                }
                fl_lds_order();
            }
            adv = moff + length_m;
            if (unfl0 + (int32_t)adv >= (int32_t)FL_INF_PILE) {
                o.wp = wp0 + adv;
                fl_inf_flush(o, ((o.wp + bias) & ~(uint64_t)511) - bias, lane);
                unfl0 = (int32_t)fl_uni((uint32_t)(o.wp - o.flushed)) - (int32_t)adv;
            }
        }
        if (lq) {
            if (FL_INV(lq)) o.ring[(vp0 + off) & rmask] = (uint8_t)lsym;
            fl_lds_order();
        }
    }
#ifdef FL_INF_COUNT
    if (blockIdx.x == 0 && lane == 0) g_fl_prof[45] += __builtin_readcyclecounter() - t1_;
#endif
    o.wp = wp0 + T;
    if (__builtin_expect(unfl0 + (int32_t)T >= (int32_t)FL_INF_PILE, 0)) fl_inf_flush(o, ((o.wp + bias) & ~(uint64_t)511) - bias, lane);
    r.left = (int64_t)(left0 - consumed);
    return rc;
}

// one symbol, inflate.zig:220-249.  Codes of up to 10 / 8 bits come out of the LDS tables built
// after the block header; longer ones (and invalid ones) take the canonical walk, which also
// keeps the reference's order of errors: a miss in the table of the decoder is InvalidCode
// before the bits are consumed (huffman_decoder.zig:156-175), running out of input is
// EndOfStream at the shift (bit_reader.zig:159-163).  Returns -1 at the end of the block.
__device__ __forceinline__ int fl_inf_dynamic_symbol(fl_bitr& r, FL_LDS fl_inflate_ws16* ws, fl_inf_out& o, uint32_t lane) {
    FL_TRY(fl_br_fill(r, 15));
    uint32_t sym, cb;
    {
        const uint32_t pk = fl_br_peek(r, 15);
        const uint32_t e = fl_uni(ws->lit_lut[pk & ((1u << FL_INF_LIT_BITS) - 1)]);
        if (e) {
            sym = e & 0x1ff;
            cb = (e >> 9) & 15;
        } else {
            FL_TRY(fl_hdec_find(&ws->lit, pk, 15, sym, cb));
            sym = fl_uni(sym);
            cb = fl_uni(cb);
        }
    }
    FL_TRY(fl_br_shift(r, cb));
    if (sym < 256) {
        FL_TRY(fl_inf_literal(o, sym, lane));
    } else if (sym == 256) {
        return -1;
    } else {
        FL_TRY(fl_br_fill(r, 5 + 15 + 13));
        uint32_t length, distance, dsym;
        FL_TRY(fl_inf_length(r, sym - 257, length));
        {
            const uint32_t pk = fl_br_peek(r, 15);
            const uint16_t e = (uint16_t)fl_uni(ws->dst_lut[pk & ((1u << FL_INF16_DST_BITS) - 1)]);
            if (e) {
                fl_dst_sym_cb(e, dsym, cb);
            } else {
                FL_TRY(fl_hdec_find(&ws->dst, pk, 15, dsym, cb));
                dsym = fl_uni(dsym);
                cb = fl_uni(cb);
            }
        }
        FL_TRY(fl_br_shift(r, cb));
        FL_TRY(fl_inf_distance(r, dsym, distance));
        FL_TRY(fl_inf_match(o, length, distance, lane));
    }
    return 0;
}

__device__ __forceinline__ int fl_inf_dynamic(fl_bitr& r, FL_LDS fl_inflate_ws16* ws, fl_inf_out& o, uint32_t lane) {
    for (;;) {
        if (r.left >= FL_INF_FAST_MIN_BITS) {
            int rc;
            do {
                rc = fl_inf_fast_round(r, ws, o, lane);
#ifdef FL_INF_COUNT
                if (blockIdx.x == 0 && lane == 0) { g_fl_prof[40]++; if (rc == 2) g_fl_prof[41]++; }
#endif
            } while (__builtin_expect(rc == 0 && r.left >= FL_INF_FAST_MIN_BITS, 1));
            fl_br_resync(r);
            if (rc == 1) return 0;
            if (rc > 2) return rc;
        }
        const int rc = (int)fl_uni((uint32_t)fl_inf_dynamic_symbol(r, ws, o, lane));
#ifdef FL_INF_COUNT
        if (blockIdx.x == 0 && lane == 0) g_fl_prof[42]++;
#endif
        if (rc < 0) return 0;
        if (rc) return rc;
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
#ifndef FL_INF_WAVES
#define FL_INF_WAVES 5
#endif
#ifndef FL_INF_ATTR
#define FL_INF_ATTR
#endif
template <uint32_t RING>
__global__ FL_INF_ATTR __launch_bounds__(64, (RING <= 4096u ? FL_INF_WAVES : 1)) void k_inflate(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks,
                                                int container, int flags, fl_crc_consts cc,
                                                uint8_t* __restrict__ out, uint64_t* __restrict__ out_len,
                                                int32_t* __restrict__ status, uint64_t* __restrict__ consumed,
                                                const int32_t* redo_only /* non-null: only streams marked -1 */) {
    __shared__ fl_inflate_ws16 ws_mem;
    __shared__ alignas(8) uint8_t ring_mem[RING];
#ifdef FL_INF_PAD  // tuning build: fewer streams per CU
    __shared__ uint32_t pad_mem[FL_INF_PAD / 4];
    if (out_len == nullptr) pad_mem[threadIdx.x] = 0;
#endif
    __shared__ uint32_t inring_mem[FL_INF_INRING / 4];
    FL_LDS fl_inflate_ws16* ws = (FL_LDS fl_inflate_ws16*)&ws_mem;
    // the CRC-32 table is only needed after the last block: it takes the place of the literal table
    FL_LDS uint32_t* crc_tab = (FL_LDS uint32_t*)ws->lit_lut;
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    const uint32_t lane = threadIdx.x;
    if (ck.skip) return;
    if (redo_only && redo_only[c] != -1) return;  // k_inflate_par has decoded this stream
    fl_bitr r;
    r.data = in + ck.in_off;
    r.nbytes = ck.in_len;
    r.left = (int64_t)ck.in_len * 8;
    r.lane = lane;
    r.inring = (FL_LDS uint32_t*)inring_mem;
    fl_br_seek(r, 0);
    fl_inf_out o;
    o.out = out + ck.out_off;
    o.ring = (FL_LDS uint8_t*)ring_mem;
    o.rmask = RING - 1;
    o.near_max = RING - 260;
    o.cap = ck.out_cap;
    o.wp = 0;
    o.flushed = 0;
    o.fenced = 0;
    o.bias = (uint32_t)((uintptr_t)o.out & 7);

    int rc = (int)fl_uni((uint32_t)fl_inf_header(r, container));
    while (rc == 0) {  // inflate.zig:251-280
        uint32_t bfinal, btype;
        if ((rc = (int)fl_uni((uint32_t)fl_br_read(r, 1, bfinal)))) break;
        if ((rc = (int)fl_uni((uint32_t)fl_br_read(r, 2, btype)))) break;
        if (btype == 2) {
            if ((rc = (int)fl_uni((uint32_t)fl_inf_dynamic_header(r, ws, flags, lane)))) break;
            for (uint32_t i = lane; i < 80; i += 64) ((FL_LDS uint32_t*)ws->lens)[i] = 0;  // the fast rounds' owner array
            fl_lds_order();
            rc = (int)fl_uni((uint32_t)fl_inf_dynamic(r, ws, o, lane));
        } else if (btype == 0) {
            rc = (int)fl_uni((uint32_t)fl_inf_stored(r, o, lane));
        } else if (btype == 1) {
            rc = (int)fl_uni((uint32_t)fl_inf_fixed(r, o, lane));
        } else {
            rc = 12;  // InvalidBlockType
        }
        if (rc) break;
        if (bfinal) {
            fl_br_align(r);
            // container.zig:154-166
            fl_inf_flush(o, o.wp, lane);
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
    fl_inf_flush(o, o.wp, lane);  // also after an error: what was decoded before it is delivered
    if (lane == 0) {
        out_len[c] = o.wp;
        status[c] = rc;
        if (consumed) consumed[c] = fl_br_consumed(r);
    }
}
