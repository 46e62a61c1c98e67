// kernels_inflate_par.h -- inflate of ONE long stream by a whole workgroup (16 waves).
//
// Reference path: Inflate.step / dynamicBlockHeader / dynamicBlock / storedBlock
// (inflate.zig:89-280), HuffmanDecoder.find (huffman_decoder.zig:156-175), CircularBuffer.writeMatch
// (CircularBuffer.zig:44-75), container header / footer (container.zig:111-166).
//
// k_inflate gives a stream one wavefront; its serial symbol walk costs about 400 cycles per token,
// so a batch of few long streams (BASELINE.json configs[4]: 128 x 1 MiB per GPU) leaves the chip
// idle at 1.7 GB/s.  Here the symbols of one stream are decoded by 16 waves at once:
//
//  * A block is decoded in rounds over a WINDOW of 4 KiB of compressed bytes; wave w owns the
//    2048 bits that start at window bit 2048 w.  A Huffman stream synchronises itself: decoding
//    from a wrong bit offset falls in step with the true symbol boundaries after a few symbols.
//    So every wave decodes its range from its nominal start, and remembers the set of bit positions
//    at which it saw a token start (a 2048-bit map).  Inside a wave, 64 lanes decode the tokens that
//    would start at 64 consecutive bit offsets from table lookups, and a short scalar walk over the
//    per-lane token lengths picks the offsets that are real starts.
//  * Stitch: the true path enters wave w's range at the bit where wave w-1's valid tokens end.
//    Each wave has already followed every one of the 48 possible entry bits through the per-offset
//    token lengths of its first 256 bits until it meets the wave's own path ("join"); the tokens
//    before the join come from that short walk, those after it are the wave's own.  No join within
//    256 bits: the window ends there and the next round starts at that bit.
//  * Output: the window's tokens get their output offsets from prefix sums.  Literal bytes and
//    bytes copied from the previous 32 KiB (kept in an LDS ring) are final at once; a byte copied
//    from inside the window points at its source and is resolved by pointer jumping over byte
//    positions (a few rounds: the depth of the copy-of-a-copy chains), so the LZ77 copies of a
//    window need no serial order.  A window that would produce more than 16 KiB is cut at a token.
//  * Block headers, the container header / footer and the code tables use the symbol-at-a-time
//    code of kernels_inflate.h, run by wave 0.
//
// This kernel only ever reports success.  Anything else -- an invalid code or
// distance on the true path, a checksum mismatch, an output slot that is too small, a truncated
// stream -- marks the stream FL_PAR_REDO and k_inflate decodes it again from the start: the
// reference's error names and their order stay the business of that one implementation.
//
// Bound: vector-ALU issue of the table decode (about 90 instructions per 64 bit offsets) and LDS
// latency; no MFMA (bit and byte work).
#pragma once
#include "kernels_inflate.h"
#include "kernels_lz.h"

#define FL_PAR_REDO (-1)  // status written for streams that k_inflate has to decode

// phase cycles of workgroup 0 (tuning aid, -DFL_PAR_PROF; read by tools/par_probe.py)
#ifdef FL_PAR_PROF
#define FP_WHY(n) (g_fl_prof[60] = (n), (uint32_t)(n))
#define FP_T(slot)                                                        \
    do {                                                                  \
        const uint64_t t_now_ = __builtin_readcyclecounter();             \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += t_now_ - t_prof_; \
        t_prof_ = t_now_;                                                 \
    } while (0)
#define FP_CNT(slot, v)                                                   \
    do {                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += (v);  \
    } while (0)
#else
#define FP_WHY(n) ((uint32_t)(n))  // (the reason a workgroup gives a stream up for: a span reports it as its status; 1: something else)
#define FP_T(slot)
#define FP_CNT(slot, v)
#endif

#define FP_WAVES 16
#define FP_THREADS (64 * FP_WAVES)
#define FP_SUBW 32                               // 64-bit sub-windows per wave and round
#define FP_WBITS (64 * FP_SUBW)                  // bits per wave and round
#define FP_WIN_BYTES (FP_WAVES * FP_WBITS / 8)   // compressed bytes per round
#define FP_STAGE_DW (FP_WIN_BYTES / 4 + 8)       // + the bytes the last tokens reach into
#ifndef FP_TOK_CAP
#define FP_TOK_CAP 512                           // tokens per wave and round
#endif
#ifndef FP_JOIN_BITS
#define FP_JOIN_BITS 1024                        // bits of a wave's range within which an entering path must join
#endif
#define FP_MAX_FIX 96                            // tokens an entering path may take before it joins
#define FP_ENTRIES 48                            // a token has at most 48 bits: entry bit < 48
#define FP_OUT_CAP 16384u                        // output bytes per round
#define FP_RING (32768u + FP_OUT_CAP)            // history + window
#ifndef FP_PTR_EXTRA
#define FP_PTR_EXTRA 0u
#endif
#define FP_RES 0xffffu                           // ptr value: byte is final
#define FP_DST_BITS 10                           // bits of the distance table
#define FP_NOJOIN 0xffffu
#define FP_JTERM 0x8000u

enum { FP_X_NORMAL = 0, FP_X_EOB = 1, FP_X_BAIL = 2, FP_X_FULL = 3, FP_X_CUT = 4 };

// tokens: literal = byte; match = 1 << 31 | length << 16 | distance - 1; terminators (never in a
// token list, only in the per-offset tables): 1 << 30 | kind << 8 | code bits
#define FP_TOK_MATCH(len, dist) (0x80000000u | ((len) << 16) | ((dist)-1u))
#define FP_TOK_TERM(kind, bits) (0x40000000u | ((uint32_t)(kind) << 8) | (bits))

struct fp_long {
    uint32_t lim[16], first[16], index[16];
};

struct fp_shared {
    fl_inflate_ws ws;
    uint32_t inring[FL_INF_INRING / 4];  // wave 0's symbol-at-a-time reader
    uint32_t stage[FP_STAGE_DW];         // the window's compressed bytes
    uint32_t tok[FP_WAVES][FP_TOK_CAP];
    uint32_t bitmap[FP_WAVES][FP_WBITS / 32];
    uint8_t tb[FP_WAVES][FP_JOIN_BITS];    // per bit offset of the first 512 bits: bits of the token that would start there (255 = terminator)
    uint32_t fixtok[FP_WAVES][FP_MAX_FIX];
    uint16_t fixpos[FP_WAVES][FP_MAX_FIX];
    uint8_t ring[FP_RING];
    uint16_t ptr[FP_OUT_CAP + FP_PTR_EXTRA];  // (+ what the 16-bit ring of the other mode needs beyond ring + ptr)
    uint32_t dst_big[1u << FP_DST_BITS];  // this kernel's distance table (k_inflate's has 8 bits: one pass in six met a longer code)
    // per wave
    uint32_t w_ntok[FP_WAVES], w_xkind[FP_WAVES], w_xpos[FP_WAVES];
    uint32_t w_valid[FP_WAVES], w_entry[FP_WAVES], w_fv[FP_WAVES], w_nfix[FP_WAVES], w_nown[FP_WAVES];
    uint32_t w_total[FP_WAVES], w_base[FP_WAVES], w_endk[FP_WAVES], w_endp[FP_WAVES];
    // stream / round state
    uint64_t bitpos, wp;
    fp_long lit_long, dst_long;
    uint32_t r_nvalid, r_kind, r_next, r_cutwave, r_cutbudget, r_nout;
    uint32_t redo, blk_done, blk_final, blk_type, unresolved[3];  // one flag per resolve round, three in rotation
    uint32_t n_pieces, piece_id[2];  // spans, run A: pieces of the pool this span has taken; the last two of them
    uint32_t piece_idb[2];           // symbols: the last two pieces of plane B
    uint32_t nblk, stop, uses_hist;  // spans: blocks decoded so far; the span ends here; a copy reached before its start
    uint32_t err_far;  // a copy of this round reaches before the start of the output (set in step 6, read after its barrier)
    uint32_t st_len;  // stored block: bytes
    uint32_t wbits;   // bits per wave and round
    uint32_t crc, adA, adB;
};

// Codes longer than the lookup table, without a loop: canonical codes of one length are consecutive
// numbers, so a 15-bit window (first stream bit = most significant) belongs to length L iff it is
// below lim[L] = (first code of L + count of L) << (15 - L) and not below lim[L - 1]
// (huffman_decoder.zig:156-175 finds the same symbol bit by bit).
template <class H>
__device__ __forceinline__ void fp_long_build(const FL_LDS H* d, FL_LDS fp_long* t, uint32_t lane) {
    uint32_t code = 0, idx = 0;
    for (uint32_t len = 1; len <= 15; len++) {  // (every lane the same values, lane 0 stores)
        const uint32_t count = d->count[len];
        if (lane == 0) {
            t->first[len] = code;
            t->index[len] = idx;
            t->lim[len] = (code + count) << (15 - len);
        }
        code = (code + count) << 1;
        idx += count;
    }
    fl_wave_lds_sync();
}
template <int TBITS, class H>
__device__ __forceinline__ int fp_find_long(const FL_LDS H* d, const FL_LDS fp_long* t, uint32_t peek15, uint32_t& sym,
                                            uint32_t& code_bits) {
    const uint32_t c = __brev(peek15) >> 17;
    uint32_t len = TBITS + 1;
#pragma unroll
    for (int k = TBITS + 1; k <= 14; k++) len += c >= t->lim[k] ? 1u : 0u;
    if (c >= t->lim[15]) return 7;  // InvalidCode
    sym = d->symbol[t->index[len] + ((c >> (15 - len)) - t->first[len])];
    code_bits = len;
    return 0;
}

// Table entries of this kernel (converted from k_inflate's after every block header):
//   literal / length: code bits | extra bits << 4 | value << 8 (byte, or base length) | EOB << 30 | length << 31
//   distance:         code bits | extra bits << 4 | base distance << 8
//   bit 29: not a valid symbol (286, 287 / 30, 31); 0: the code is longer than the table
#define FP_E_BAD (1u << 29)
#define FP_E_EOB (1u << 30)
#define FP_E_LEN (1u << 31)
__device__ __forceinline__ uint32_t fp_lit_entry(uint32_t sym, uint32_t cb) {
    if (sym < 256) return cb | (sym << 8);
    if (sym == 256) return cb | FP_E_EOB;
    if (sym > 285) return cb | FP_E_BAD;
    return cb | (fl_len_extra_bits(sym - 257) << 4) | ((fl_len_base_scaled(sym - 257) + 3) << 8) | FP_E_LEN;
}
__device__ __forceinline__ uint32_t fp_dst_entry(uint32_t sym, uint32_t cb) {
    if (sym > 29) return cb | FP_E_BAD;
    return cb | (fl_dist_extra_bits(sym) << 4) | ((fl_dist_base_scaled(sym) + 1) << 8);
}
// k_inflate's entry (symbol | code_bits << 9 | ...) -> this kernel's
__device__ __noinline__ void fp_convert_luts(FL_LDS fl_inflate_ws* ws, FL_LDS uint32_t* big, uint32_t lane) {
    fl_hdec_build_lut<true>(&ws->dst, big, FP_DST_BITS, lane);  // (k_inflate's entries, FP_DST_BITS of them)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (uint32_t i = lane; i < (1u << FL_INF_LIT_BITS); i += 64) {
        const uint32_t e = ws->lit_lut[i];
        uint32_t v = 0;
        if (e != 0) v = fp_lit_entry(e & 511, (e >> 9) & 15);
        ws->lit_lut[i] = v;
    }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (uint32_t i = lane; i < (1u << FP_DST_BITS); i += 64) {
        const uint32_t e = big[i];
        uint32_t v = 0;
        if (e != 0) v = fp_dst_entry(e & 511, (e >> 9) & 15);
        big[i] = v;
    }
    fl_wave_lds_sync();
}

// The token that would start at a bit offset whose next 64 stream bits are hi:lo.
// bits = its length in bits (terminators: the code's bits), pay = the token.  Straight-line for the
// common case; codes longer than the tables take fp_find_long (whole wave, when any lane needs it).
__device__ __forceinline__ void fp_decode_at(const FL_LDS fp_shared* sh, uint32_t lo, uint32_t hi, uint32_t& bits,
                                             uint32_t& pay) {
    const FL_LDS fl_inflate_ws* ws = &sh->ws;
    uint32_t le = ws->lit_lut[lo & ((1u << FL_INF_LIT_BITS) - 1)];
    FP_CNT(61, 1);
    if (__any(le == 0)) {
        FP_CNT(62, 1);
        if (le == 0) {
            uint32_t sym, cb;
            le = fp_find_long<FL_INF_LIT_BITS>(&ws->lit, &sh->lit_long, lo & 0x7fffu, sym, cb) ? FP_E_BAD : fp_lit_entry(sym, cb);
        }
    }
    const uint32_t cb = le & 15, eb = (le >> 4) & 15, val = (le >> 8) & 0x1ff;
    const uint32_t length = val + ((lo >> cb) & ((1u << eb) - 1u));
    const uint32_t lb = cb + eb;  // <= 20
    const uint32_t dw = lb ? ((lo >> lb) | (hi << (32 - lb))) : lo;  // lb <= 20
    uint32_t de = sh->dst_big[dw & ((1u << FP_DST_BITS) - 1)];
    const bool is_len = (le >> 31) != 0;
    if (__any(is_len && de == 0)) {
        FP_CNT(63, 1);
        if (is_len && de == 0) {
            uint32_t dsym, dcb;
            de = fp_find_long<FP_DST_BITS>(&ws->dst, &sh->dst_long, dw & 0x7fffu, dsym, dcb) ? FP_E_BAD : fp_dst_entry(dsym, dcb);
        }
    }
    const uint32_t dcb = de & 15, deb = (de >> 4) & 15, dval = (de >> 8) & 0xffff;
    const uint32_t distance = dval + ((dw >> dcb) & ((1u << deb) - 1u));
    const bool bad = (le & FP_E_BAD) || (is_len && (de & FP_E_BAD));
    bits = is_len ? lb + dcb + deb : cb;  // <= 48
    pay = is_len ? FP_TOK_MATCH(length, distance) : ((le & FP_E_EOB) ? FP_TOK_TERM(FP_X_EOB, cb) : val);
    if (bad) {
        bits = 0;
        pay = FP_TOK_TERM(FP_X_BAIL, 0);
    }
}

// the 64 stream bits that start at window bit `bp`
__device__ __forceinline__ void fp_fetch64(const FL_LDS uint32_t* stage, uint32_t bp, uint32_t& lo, uint32_t& hi) {
    const uint32_t di = bp >> 5, sh = bp & 31;
    const uint32_t d0 = stage[di], d1 = stage[di + 1], d2 = stage[di + 2];
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
}

template <uint32_t RING = FP_RING>
__device__ __forceinline__ uint32_t fp_ring_idx(uint32_t wbase, int32_t rel) {  // rel in [-32768, the round's output cap)
    uint32_t x = wbase + (uint32_t)((int32_t)RING + rel);  // wbase < RING
    if (x >= RING) x -= RING;
    if (x >= RING) x -= RING;
    return x;
}
// SYMBOLS (spans that do not know their history, round 5): the ring holds 16 bits per output byte -- below 256 a byte's value,
// 256 + h "whatever byte h of the 32 KiB before the span is", 0xC000 + o "not resolved yet: what the window's byte o is" -- so one
// decode gives what runs A and B gave (a = e & 255, b = a ^ (e >> 8): the two fillings of the history, byte for byte).  The 16-bit
// ring lies where the byte ring and the pointers of the other mode lie; a round makes 8 KiB at most instead of 16.
#ifndef FP_RESOLVE_SWEEPS
#define FP_RESOLVE_SWEEPS 1
#endif
#ifndef FP_ADAPT_WBITS
#define FP_ADAPT_WBITS 0
#endif
#ifndef FP_SYM_OUT_CAP
#define FP_SYM_OUT_CAP 8192u
#endif
#define FP_SYM_RING (32768u + FP_SYM_OUT_CAP)
#define FP_SYM_PTR 0xC000u
#ifndef FP_SYM_WBITS
#define FP_SYM_WBITS 1536u  // compressed bits per wave and round: what 8 KiB of output hold (text: 2.6 bytes out per byte in)
#endif

__device__ __forceinline__ uint32_t fp_tok_len(uint32_t t) { return (t >> 31) ? ((t >> 16) & 0x1ffu) : 1u; }

// ------------------------------------------------------------------ spans: one long stream, many workgroups
// (round 3).  A stream is cut at block starts found by k_span_scan (dynamic-block headers that parse) into SPANS;
// a span is decoded by one workgroup from its block start until it lands on the start of another span or behind
// the final block.  What a span cannot know is the 32 KiB of output before it, and how far into the output it
// starts:
//   runs A and B (MODE 1)  every span at once: where the span ends, how many bytes it makes, the last 32 KiB of them
//                    (its TAIL), and the bytes themselves -- with two fillings of the unknown history: byte h of
//                    it is A[h] = h & 255 in run A and B[h] = A[h] ^ ((h >> 8) + 1) in run B.  A byte that comes
//                    out the same in both runs does not depend on the history; one that differs is a copy (of a
//                    copy ...) of history byte h = ((a ^ b) - 1) << 8 | a -- the same h in both runs, because
//                    where a byte is copied from is decided by the tokens, not by the bytes.  Run A does not know
//                    where its bytes belong (pieces of a pool); before run B
//   the host         follows the chain of spans from the stream start (a span that is not landed on is dead) and
//                    adds up the output offsets: run B writes in place;
//   k_span_resolve   the true tails, span after span along the chain: a byte that depends on the history is
//                    byte h of the (already true) tail before it;
//   k_span_fix       every byte of every live span once: a == b: final; else byte h of the true tail before the
//                    span; and the checksum pieces.
// The pieces are folded and the footer is checked by the host.  Anything irregular: the stream is decoded again by
// k_inflate_par / k_inflate, as before.  inflate.zig:220-239 is serial per stream; this is the same function, with
// the one thing a span cannot know kept symbolic until it is known.
struct fl_span {
    uint64_t start_bit;  // of its first block header, from the first byte of the stream (first span: unused)
    uint64_t wp;         // run B: offset of its first output byte in the stream's output (run A: 0)
    uint32_t stream;     // chunk index
    uint32_t first;      // 1: starts at the stream's first byte (container header)
    uint32_t prev;       // the span before it in the chain (its tail is this span's history), ~0u: none
    uint32_t live;       // run B: on the chain
};
struct fl_span_res {
    uint64_t end_bit;  // where it stopped: the start of another span, or the bit behind the final block
    uint64_t out_len;
    uint32_t status;      // 0: decoded; anything else: the stream goes the old way
    uint32_t final_seen;  // stopped behind the final block
    uint32_t uses_hist;   // a copy reaches before the span's first byte
    uint32_t n_pieces;    // run A: pieces of the pool taken
};
#define FP_TAIL 32768u
#define FP_NO_SPAN 0xffffffffu
// Run A of a span that is not the first of its stream does not know where its bytes belong: they go to PIECES of a
// pool, taken as the span grows (one atomic per 64 KiB); k_span_fix moves them to their place.
#define FP_PIECE_LOG 16u
#define FP_PIECE (1u << FP_PIECE_LOG)
#define FP_MAX_PIECES 2048u  // per span: 128 MiB of output
struct fl_span_pool {
    uint8_t* base;    // pieces of FP_PIECE bytes
    uint32_t* next;   // [0]: pieces handed out so far
    uint32_t* tab;    // [span][FP_MAX_PIECES]: a span's pieces in order
    uint32_t pieces;  // in the pool
    uint32_t pad;
};

// MODE 0: the whole stream (k_inflate_par); 1: a span, run A (fill 0) or B (fill 1) (k_inflate_span)
template <int MODE, bool SYM = false>
__device__ __forceinline__ void fp_body(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks, int container,
                                        int flags, uint32_t min_bytes, fl_crc_consts cc, uint8_t* __restrict__ out,
                                        uint64_t* __restrict__ out_len, int32_t* __restrict__ status,
                                        uint64_t* __restrict__ consumed, const fl_span* __restrict__ spans,
                                        fl_span_res* __restrict__ sres, const uint64_t* __restrict__ cand,
                                        const uint32_t* __restrict__ cand_off, uint8_t* __restrict__ tails, uint32_t fill,
                                        fl_span_pool pool, uint32_t unit, bool b_pool, FL_LDS fp_shared* sh,
                                        uint32_t b_base = 0u, uint8_t* __restrict__ tails_b = nullptr) {
    // unit: the stream (MODE 0) / the span (MODE 1); b_pool: run B at the same time as run A, its bytes to the pool as well
    // SYM: one decode in symbols, both planes to the pool (plane B's pieces: table rows b_base + unit) and both tails
    constexpr uint32_t OUT_CAP = SYM ? FP_SYM_OUT_CAP : FP_OUT_CAP, RING = 32768u + OUT_CAP;
    static_assert(offsetof(fp_shared, ptr) == offsetof(fp_shared, ring) + FP_RING && 2u * FP_SYM_RING <= FP_RING + 2u * (FP_OUT_CAP + FP_PTR_EXTRA), "the 16-bit ring lies over ring + ptr");
    FL_LDS uint16_t* ring16 = (FL_LDS uint16_t*)sh->ring;
    FL_LDS fl_inflate_ws* ws = &sh->ws;
    fl_span sp;
    sp.start_bit = 0;
    sp.wp = 0;
    sp.stream = unit;
    sp.first = 1;
    sp.prev = FP_NO_SPAN;
    sp.live = 1;
    if (MODE != 0) sp = spans[unit];
    if (MODE == 1 && fill && ((!b_pool && !sp.live) || sp.first)) return;  // (run A finds out which spans are live; a first span's run A is final)
    const uint32_t c = sp.stream;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (MODE == 0 && ((ck.in_len < min_bytes && ck.out_cap < 16ull * min_bytes) || (flags & 1))) {  // (reference-strict Q6 headers: k_inflate's job as well)
        if (tid == 0) status[c] = FL_PAR_REDO;
        return;
    }
    const uint8_t* src = in + ck.in_off;
    uint8_t* dst = out + ck.out_off + sp.wp;  // (sh->wp counts from the span's first output byte)
    // run A of a span that does not know its place (sp.wp is 0 in run A, the host sets it before run B)
    const bool to_pool = MODE == 1 && (!fill || b_pool) && !sp.first;
    const uint64_t out_room = to_pool ? ~0ull : (ck.out_cap > sp.wp ? ck.out_cap - sp.wp : 0ull);
    // output bytes that exist before the span's first: a distance may reach that far back
    const uint64_t hist_avail = MODE == 0 || sp.first ? 0ull : (fill && !b_pool) ? min((uint64_t)FP_TAIL, sp.wp) : (uint64_t)FP_TAIL;
    const uint64_t total_bits = (uint64_t)ck.in_len * 8;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

    fl_bitr r;  // wave 0 only
    r.data = src;
    r.nbytes = ck.in_len;
    r.lane = lane;
    r.inring = (FL_LDS uint32_t*)sh->inring;
    r.left = (int64_t)total_bits;
    auto reader_at = [&](uint64_t bitpos) {  // restart the symbol-at-a-time reader at a bit position
        r.left = (int64_t)(total_bits - bitpos);
        fl_br_seek(r, (uint32_t)(bitpos >> 3));
        const uint32_t k = (uint32_t)bitpos & 7;
        if (k) {
            fl_br_refill(r);
            r.buf >>= k;
            r.have -= k;
        }
    };
    auto reader_pos = [&]() -> uint64_t { return total_bits - (uint64_t)r.left; };
    // where output byte o of the span goes
    auto out_at = [&](uint64_t o) -> uint8_t* {
        if (MODE == 1 && to_pool)
            return pool.base + (uint64_t)sh->piece_id[(uint32_t)(o >> FP_PIECE_LOG) & 1u] * FP_PIECE + ((uint32_t)o & (FP_PIECE - 1u));
        return dst + o;
    };
    auto out_b_at = [&](uint64_t o) -> uint8_t* {  // (SYM) plane B
        return pool.base + (uint64_t)sh->piece_idb[(uint32_t)(o >> FP_PIECE_LOG) & 1u] * FP_PIECE + ((uint32_t)o & (FP_PIECE - 1u));
    };
    // (thread 0) pieces for the span's bytes below `end`; the bytes about to be written lie in the last two
    auto pieces_to = [&](uint64_t end) -> bool {
        const uint64_t need = (end + FP_PIECE - 1u) >> FP_PIECE_LOG;
        uint32_t n = sh->n_pieces;
        while (n < need) {
            if (n >= FP_MAX_PIECES) return false;
            const uint32_t id = atomicAdd(pool.next, 1u);
            if (id >= pool.pieces) return false;
            pool.tab[(uint64_t)unit * FP_MAX_PIECES + n] = id;
            sh->piece_id[n & 1u] = id;
            if (SYM) {
                const uint32_t idb = atomicAdd(pool.next, 1u);
                if (idb >= pool.pieces) return false;
                pool.tab[(uint64_t)(b_base + unit) * FP_MAX_PIECES + n] = idb;
                sh->piece_idb[n & 1u] = idb;
            }
            n++;
        }
        sh->n_pieces = n;
        return true;
    };

    if (tid == 0) {
        sh->redo = 0;
        sh->wp = 0;
        sh->wbits = SYM ? FP_SYM_WBITS : FP_WBITS;
    }
    if (wave == 0) {
        if (sp.first) {
            fl_br_seek(r, 0);
            const int rc = fl_inf_header(r, container);
            if (lane == 0) {
                if (rc) sh->redo = FP_WHY(1);
                sh->bitpos = reader_pos();
            }
        } else if (lane == 0) {
            sh->bitpos = sp.start_bit;
        }
    }
    if (SYM) {
        for (uint32_t hh = tid; hh < FP_TAIL; hh += FP_THREADS) ring16[RING - FP_TAIL + hh] = (uint16_t)(256u + hh);
        if (tid == 0) {
            sh->nblk = 0;
            sh->uses_hist = 0;
            sh->n_pieces = 0;
        }
    } else if (MODE != 0) {
        // the 32 KiB before the span: filling A or B (see above)
        for (uint32_t i = tid; i < FP_TAIL / 4; i += FP_THREADS) {
            // bytes 4 i .. 4 i + 3: A = low byte of the index, B = A ^ (high byte of the index + 1)
            const uint32_t lowb = ((4 * i) & 0xff) * 0x01010101u + 0x03020100u;
            const uint32_t hib = ((4 * i) >> 8) + 1u;
            const uint32_t v = fill ? (lowb ^ (hib * 0x01010101u)) : lowb;
            ((FL_LDS uint32_t*)&sh->ring[FP_RING - FP_TAIL])[i] = v;
        }
        if (tid == 0) {
            sh->nblk = 0;
            sh->uses_hist = 0;
            sh->n_pieces = 0;
        }
    }
    __syncthreads();

#ifdef FL_PAR_PROF
    uint64_t t_prof_ = __builtin_readcyclecounter();
#endif
    // ================================================================ blocks
    // The flags in LDS (redo, blk_done, blk_final, blk_type, err_far) are only read right behind a barrier that
    // follows their last write, into registers, and another barrier stands between those reads and the next write:
    // every wave takes the same way through the loops.  `bail` = the stream goes to k_inflate.
    bool bail = false, final_seen = false;
    for (;;) {
        if (MODE != 0 && tid == 0) {
            // a span ends where another one starts (the sorted list of the stream's span starts)
            uint32_t stop = 0;
            if (sh->nblk != 0) {
                const uint64_t bp = sh->bitpos;
                uint32_t lo_i = cand_off[c], hi_i = cand_off[c + 1];
                while (lo_i < hi_i) {
                    const uint32_t mid = (lo_i + hi_i) >> 1;
                    if (cand[mid] < bp) lo_i = mid + 1; else hi_i = mid;
                }
                stop = lo_i < cand_off[c + 1] && cand[lo_i] == bp;
            }
            sh->stop = stop;
            sh->nblk = sh->nblk + 1;
        }
        __syncthreads();  // (nobody writes between here and the header below)
        bail = sh->redo != 0;
        const bool stop_here = MODE != 0 && sh->stop != 0;
        __syncthreads();  // every wave has looked before wave 0 parses the next header
        if (bail || stop_here) break;
        FP_T(32);
        // ---- block header (wave 0)
        if (wave == 0) {
            reader_at(sh->bitpos);
            uint32_t bfinal = 0, btype = 3;
            int rc = fl_br_read(r, 1, bfinal);
            if (!rc) rc = fl_br_read(r, 2, btype);
            uint32_t stlen = 0;
            if (!rc && btype == 2) {
                rc = fl_inf_dynamic_header(r, ws, flags, lane);
                if (!rc) {
                    fp_long_build(&ws->lit, &sh->lit_long, lane);
                    fp_long_build(&ws->dst, &sh->dst_long, lane);
                    fp_convert_luts(ws, (FL_LDS uint32_t*)sh->dst_big, lane);
                }
            } else if (!rc && btype == 1) {
                // fixed codes (RFC 1951 3.2.6, inflate.zig:104-121) through the same tables: lengths 8/9/7/8
                // for the 288 literal/length codes, 5 for the 32 distance codes (286, 287, 30, 31: invalid)
                fl_wave_lds_sync();
                for (uint32_t i = lane; i < 320; i += 64)
                    ws->lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
                fl_wave_lds_sync();
                rc = fl_hdec_generate(&ws->lit, ws->lens, ws->offs, 288, 286, 15, lane);
                if (!rc) rc = fl_hdec_generate(&ws->dst, ws->lens + 288, ws->offs, 32, 30, 15, lane);
                if (!rc) {
                    fl_hdec_build_lut<false>(&ws->lit, ws->lit_lut, FL_INF_LIT_BITS, lane);
                    fl_hdec_build_lut<true>(&ws->dst, ws->dst_lut, FL_INF_DST_BITS, lane);
                    fp_long_build(&ws->lit, &sh->lit_long, lane);
                    fp_long_build(&ws->dst, &sh->dst_long, lane);
                    fp_convert_luts(ws, (FL_LDS uint32_t*)sh->dst_big, lane);
                }
            } else if (!rc && btype == 0) {
                fl_br_align(r);
                uint32_t len = 0, nlen = 0;
                rc = fl_br_read(r, 16, len);
                if (!rc) rc = fl_br_read(r, 16, nlen);
                if (!rc && len != ((~nlen) & 0xffff)) rc = 13;
                if (!rc && (int64_t)len * 8 > r.left) rc = 1;
                stlen = len;
            } else if (!rc) {
                rc = 12;  // invalid block type: k_inflate
            }
            if (lane == 0) {
                if (rc) sh->redo = FP_WHY(2);
                sh->bitpos = reader_pos();
                sh->blk_final = bfinal;
                sh->blk_type = btype;
                sh->st_len = stlen;
                sh->blk_done = 0;
            }
        }
        __syncthreads();
        FP_T(33);
        const uint32_t blk_type = sh->blk_type, blk_final = sh->blk_final;
        bail = sh->redo != 0;
        if (bail) break;

        if (blk_type == 0) {
            // ---- stored block (inflate.zig:89-102): bytes go out as they are, the ring keeps the last of them
            const uint32_t len = sh->st_len;
            const uint64_t wp = sh->wp;
            const uint32_t from = (uint32_t)(sh->bitpos >> 3);
            if (wp + len > out_room) {
                bail = true;  // (the same in every thread) OutputTooSmall is k_inflate's to report
            } else if (MODE == 1 && to_pool) {
                if (tid == 0 && !pieces_to(wp + len)) sh->redo = FP_WHY(9);
                __syncthreads();
                bail = sh->redo != 0;
            }
            if (!bail) {
                for (uint32_t i = tid; i < len; i += FP_THREADS) {
                    *out_at(wp + i) = src[from + i];
                    if (SYM) *out_b_at(wp + i) = src[from + i];
                }
                const uint32_t tail = min(len, RING);
                for (uint32_t i = tid; i < tail; i += FP_THREADS) {
                    const uint64_t o = wp + len - tail + i;
                    if (SYM)
                        ring16[(uint32_t)(o % RING)] = src[from + len - tail + i];
                    else
                        sh->ring[(uint32_t)(o % RING)] = src[from + len - tail + i];
                }
            }
            __syncthreads();
            if (tid == 0) {
                sh->wp = wp + len;
                sh->bitpos += (uint64_t)len * 8;
            }
            __syncthreads();
            if (bail) break;
        } else {
            // ---- Huffman block: rounds over windows of 4 KiB
            for (;;) {
                const uint64_t bitpos = sh->bitpos;
                const uint64_t wp = sh->wp;
                const uint32_t byte0 = (uint32_t)(bitpos >> 3), bit0 = (uint32_t)bitpos & 7;
                const uint32_t wbase = (uint32_t)(wp % RING);
                // bits per wave: halved (for the rest of the stream) when a wave ran out of token slots
                const uint32_t wbits = sh->wbits, jbits = min(wbits, (uint32_t)FP_JOIN_BITS);
                // (1) stage the window
                for (uint32_t i = tid; i < FP_STAGE_DW; i += FP_THREADS)
                    sh->stage[i] = fl_load_u32_clamped(src, byte0 + 4 * i, ck.in_len);
                sh->bitmap[wave][lane] = 0;
                __syncthreads();
                FP_T(34);
                FP_CNT(48, 1);
                // (2) every wave decodes its 2048 bits from its nominal start
                {
                    const uint32_t wb0 = bit0 + wave * wbits;
                    uint32_t carry = 0, ntok = 0, xkind = FP_X_NORMAL, xpos = 0;
                    uint32_t s_tab = 0;  // sub-windows whose per-offset token lengths are stored
                    for (uint32_t s = 0; s < wbits / 64; s++) {
                        if (ntok > FP_TOK_CAP - 64) {
                            xkind = FP_X_FULL;
                            xpos = 64 * s + carry;
                            break;
                        }
                        uint32_t lo, hi, bits, pay;
                        fp_fetch64(sh->stage, wb0 + 64 * s + lane, lo, hi);
                        fp_decode_at(sh, lo, hi, bits, pay);
                        const uint32_t nextv = ((pay >> 30) == 1) ? 255u : lane + bits;
                        if (s < jbits / 64) {
                            sh->tb[wave][64 * s + lane] = nextv == 255u ? (uint8_t)255 : (uint8_t)bits;
                            s_tab = s + 1;
                        }
                        // scalar walk over the real starts of this sub-window (kept to a handful of scalar
                        // instructions per token: the one scalar unit of the CU serves all 16 waves)
                        // (four instructions per token, by hand, as in k_inflate: the position is kept as p - 64 mod 2^32 --
                        // s_bitset1 and the lane select look at its low six bits only -- and its sum with the token's bits
                        // carries exactly when the next start is beyond this sub-window; a terminator counts as 255 bits)
                        uint64_t mask = 0;
                        const uint32_t relv = nextv == 255u ? 255u : bits;
                        uint32_t pb = carry - 64u, pn;
                        asm volatile(
                            "1:\n\t"
                            "s_bitset1_b64 %[S], %[p]\n\t"
                            "v_readlane_b32 %[n], %[nb], %[p]\n\t"
                            "s_add_u32 %[p], %[p], %[n]\n\t"
                            "s_cbranch_scc0 1b\n\t"
                            : [S] "+s"(mask), [p] "+s"(pb), [n] "=&s"(pn)
                            : [nb] "v"(relv)
                            : "scc");
                        const uint32_t p = pb + 64u - pn;   // the last token's start
                        const uint32_t nx = pn == 255u ? 255u : pb + 64u;
                        const uint32_t term = nx == 255u ? p + 1 : 0u;  // the token at p ends the wave's path
                        if (lane == 0) {
                            sh->bitmap[wave][2 * s] = (uint32_t)mask;
                            sh->bitmap[wave][2 * s + 1] = (uint32_t)(mask >> 32);
                        }
                        const uint64_t tokmask = term ? (mask & ~(1ull << (term - 1))) : mask;
                        if ((tokmask >> lane) & 1) sh->tok[wave][ntok + (uint32_t)__popcll(tokmask & lt_mask)] = pay;
                        ntok += (uint32_t)__popcll(tokmask);
                        if (term) {
                            const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)pay, (int)(term - 1));
                            xkind = (tp >> 8) & 0xff;
                            xpos = 64 * s + (term - 1) + (xkind == FP_X_EOB ? (tp & 0xff) : 0u);
                            break;
                        }
                        carry = nx - 64;
                    }
                    if (xkind == FP_X_NORMAL) xpos = wbits + carry;
                    for (uint32_t s = s_tab; s < jbits / 64; s++) {  // (the wave's own path ended early)
                        uint32_t lo, hi, bits, pay;
                        fp_fetch64(sh->stage, wb0 + 64 * s + lane, lo, hi);
                        fp_decode_at(sh, lo, hi, bits, pay);
                        sh->tb[wave][64 * s + lane] = ((pay >> 30) == 1) ? (uint8_t)255 : (uint8_t)bits;
                    }
                    if (lane == 0) {
                        sh->w_ntok[wave] = ntok;
                        sh->w_xkind[wave] = xkind;
                        sh->w_xpos[wave] = xpos;
                    }
                    fl_lds_order();
                    FP_T(41);
                }
                __syncthreads();
                FP_T(35);
                // (4) stitch.  Once a path has joined a wave's own path it leaves the wave where that one does,
                // whatever its entry bit was: every wave can judge its own entry, the window ends at the first
                // wave that ends it.
                {
                    const uint32_t w = wave;
                    uint32_t fv = 0, nfix = 0, e = 0, endk = FP_X_NORMAL, endp = 0;  // endk != NORMAL: the window ends in this wave
                    bool enter = true;  // false: the window ends before this wave's tokens
                    const uint32_t ntok = sh->w_ntok[w];
                    uint32_t xk = sh->w_xkind[w], xp = sh->w_xpos[w];
                    if (w > 0) {
                        if (sh->w_xkind[w - 1] != FP_X_NORMAL) {
                            enter = false;  // (an earlier wave ends the window anyway)
                            endk = FP_X_CUT;
                        } else {
                            // follow the entering path through the per-offset token lengths until it meets the
                            // wave's own path (wave-uniform; its tokens are decoded in step 5)
                            e = fl_uni(sh->w_xpos[w - 1]) - wbits;
                            uint32_t p = e;
                            int how = 0;  // 1 joined, 2 ended by a terminator, 0 no join
                            for (;;) {
                                if (p >= jbits) break;
                                if ((fl_uni(sh->bitmap[w][p >> 5]) >> (p & 31)) & 1) {
                                    how = 1;
                                    break;
                                }
                                const uint32_t t = fl_uni(sh->tb[w][p]);
                                if (t == 255u) {
                                    how = 2;
                                    break;
                                }
                                if (nfix == FP_MAX_FIX) break;
                                if (lane == 0) sh->fixpos[w][nfix] = (uint16_t)p;
                                p += t;
                                nfix++;
                            }
                            if (how == 0) {  // the window ends where this wave's range is entered
                                enter = false;
                                endk = FP_X_CUT;
                                endp = e;
                                nfix = 0;
                                FP_CNT(57, 1);
                            } else if (how == 2) {  // the entering path ends before it joins: EOB or a bad code
                                uint32_t lo, hi, tbits, tp;
                                fp_fetch64(sh->stage, bit0 + w * wbits + p, lo, hi);
                                fp_decode_at(sh, lo, hi, tbits, tp);
                                tp = fl_uni(tp);
                                xk = (tp >> 8) & 0xff;
                                xp = p + (xk == FP_X_EOB ? (tp & 0xff) : 0u);
                                fv = ntok;
                            } else {
                                uint32_t cnt = 0;
                                if (lane < (p >> 5)) cnt = (uint32_t)__popc(sh->bitmap[w][lane]);
                                if (lane == (p >> 5)) cnt = (uint32_t)__popc(sh->bitmap[w][lane] & ((1u << (p & 31)) - 1u));
                                fv = fl_wave_sum(cnt);
                                if (fv > ntok) fv = ntok;  // a join at the wave's terminator: no own tokens
                            }
                        }
                    }
                    if (enter && xk != FP_X_NORMAL) {
                        endk = xk;
                        endp = xp;
                    }
                    if (lane == 0) {
                        sh->w_valid[w] = enter ? 1u : 0u;
                        sh->w_entry[w] = e;
                        sh->w_fv[w] = fv;
                        sh->w_nfix[w] = nfix;
                        sh->w_nown[w] = ntok - fv;
                        sh->w_endk[w] = endk;
                        sh->w_endp[w] = endp;
                    }
                    // (5) the tokens of the entering path; output bytes of this wave's part of the window
                    // (for every wave: which ones count is known after the barrier)
                    if (enter) {
                        for (uint32_t k = lane; k < nfix; k += 64) {
                            uint32_t lo, hi, tbits, tp;
                            fp_fetch64(sh->stage, bit0 + wave * wbits + sh->fixpos[wave][k], lo, hi);
                            fp_decode_at(sh, lo, hi, tbits, tp);
                            sh->fixtok[wave][k] = tp;
                        }
                        fl_lds_order();
                        const uint32_t nown = ntok - fv, n = nfix + nown;
                        uint32_t sum = 0;
                        for (uint32_t k = lane; k < n; k += 64)
                            sum += fp_tok_len(k < nfix ? sh->fixtok[wave][k] : sh->tok[wave][fv + k - nfix]);
                        sum = fl_wave_sum(sum);
                        if (lane == 0) sh->w_total[wave] = sum;
                    }
                }
                __syncthreads();
                FP_T(36);
                if (wave == 0) {
                    // (lane w speaks for wave w: where the window ends, the output offsets of the waves before that)
                    const bool in = lane < FP_WAVES;
                    const uint32_t endk = in ? sh->w_endk[lane] : (uint32_t)FP_X_NORMAL;
                    const uint32_t valid = in ? sh->w_valid[lane] : 0u, endp = in ? sh->w_endp[lane] : 0u;
                    const uint32_t total = in ? sh->w_total[lane] : 0u;
                    uint32_t nvalid = FP_WAVES, kind = FP_X_NORMAL, next = FP_WAVES * wbits + (sh->w_xpos[FP_WAVES - 1] - wbits);
                    const uint64_t enders = __ballot(in && endk != FP_X_NORMAL);
                    if (enders) {
                        const int fw = __builtin_ctzll(enders);
                        kind = (uint32_t)__builtin_amdgcn_readlane((int)endk, fw);
                        nvalid = (uint32_t)fw + (__builtin_amdgcn_readlane((int)valid, fw) ? 1u : 0u);
                        next = (uint32_t)fw * wbits + (uint32_t)__builtin_amdgcn_readlane((int)endp, fw);
                    }
                    const uint32_t mine = lane < nvalid ? total : 0u;
                    const uint32_t incl = fl_wave_incl_scan(mine, lane);
                    const uint32_t base = incl - mine;
                    uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), cutwave = 0xffffffffu, cutbudget = 0;
                    const uint64_t over = __ballot(lane < nvalid && base + total > OUT_CAP);
                    if (over) {  // the window is cut at a token of this wave (which adds what fits)
                        const int cw = __builtin_ctzll(over);
                        run = (uint32_t)__builtin_amdgcn_readlane((int)base, cw);
                        cutwave = (uint32_t)cw;
                        cutbudget = OUT_CAP - run;
                        nvalid = (uint32_t)cw + 1u;
                    }
                    if (lane < nvalid) sh->w_base[lane] = base;
                    if (lane == 0) {
                        FP_CNT(52 + kind, 1);
                        FP_CNT(49, nvalid);
                        sh->r_kind = kind;
                        sh->r_next = next;  // window bit where the next round / block starts
                        sh->r_cutwave = cutwave;
                        sh->r_cutbudget = cutbudget;
                        if (kind == FP_X_BAIL) sh->redo = FP_WHY(4);
                        sh->err_far = 0;
                        if (MODE == 1 && to_pool && !pieces_to(wp + OUT_CAP)) sh->redo = FP_WHY(9);
                        sh->r_nvalid = nvalid;
                        sh->r_nout = run;
                        sh->unresolved[0] = 0;
                        sh->unresolved[1] = 0;
                        sh->unresolved[2] = 0;
                    }
                }
                __syncthreads();
                bail = sh->redo != 0;  // (step 6 below does not write it)
                if (bail) break;
                FP_T(37);
                const uint32_t nv2 = sh->r_nvalid, cutwave = sh->r_cutwave;
                // (6) fill: literals and copies from the history are final, the rest points at its source
                if (wave < nv2) {
                    const uint32_t nfix = sh->w_nfix[wave], nown = sh->w_nown[wave], fv = sh->w_fv[wave];
                    const uint32_t n = nfix + nown;
                    const bool cut = wave == cutwave;
                    const uint32_t budget = cut ? sh->r_cutbudget : 0xffffffffu;
                    uint32_t off = sh->w_base[wave], used = 0;
                    for (uint32_t k0 = 0; k0 < n; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        uint32_t t = 0, len = 0;
                        if (k < n) {
                            t = k < nfix ? sh->fixtok[wave][k] : sh->tok[wave][fv + k - nfix];
                            len = fp_tok_len(t);
                        }
                        const uint32_t incl = fl_wave_incl_scan(len, lane);
                        uint64_t okm = __ballot(k < n && used + incl <= budget);
                        uint32_t stop = 64;
                        if (cut) {
                            const uint64_t over = __ballot(k < n && used + incl > budget);
                            if (over) {
                                stop = (uint32_t)__builtin_ctzll(over);
                                okm &= (1ull << stop) - 1ull;
                            }
                        }
                        const uint32_t my_off = off + used + incl - len;
                        const bool ok = (okm >> lane) & 1;
                        if (ok && !(t >> 31)) {
                            if (SYM) {
                                ring16[fp_ring_idx<RING>(wbase, (int32_t)my_off)] = (uint16_t)(t & 0xffu);
                            } else {
                                sh->ring[fp_ring_idx(wbase, (int32_t)my_off)] = (uint8_t)t;
                                sh->ptr[my_off] = (uint16_t)FP_RES;
                            }
                        }
                        // a copied byte points at its source: window position + 32768 (below: the 32 KiB before
                        // the window); short copies are written by their own lanes, long ones by the wave
                        const bool is_m = ok && (t >> 31);
                        const uint32_t mlen_l = (t >> 16) & 0x1ff, mdist_l = (t & 0xffff) + 1;
                        if (is_m && (uint64_t)mdist_l > wp + my_off + hist_avail) sh->err_far = 1;
                        if (MODE == 1 && is_m && (uint64_t)mdist_l > wp + my_off) sh->uses_hist = 1;  // reaches before the start of the output
                        const uint32_t src0 = my_off + 32768u - mdist_l;
                        const uint32_t shortmax = fl_wave_max(is_m && mlen_l <= 32 ? mlen_l : 0u);
                        // (SYM: a source before the window is a symbol that stands: taken at once; one inside it: where it is)
                        auto sym_of = [&](uint32_t s_) -> uint16_t {
                            return s_ < 32768u ? ring16[fp_ring_idx<RING>(wbase, (int32_t)s_ - 32768)] : (uint16_t)(FP_SYM_PTR + (s_ - 32768u));
                        };
                        for (uint32_t i = 0; i < shortmax; i++)
                            if (is_m && mlen_l <= 32 && i < mlen_l) {
                                if (SYM)
                                    ring16[fp_ring_idx<RING>(wbase, (int32_t)(my_off + i))] = sym_of(src0 + i);
                                else
                                    sh->ptr[my_off + i] = (uint16_t)(src0 + i);
                            }
                        uint64_t mm = __ballot(is_m && mlen_l > 32);
                        while (mm) {
                            const uint32_t l0 = (uint32_t)__builtin_ctzll(mm);
                            mm &= mm - 1;
                            const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)mlen_l, (int)l0);
                            const uint32_t mo = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)l0);
                            const uint32_t ms = (uint32_t)__builtin_amdgcn_readlane((int)src0, (int)l0);
                            for (uint32_t i = lane; i < mlen; i += 64) {
                                if (SYM)
                                    ring16[fp_ring_idx<RING>(wbase, (int32_t)(mo + i))] = sym_of(ms + i);
                                else
                                    sh->ptr[mo + i] = (uint16_t)(ms + i);
                            }
                        }
                        if (stop < 64) {
                            // the token at k0 + stop does not fit: the next round starts at its bit
                            const uint32_t kc = k0 + stop;
                            const uint32_t fit = stop ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(stop - 1)) : 0u;
                            uint32_t pos;
                            if (kc < nfix) {
                                pos = sh->fixpos[wave][kc];
                            } else {
                                // own token number fv + kc - nfix = that set bit of the wave's map
                                uint32_t want = fv + kc - nfix, wd = 0;
                                while (wd < FP_WBITS / 32 && want >= (uint32_t)__popc(sh->bitmap[wave][wd])) {
                                    want -= (uint32_t)__popc(sh->bitmap[wave][wd]);
                                    wd++;
                                }
                                uint32_t bitsw = wd < FP_WBITS / 32 ? sh->bitmap[wave][wd] : 0u;
                                for (uint32_t q = 0; q < want; q++) bitsw &= bitsw - 1;
                                pos = 32 * wd + (bitsw ? (uint32_t)__builtin_ctz(bitsw) : 0u);
                            }
                            if (lane == 0) {
                                sh->r_kind = FP_X_CUT;
                                sh->r_next = wave * wbits + pos;
                                sh->r_nout = off + used + fit;
                            }
                            break;
                        }
                        used += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    }
                }
                __syncthreads();
                FP_T(38);
                const uint32_t nout = sh->r_nout;
                bail = sh->err_far != 0 || wp + nout > out_room;  // InvalidMatch / OutputTooSmall are k_inflate's to report
                if (bail) break;
                FP_CNT(50, nout);
                // (7) resolve the copies inside the window by pointer jumping over byte positions
                // (a thread's positions tid + 1024 k that are not final yet: one bit each, so that the later rounds look
                // at those only)
                uint32_t pend = nout > tid ? (1u << ((nout - tid + FP_THREADS - 1) / FP_THREADS)) - 1u : 0u;
                for (uint32_t rr = 0;;) {
                    // (several sweeps between two barriers: a position follows its chain on what the other threads have written
                    // so far -- any order is right, a byte is written before its flag and read after it)
                    for (int sweep = 0; sweep < FP_RESOLVE_SWEEPS && (sweep == 0 || pend != 0); sweep++) {
                    uint32_t todo = pend;
                    pend = 0;
                    while (todo) {
                        const uint32_t kb = (uint32_t)__builtin_ctz(todo);
                        todo &= todo - 1;
                        const uint32_t j = tid + kb * FP_THREADS;
                        if (SYM) {
                            // the source's symbol, or -- if that is not resolved either -- the source's source: one 16-bit word
                            const uint32_t xi = fp_ring_idx<RING>(wbase, (int32_t)j);
                            const uint32_t v = ring16[xi];
                            if (v >= FP_SYM_PTR) {
                                const uint32_t pv = ring16[fp_ring_idx<RING>(wbase, (int32_t)(v - FP_SYM_PTR))];
                                ring16[xi] = (uint16_t)pv;
                                if (pv >= FP_SYM_PTR) pend |= 1u << kb;
                            }
                            continue;
                        }
                        const uint32_t v = sh->ptr[j];
                        if (v != FP_RES) {
                            if (v < 32768u) {  // the source lies before the window: final
                                sh->ring[fp_ring_idx(wbase, (int32_t)j)] = sh->ring[fp_ring_idx(wbase, (int32_t)v - 32768)];
                                asm volatile("" ::: "memory");  // the flag is written after the byte
                                sh->ptr[j] = (uint16_t)FP_RES;
                            } else {
                                const uint32_t pv = sh->ptr[v - 32768u];
                                asm volatile("" ::: "memory");  // the source byte is read after its flag
                                if (pv == FP_RES) {
                                    sh->ring[fp_ring_idx(wbase, (int32_t)j)] = sh->ring[fp_ring_idx(wbase, (int32_t)v - 32768)];
                                    asm volatile("" ::: "memory");
                                    sh->ptr[j] = (uint16_t)FP_RES;
                                } else {
                                    sh->ptr[j] = (uint16_t)pv;  // the source's source
                                    pend |= 1u << kb;
                                }
                            }
                        }
                    }
                    }
                    const bool mine = pend != 0;
                    FP_CNT(51, 1);
                    // round r's flag is read after this barrier by everyone and cleared two rounds later,
                    // before the barrier of round r + 2: every wave has read it by then
                    if (__any(mine) && lane == 0) sh->unresolved[rr] = 1;
                    if (tid == 0) sh->unresolved[rr == 2 ? 0 : rr + 1] = 0;
                    __syncthreads();
                    if (!sh->unresolved[rr]) break;
                    rr = rr == 2 ? 0 : rr + 1;
                }
                FP_T(39);
                // (8) the window's bytes leave
                if (SYM) {
                    for (uint32_t j = tid; j < nout; j += FP_THREADS) {
                        const uint32_t e = ring16[fp_ring_idx<RING>(wbase, (int32_t)j)];
                        *out_at(wp + j) = (uint8_t)e;
                        *out_b_at(wp + j) = (uint8_t)(e ^ (e >> 8));
                    }
                } else {
                    for (uint32_t j = tid; j < nout; j += FP_THREADS) *out_at(wp + j) = sh->ring[fp_ring_idx(wbase, (int32_t)j)];
                }
                __syncthreads();
                if (tid == 0) {
                    const uint64_t nb = bitpos + sh->r_next;  // r_next counts from the window's first bit
                    if (nb > total_bits || (nb <= bitpos && sh->r_kind != FP_X_EOB)) sh->redo = FP_WHY(7);  // past the end / no progress
                    sh->bitpos = nb;
                    sh->wp = wp + nout;
                    sh->blk_done = sh->r_kind == FP_X_EOB;
                    if (sh->r_kind == FP_X_FULL && wbits > 512) sh->wbits = wbits >> 1;  // 512 bits hold at most 512 tokens
#if FP_ADAPT_WBITS
                    // A round that was CUT (its output was full) decoded bits it had no room for: the next rounds take what this
                    // one used and an eighth (data that compresses well: records, markup, runs); a round that used all its bits
                    // grows by a quarter again.
                    else if (sh->r_kind == FP_X_CUT) {
                        const uint32_t want = (((sh->r_next + (sh->r_next >> 3)) / FP_WAVES) + 63u) & ~63u;
                        sh->wbits = min(wbits, max(512u, want));
                    } else if (sh->r_kind == FP_X_NORMAL && wbits < (SYM ? FP_SYM_WBITS : (uint32_t)FP_WBITS)) {
                        sh->wbits = min((SYM ? FP_SYM_WBITS : (uint32_t)FP_WBITS), ((wbits + (wbits >> 2)) + 63u) & ~63u);
                    }
#endif
                }
                __syncthreads();
                FP_T(40);
                bail = sh->redo != 0;
                if (bail || sh->blk_done) break;  // (the next writes of either flag come behind another barrier)
            }
            if (bail) break;
        }
        if (blk_final) {
            final_seen = true;
            break;
        }
    }
    __syncthreads();
    if (bail) {
        if (tid == 0) {
            if (MODE == 0)
                status[c] = FL_PAR_REDO;
            else {
                sres[unit].status = sh->redo ? sh->redo : 1u;
                sres[unit].end_bit = sh->bitpos;  // (where it gave up: FLATE_HIP_SPAN_DEBUG prints it)
                sres[unit].out_len = sh->wp;
                sres[unit].n_pieces = sh->nblk;
            }
        }
        return;
    }

    // ================================================================ footer (container.zig:154-166)
    const uint64_t n_out = sh->wp;
    if (MODE != 0) {
        // the span's tail: the last 32 KiB of the output up to its end (what was there before it included)
        const uint32_t wb = (uint32_t)(n_out % RING);
        uint8_t* tl = tails + (uint64_t)unit * FP_TAIL;
        if (SYM) {
            uint8_t* tlb = tails_b + (uint64_t)unit * FP_TAIL;
            for (uint32_t i = tid; i < FP_TAIL; i += FP_THREADS) {
                const uint32_t e = ring16[fp_ring_idx<RING>(wb, (int32_t)i - (int32_t)FP_TAIL)];
                tl[i] = (uint8_t)e;
                tlb[i] = (uint8_t)(e ^ (e >> 8));
            }
        } else {
            for (uint32_t i = tid; i < FP_TAIL; i += FP_THREADS) tl[i] = sh->ring[fp_ring_idx(wb, (int32_t)i - (int32_t)FP_TAIL)];
        }
        if (tid == 0) {
            fl_span_res* rr = &sres[unit];
            rr->end_bit = sh->bitpos;
            rr->out_len = n_out;
            rr->final_seen = final_seen ? 1u : 0u;
            rr->uses_hist = sh->uses_hist;
            rr->n_pieces = sh->n_pieces;
            rr->status = 0;
        }
        return;  // (checksums: k_span_fix; the footer: the host)
    }
    if (container != 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        FL_LDS uint32_t* tab = (FL_LDS uint32_t*)ws->lit_lut;  // the code tables are no longer needed
        if (container == 1) {
            for (uint32_t t = tid; t < 256; t += FP_THREADS) {
                uint32_t v = t;
                for (int k = 0; k < 8; k++) v = (v & 1) ? (FL_CRC_POLY ^ (v >> 1)) : (v >> 1);
                tab[t] = v;
            }
        }
        if (tid == 0) {
            sh->crc = 0;
            sh->adA = 0;
            sh->adB = 0;
        }
        __syncthreads();
        const uint64_t per = (n_out + FP_THREADS - 1) / FP_THREADS;
        const uint64_t lo = min(n_out, (uint64_t)tid * per), hi = min(n_out, lo + per);
        if (container == 1) {
            uint32_t v = 0xffffffffu;
            for (uint64_t i = lo; i < hi; i++) v = tab[(v ^ dst[i]) & 0xff] ^ (v >> 8);
            v = hi > lo ? ~v : 0u;
            v = fl_crc_mulmod(v, fl_crc_xpow8n(cc.xpow8, n_out - hi));
            v = fl_wave_xor(v);
            if (lane == 0) atomicXor((uint32_t*)&sh->crc, v);
        } else {
            uint32_t A = 0, B = 0;
            uint64_t i = lo;
            while (i < hi) {
                const uint64_t e = min(hi, i + 5552);
                for (; i < e; i++) {
                    A += dst[i];
                    B += A;
                }
                A %= 65521u;
                B %= 65521u;
            }
            const uint64_t after = (n_out - hi) % 65521u;
            uint32_t Bm = (uint32_t)((B + (uint64_t)A * after) % 65521u);
            const uint32_t As = fl_wave_sum(A), Bs = fl_wave_sum(Bm);  // 64 values < 65521
            if (lane == 0) {
                atomicAdd((uint32_t*)&sh->adA, As);  // 16 partial sums < 2^22: no overflow
                atomicAdd((uint32_t*)&sh->adB, Bs);
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        reader_at(sh->bitpos);
        fl_br_align(r);
        int rc = 0;
        uint32_t v = 0;
        if (container == 1) {
            rc = fl_br_read(r, 32, v);
            if (!rc && v != sh->crc) rc = 4;
            if (!rc) rc = fl_br_read(r, 32, v);
            if (!rc && v != (uint32_t)n_out) rc = 5;
        } else if (container == 2) {
            const uint32_t a = (1 + sh->adA % 65521u) % 65521u;
            const uint32_t b = (uint32_t)((n_out % 65521u + sh->adB % 65521u) % 65521u);
            rc = fl_br_read(r, 32, v);
            if (!rc && v != __builtin_bswap32(a | (b << 16))) rc = 6;
        }
        if (lane == 0) {
            if (rc) {
                status[c] = FL_PAR_REDO;
            } else {
                out_len[c] = n_out;
                status[c] = 0;
                if (consumed) consumed[c] = fl_br_consumed(r);
            }
        }
    }
}

// ------------------------------------------------------------------ k_span_scan
// Where can a span start?  At the first bit position at or behind a target position at which a dynamic block
// header parses (inflate.zig:137-218: type bits 10, HLIT / HDIST in range, a complete code-length code, code
// lengths that decode without overrun into a complete literal / length code with an end-of-block symbol and a
// usable distance code), or at a stored block that follows a stored block.  One workgroup per target, windows of 128 Kibit:
//   1. every lane tests bit positions against what costs a few instructions (type, HLIT, HDIST, the Kraft sum of
//      the code-length code): about 1 in 250 survives;
//   2. every survivor gets a LANE that decodes its code lengths on its own (canonical code-length code held in
//      registers, bits straight from memory) and checks the sums;
//   3. wave 0 runs the real header parser (fl_inf_dynamic_header) on what is left, lowest position first.
// Steps 1 and 2 may refuse a header the parser would take (the span before just gets longer) but what they pass
// is only a candidate until step 3 agrees.  A position that parses by accident inside compressed data is
// harmless too: no span lands on it, its span is dead.
struct fl_scan_point {
    uint64_t from_bit, limit_bit;  // search [from_bit, limit_bit)
    uint32_t stream, pad;
};
#ifndef FP_SCAN_WIN_BITS
#define FP_SCAN_WIN_BITS 131072u  // (64 / 128 / 256 Kibit: 170 MiB of text 1.04 / 0.73 / 0.82 ms, config #4's stream 2.5 / 1.85 / 1.4: step 2 costs its slowest lane once per window)
#endif
#define FP_SCAN_STAGE_DW (FP_SCAN_WIN_BITS / 32 + 8 + 80)  // + what a header at the end of the window reaches into (at most 17 + 57 + 316 * 14 bits ... the lane gives up beyond the stage)
#define FP_SCAN_CAP (FP_SCAN_WIN_BITS / 32u)
#define FP_SCAN_LANES 256u  // lanes that validate at a time (128 bytes of LDS each)
struct fp_scan_shared {
    fl_inflate_ws ws;
    uint32_t inring[FL_INF_INRING / 4];
    uint32_t stage[FP_SCAN_STAGE_DW];
    uint32_t surv[FP_SCAN_CAP];  // window bit offsets of the survivors of step 1 (any order)
    uint8_t vtab[FP_SCAN_LANES][128];  // step 2: a lookup table of the code-length code per validating lane
    uint32_t nsurv, npass;
    uint32_t passed[64];         // ... of step 2
    uint32_t found;
    uint32_t st_found;           // lowest window bit of a stored block's header (0xffffffff: none)
};

// 64 bits of the stream from bit position `bit` (zero beyond the end)
__device__ __forceinline__ uint64_t fp_bits_at(const uint8_t* src, uint32_t in_len, uint64_t bit) {
    const uint32_t by = (uint32_t)(bit >> 3), k = (uint32_t)bit & 7;
    const uint64_t lo = fl_load_u32_clamped(src, by, in_len), mid = fl_load_u32_clamped(src, by + 4, in_len);
    const uint64_t hi = fl_load_u32_clamped(src, by + 8, in_len);
    const uint64_t v = lo | (mid << 32);
    return k ? (v >> k) | (hi << (64 - k)) : v;
}

// Step 2 for one position, by one lane.  `tab`: the lane's 128 bytes of LDS -- the code-length code as a lookup table by
// the next 7 stream bits: symbol << 3 | code bits, 0 = no code.
__device__ bool fp_scan_header_lane(const FL_LDS uint32_t* stage, uint32_t rel0, uint32_t in_len, uint64_t bit, FL_LDS uint8_t* tab) {
    // rel0: position of `bit` in the staged window (bits); the stage ends at 32 * FP_SCAN_STAGE_DW - 64
    const uint64_t total_bits = (uint64_t)in_len * 8;
    auto bits_at = [&](uint64_t at) -> uint64_t {
        uint32_t lo, hi;
        fp_fetch64(stage, rel0 + (uint32_t)(at - bit), lo, hi);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    };
    const uint32_t rel_end = 32u * FP_SCAN_STAGE_DW - 64u - 64u;
    uint64_t b = bits_at(bit);
    const uint32_t hlit = (uint32_t)(b >> 3) & 31, hdist = (uint32_t)(b >> 8) & 31, ncl = ((uint32_t)(b >> 13) & 15) + 4;
    const uint32_t nlit = hlit + 257, ntot = nlit + hdist + 1;
    uint64_t pos = bit + 17;
    // code lengths of the code-length alphabet (3 bits each, permuted order), packed by symbol
    uint64_t clen = 0;
    {
        const uint64_t b2 = bits_at(pos);
        // order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const uint64_t order = 0x10ull | (0x11ull << 5) | (0x12ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) |
                               (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
        const uint64_t order2 = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
        for (uint32_t i = 0; i < ncl; i++) {
            const uint32_t sym = i < 12 ? (uint32_t)(order >> (5 * i)) & 31 : (uint32_t)(order2 >> (5 * (i - 12))) & 31;
            const uint64_t l = (b2 >> (3 * i)) & 7;  // (3 * 18 + 3 = 57 bits)
            clen |= l << (3 * sym);
        }
        pos += 3 * ncl;
    }
    // the canonical code (huffman_decoder.zig:62-117: codes of one length are consecutive, shorter ones first) as a
    // table by the next 7 bits of the stream, first bit = most significant bit of the code
    {
        uint32_t cnt8 = 0, nz = 0;  // codes per length, 4 bits each (at most 19 ... fits: a complete code has at most 2 of length 1)
        uint32_t count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t sy = 0; sy < 19; sy++) {
            const uint32_t l = (uint32_t)(clen >> (3 * sy)) & 7;
#pragma unroll
            for (uint32_t k = 1; k < 8; k++) count[k] += l == k ? 1u : 0u;
            nz += l ? 1u : 0u;
        }
        (void)cnt8;
        if (nz <= 1) return false;  // (a code-length code of one symbol: legal, never seen; the span before gets longer)
        uint32_t next[8];
        uint32_t code = 0;
        next[0] = 0;
#pragma unroll
        for (uint32_t k = 1; k < 8; k++) {
            code = (code + count[k - 1]) << 1;
            next[k] = code;
        }
        for (uint32_t i = 0; i < 32; i++) ((FL_LDS uint32_t*)tab)[i] = 0;
        for (uint32_t sy = 0; sy < 19; sy++) {
            const uint32_t l = (uint32_t)(clen >> (3 * sy)) & 7;
            if (!l) continue;
            uint32_t c = 0;
#pragma unroll
            for (uint32_t k = 1; k < 8; k++)
                if (l == k) {
                    c = next[k];
                    next[k] = c + 1;
                }
            if (c >> l) return false;  // oversubscribed
            const uint32_t rev = __brev(c) >> (32 - l);
            for (uint32_t j = rev; j < 128; j += 1u << l) tab[j] = (uint8_t)((sy << 3) | l);
        }
    }
    uint32_t kl = 0, kd = 0, nd = 0, prev = 0, eob = 0;  // Kraft sums (in units of 2^-15), distance codes, last length
    uint32_t i = 0;
    // One symbol per turn, without branches on the symbol (the lanes of a wave are at different symbols: every branch
    // taken by one lane is paid by all; with them a turn cost 850 cycles, most of step 2 of the scan).
    while (i < ntot) {
        if (pos >= total_bits || rel0 + (uint32_t)(pos - bit) > rel_end) return false;  // (beyond the stage: given up)
        b = bits_at(pos);
        const uint32_t e = tab[(uint32_t)b & 127u];
        const uint32_t used = e & 7, sym = e >> 3;
        if (!used) return false;
        const bool is16 = sym == 16, is17 = sym == 17, is18 = sym == 18;
        const uint32_t xb = is16 ? 2u : is17 ? 3u : is18 ? 7u : 0u;
        const uint32_t xv = ((uint32_t)(b >> used)) & ((1u << xb) - 1u);
        const uint32_t rep = is18 ? 11u + xv : (is16 || is17) ? 3u + xv : 1u;
        const uint32_t len = sym < 16 ? sym : is16 ? prev : 0u;
        if (is16 && i == 0) return false;
        pos += used + xb;
        if (i + rep > ntot) return false;
        // (a run may cross from the literal / length lengths into the distance lengths)
        const uint32_t nl = i < nlit ? min(rep, nlit - i) : 0u;
        const uint32_t w = len ? (32768u >> len) : 0u;
        kl += nl * w;
        kd += (rep - nl) * w;
        nd += len ? rep - nl : 0u;
        eob |= (len && i <= 256 && i + rep > 256) ? 1u : 0u;
        // (an oversubscribed code cannot become complete again: what is not a header is given up here, after a few
        // dozen symbols, long before its lengths have all been decoded)
        if (kl > 32768u || kd > 32768u) return false;
        prev = len;
        i += rep;
    }
    if (pos > total_bits) return false;
    return eob && kl == 32768u && (kd == 32768u || nd <= 1);
}

__global__ __launch_bounds__(FP_THREADS) void k_span_scan(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks,
                                                          int flags, const fl_scan_point* __restrict__ points,
                                                          uint64_t* __restrict__ found_out) {
    __shared__ fp_scan_shared sm;
    FL_LDS fp_scan_shared* sh = (FL_LDS fp_scan_shared*)&sm;
    FL_LDS fl_inflate_ws* ws = &sh->ws;
    const fl_scan_point pt = points[blockIdx.x];
    const fl_chunk ck = chunks[pt.stream];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* src = in + ck.in_off;
    const uint64_t total_bits = (uint64_t)ck.in_len * 8;
    fl_bitr r;  // wave 0 only
    r.data = src;
    r.nbytes = ck.in_len;
    r.lane = lane;
    r.inring = (FL_LDS uint32_t*)sh->inring;
    r.left = (int64_t)total_bits;
    uint64_t found_bit = ~0ull;
    // ---- a target inside a stored block (store-only streams, incompressible stretches): nothing parses there, and
    // the windows below would take their time to find that out.  The first stored header behind the target that is
    // followed by another one (at most 65540 bytes on), then the stored header before it whose LEN ends there: if that
    // block holds the target, the header behind it is the place.
    {
        auto stored_at = [&](uint64_t q, uint32_t low_mask) -> uint32_t {  // LEN of a stored header at byte q, or ~0u
            if (q + 5 > ck.in_len) return 0xffffffffu;
            const uint32_t w0 = fl_load_u32_clamped(src, (uint32_t)q, ck.in_len), w1 = fl_load_u32_clamped(src, (uint32_t)q + 4, ck.in_len);
            const uint32_t len = (w0 >> 8) & 0xffff, nlen = (w0 >> 24) | ((w1 & 0xff) << 8);
            return ((w0 & low_mask) == 0 && (len ^ nlen) == 0xffff) ? len : 0xffffffffu;
        };
        if (tid == 0) {
            sh->st_found = 0xffffffffu;
            sh->found = 0;
        }
        __syncthreads();
        const uint64_t q0 = (pt.from_bit + 7) >> 3, q1 = min((pt.limit_bit + 7) >> 3, q0 + 65541u);
        for (uint64_t q = q0 + tid; q < q1; q += FP_THREADS) {
            const uint32_t len = stored_at(q, 7u);
            if (len != 0xffffffffu && stored_at(q + 5 + len, 6u) != 0xffffffffu) atomicMin(&sm.st_found, (uint32_t)(q - q0));
        }
        __syncthreads();
        const uint32_t sf = sh->st_found;
        if (sf != 0xffffffffu) {
            const uint64_t s = q0 + sf;
            for (uint32_t len = tid; len < 65536u && len + 5 <= s; len += FP_THREADS) {
                const uint64_t q = s - 5 - len;
                if (q * 8 + 3 <= pt.from_bit + 7 && stored_at(q, 6u) == len) sh->found = 1;  // (header bits at or before the target)
            }
        }
        __syncthreads();
        const bool inside = sf != 0xffffffffu && sh->found != 0;
        __syncthreads();
        if (inside) {
            if (tid == 0) found_out[blockIdx.x] = (q0 + sf) * 8;
            return;
        }
    }
    for (uint64_t base = pt.from_bit; base < pt.limit_bit; base += FP_SCAN_WIN_BITS) {
        const uint32_t byte0 = (uint32_t)(base >> 3), bsh = (uint32_t)base & 7;
        for (uint32_t i = tid; i < FP_SCAN_STAGE_DW; i += FP_THREADS) sh->stage[i] = fl_load_u32_clamped(src, byte0 + 4 * i, ck.in_len);
        if (tid == 0) {
            sh->nsurv = 0;
            sh->npass = 0;
            sh->found = 0;
            sh->st_found = 0xffffffffu;
        }
        __syncthreads();
#ifdef FP_SCAN_PROF
        uint64_t tp_ = __builtin_readcyclecounter();
#define FP_SCAN_T(slot) do { const uint64_t n_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd((unsigned long long*)&g_fl_prof[slot], (unsigned long long)(n_ - tp_)); tp_ = n_; } while (0)
#else
#define FP_SCAN_T(slot)
#endif
        // ---- 0: stored blocks in a row.  Behind a stored block the next header starts a byte: BFINAL 0, BTYPE 00 in
        // its low three bits, then LEN and its complement (inflate.zig:89-102); taken when the block behind it starts
        // the same way (one in 2^38 bytes does by accident).  A run of stored blocks has no other place to cut at.
        {
            const uint32_t first_byte = (uint32_t)((base + 7) >> 3);  // the window's byte positions
            for (uint32_t j = tid; j < FP_SCAN_WIN_BITS / 8; j += FP_THREADS) {
                const uint32_t q = first_byte + j;
                const uint64_t bit = (uint64_t)q * 8;
                if (bit >= pt.limit_bit || (uint64_t)q + 5 > ck.in_len) break;
                const uint32_t rel = q - byte0;  // in the stage
                const uint32_t d0 = sh->stage[rel >> 2], d1 = sh->stage[(rel >> 2) + 1], d2 = sh->stage[(rel >> 2) + 2];
                const uint32_t sft = (rel & 3) * 8;
                const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sft), w1 = __builtin_amdgcn_alignbit(d2, d1, sft);
                const uint32_t len = (w0 >> 8) & 0xffff, nlen = (w0 >> 24) | ((w1 & 0xff) << 8);
                if ((w0 & 7) == 0 && (len ^ nlen) == 0xffff) {
                    const uint64_t nq = (uint64_t)q + 5 + len;  // the header behind it
                    if (nq + 5 <= ck.in_len) {
                        const uint32_t n0 = src[nq], nl = (uint32_t)src[nq + 1] | ((uint32_t)src[nq + 2] << 8);
                        const uint32_t nn = (uint32_t)src[nq + 3] | ((uint32_t)src[nq + 4] << 8);
                        if ((n0 & 6) == 0 && (nl ^ nn) == 0xffff) atomicMin(&sm.st_found, (uint32_t)(bit - base));
                    }
                }
            }
        }
        FP_SCAN_T(13);
        // ---- 1: first what costs a handful of instructions (type bits, HLIT, HDIST: 1 position in 4.5 passes), for all
        // 64 positions of the lane; then the Kraft sum of the code-length code for those that passed
        for (uint32_t g = 0; g < FP_SCAN_WIN_BITS / FP_THREADS; g += 64) {
        uint64_t cheap = 0;
#pragma unroll 4
        for (uint32_t jj = 0; jj < 64; jj++) {
            const uint32_t wb = (g + jj) * FP_THREADS + tid;
            const uint64_t bit = base + wb;
            const uint32_t di = (bsh + wb) >> 5, shf = (bsh + wb) & 31;
            const uint32_t lo = __builtin_amdgcn_alignbit(sh->stage[di + 1], sh->stage[di], shf);  // 32 bits from that position
            const uint32_t btype = (lo >> 1) & 3, hlit = (lo >> 3) & 31, hdist = (lo >> 8) & 31;
            if (btype == 2 && hlit <= 29 && hdist <= 29 && bit < pt.limit_bit && bit + 17 + 12 <= total_bits) cheap |= 1ull << jj;
        }
        while (cheap) {
            const uint32_t jj = (uint32_t)__builtin_ctzll(cheap);
            cheap &= cheap - 1;
            const uint32_t wb = (g + jj) * FP_THREADS + tid;
            const uint64_t bit = base + wb;
            uint32_t lo, hi, lo2, hi2;
            fp_fetch64(sh->stage, bsh + wb, lo, hi);
            fp_fetch64(sh->stage, bsh + wb + 62, lo2, hi2);
            const uint32_t ncl = ((lo >> 13) & 15) + 4;
            const uint64_t b64 = (uint64_t)lo | ((uint64_t)hi << 32);
            uint32_t kraft = 0, nz = 0;
#pragma unroll
            for (uint32_t i = 0; i < 19; i++) {
                const uint32_t len = i < 15 ? (uint32_t)(b64 >> (17 + 3 * i)) & 7u : (lo2 >> (3 * i - 45)) & 7u;
                if (i < ncl && len) {
                    kraft += 128u >> len;
                    nz++;
                }
            }
            if ((kraft == 128 || nz <= 1) && bit + 17 + 3 * ncl <= total_bits) {
                const uint32_t k = atomicAdd(&sm.nsurv, 1u);
                if (k < FP_SCAN_CAP) sh->surv[k] = wb;
            }
        }
        }
        __syncthreads();
        FP_SCAN_T(14);
        // ---- 2
        const uint32_t ns = min(sh->nsurv, FP_SCAN_CAP);
#ifdef FP_SCAN_PROF
        if (tid == 0) { atomicAdd((unsigned long long*)&g_fl_prof[10], 1ull); atomicAdd((unsigned long long*)&g_fl_prof[11], (unsigned long long)sh->nsurv); }
#endif
        for (uint32_t k = tid; tid < FP_SCAN_LANES && k < ns; k += FP_SCAN_LANES) {
            const uint32_t wb = sh->surv[k];
            if (fp_scan_header_lane(sh->stage, bsh + wb, ck.in_len, base + wb, (FL_LDS uint8_t*)sh->vtab[tid])) {
                const uint32_t q = atomicAdd(&sm.npass, 1u);
                if (q < 64) sh->passed[q] = wb;
            }
        }
        __syncthreads();
        FP_SCAN_T(15);
        // ---- 3
#ifdef FP_SCAN_PROF
        if (tid == 0) atomicAdd((unsigned long long*)&g_fl_prof[12], (unsigned long long)sh->npass);
#endif
        if (wave == 0) {
            const uint32_t np = min(sh->npass, 64u);
            uint32_t mine = lane < np ? sh->passed[lane] : 0xffffffffu;
            for (uint32_t round = 0; round < np; round++) {
                // the lowest position not tried yet
                uint32_t lowest = mine;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) lowest = min(lowest, (uint32_t)__shfl_xor((int)lowest, d, 64));
                if (lowest == 0xffffffffu) break;
                if (mine == lowest) mine = 0xffffffffu;
                const uint64_t cb = base + lowest;
                r.left = (int64_t)(total_bits - cb);  // the reader at that bit (as in fp_body)
                fl_br_seek(r, (uint32_t)(cb >> 3));
                const uint32_t kb = (uint32_t)cb & 7;
                if (kb) {
                    fl_br_refill(r);
                    r.buf >>= kb;
                    r.have -= kb;
                }
                uint32_t bfinal = 0, btype = 3;
                int rc = fl_br_read(r, 1, bfinal);
                if (!rc) rc = fl_br_read(r, 2, btype);
                if (!rc && btype == 2) rc = fl_inf_dynamic_header(r, ws, flags, lane);
                if (!rc && btype == 2) {
                    if (lane == 0) sh->found = lowest + 1u;
                    break;
                }
            }
        }
        __syncthreads();
        FP_SCAN_T(16);
        {
            const uint32_t f = min(sh->found ? sh->found - 1u : 0xffffffffu, sh->st_found);
            if (f != 0xffffffffu) {
                found_bit = base + f;
                break;
            }
        }
        __syncthreads();  // (the lists are rewritten by the next window)
    }
    if (tid == 0) found_out[blockIdx.x] = found_bit;
}

// One workgroup per stream.  `min_bytes`: shorter streams are left to k_inflate.
__global__ __launch_bounds__(FP_THREADS, 1) void k_inflate_par(const uint8_t* __restrict__ in,
                                                               const fl_chunk* __restrict__ chunks, int container,
                                                               int flags, uint32_t min_bytes, fl_crc_consts cc,
                                                               uint8_t* __restrict__ out, uint64_t* __restrict__ out_len,
                                                               int32_t* __restrict__ status,
                                                               uint64_t* __restrict__ consumed) {
    __shared__ fp_shared sh_mem;
    fp_body<0>(in, chunks, container, flags, min_bytes, cc, out, out_len, status, consumed, nullptr, nullptr, nullptr,
               nullptr, nullptr, 0u, fl_span_pool{nullptr, nullptr, nullptr, 0u, 0u}, blockIdx.x, false, (FL_LDS fp_shared*)&sh_mem);
}

// One workgroup per span (see above); fill 0: run A, 1: run B.
__global__ __launch_bounds__(FP_THREADS, 1) void k_inflate_span(const uint8_t* __restrict__ in,
                                                                const fl_chunk* __restrict__ chunks, int container,
                                                                int flags, fl_crc_consts cc, uint8_t* __restrict__ out,
                                                                const fl_span* __restrict__ spans,
                                                                fl_span_res* __restrict__ sres,
                                                                const uint64_t* __restrict__ cand,
                                                                const uint32_t* __restrict__ cand_off,
                                                                uint8_t* __restrict__ tails, uint32_t fill,
                                                                fl_span_pool pool, uint32_t twin_nsp,
                                                                uint8_t* __restrict__ tails_b, uint32_t sym_nsp) {
    // twin_nsp != 0: both runs in one launch -- workgroups [0, twin_nsp) are run A of the spans, [twin_nsp, 2 twin_nsp)
    // run B of the same spans, with its results, tails and pieces behind run A's
    uint32_t unit = blockIdx.x;
    bool b_pool = false;
    if (twin_nsp && unit >= twin_nsp) {
        unit -= twin_nsp;
        fill = 1;
        b_pool = true;
        sres += twin_nsp;
        tails = tails_b;
        pool.tab += (uint64_t)twin_nsp * FP_MAX_PIECES;
    }
    __shared__ fp_shared sh_mem;
    if (sym_nsp) {
        // ONE decode per span (round 5): a stream's first span writes its bytes in place, every other one symbols -- both planes
        // (what runs A and B made) to the pool, plane B's pieces and results where a twin launch's run B had them
        if (spans[unit].first)
            fp_body<1, false>(in, chunks, container, flags, 0u, cc, out, nullptr, nullptr, nullptr, spans, sres, cand, cand_off, tails, 0u,
                              pool, unit, false, (FL_LDS fp_shared*)&sh_mem);
        else
            fp_body<1, true>(in, chunks, container, flags, 0u, cc, out, nullptr, nullptr, nullptr, spans, sres, cand, cand_off, tails, 0u,
                             pool, unit, false, (FL_LDS fp_shared*)&sh_mem, sym_nsp, tails_b);
        return;
    }
    fp_body<1>(in, chunks, container, flags, 0u, cc, out, nullptr, nullptr, nullptr, spans, sres, cand, cand_off, tails, fill,
               pool, unit, b_pool, (FL_LDS fp_shared*)&sh_mem);
}

// The bytes of the live spans, 64 KiB (an ITEM) per wave, a lane per 1024 of them: moved from the pool to their place
// where run A could not know it, made true where they depend on the history (a != b: byte ((a ^ b) - 1) << 8 | a of
// the true tail before the span), and their CRC-32 / Adler-32 piece (folded by the host in stream order).
//   kind 0  a first span: the bytes are in place and final
//   kind 1  no span of the batch copies from before its start (run B was skipped): run A's bytes are final
//   kind 2  run A's bytes in the pool, run B's in place
//   kind 3  both in the pool (the runs were one launch: pad = the number of spans, run B's pieces follow run A's)
struct __attribute__((aligned(4))) fl_u4a {  // four words at a word-aligned address
    uint32_t x[4];
};
#define FP_FIX_LANE 1024u                 // bytes per lane
#define FP_FIX_ITEM (64u * FP_FIX_LANE)   // ... and item (a wave): a piece of the pool (16 KiB items, 256 bytes a lane: 0.84 -> 1.21 ms)
#define FP_FIX_WAVES 4u                   // items per workgroup: of ONE span, whose history (the true tail before it) they share in LDS
struct fl_fix_item {
    uint64_t dst;    // of the item's first byte in the output buffer
    uint32_t span;   // whose pieces
    uint32_t local;  // offset of the item in the span's output (a multiple of FP_FIX_ITEM)
    uint32_t len;    // <= FP_FIX_ITEM
    uint32_t kind;
    uint32_t prev;   // kind 2: the span before it in the chain
    uint32_t pad;
};
__global__ __launch_bounds__(64 * FP_FIX_WAVES) void k_span_fix(const fl_fix_item* __restrict__ items, fl_span_pool pool,
                                                 const uint8_t* __restrict__ tails, uint8_t* out, int container,
                                                 fl_crc_consts cc, uint32_t* __restrict__ part /* [n_items][2] */) {
    __shared__ uint32_t tab[4][256];
    // the history of the workgroup's items (all of one span): a byte that depends on it is a lookup here, not a gather from L2 --
    // in text half of all bytes are such, a cache line each from L2 (the tails of the spans a CU works on do not fit its L1)
    __shared__ uint32_t T32[FP_TAIL / 4];
    const uint32_t item_no = blockIdx.x * FP_FIX_WAVES + (threadIdx.x >> 6);
    const fl_fix_item it = items[item_no];
    const uint32_t lane = threadIdx.x & 63u;
    if (container == 1) {
        const uint32_t t = threadIdx.x;
        {
            uint32_t c = t;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (FL_CRC_POLY ^ (c >> 1)) : (c >> 1);
            tab[0][t] = c;
        }
        __syncthreads();
        for (int k = 1; k < 4; k++) {
            const uint32_t c = tab[k - 1][t];
            tab[k][t] = tab[0][c & 0xff] ^ (c >> 8);
            __syncthreads();
        }
    }
    {
        const fl_fix_item it0 = items[blockIdx.x * FP_FIX_WAVES];  // (the host pads a span's items to whole workgroups)
        if (it0.kind >= 2) {
            const uint4* tg = (const uint4*)(tails + (uint64_t)it0.prev * FP_TAIL);
            for (uint32_t i = threadIdx.x; i < FP_TAIL / 16; i += 64 * FP_FIX_WAVES) ((uint4*)T32)[i] = tg[i];
        }
        __syncthreads();
    }
    const uint8_t* T = (const uint8_t*)T32;
    const uint32_t lo = min(it.len, lane * FP_FIX_LANE), hi = min(it.len, lane * FP_FIX_LANE + FP_FIX_LANE);
    uint8_t* D = out + it.dst;
    const uint32_t in_piece = it.local & (FP_PIECE - 1u);  // (an item is a part of a piece)
    const uint8_t* A = it.kind ? pool.base + (uint64_t)pool.tab[(uint64_t)it.span * FP_MAX_PIECES + (it.local >> FP_PIECE_LOG)] * FP_PIECE + in_piece : D;
    // run B's bytes: in place, or in its pieces (which lie as run A's do: the same shift brings both to the output's words)
    const uint8_t* Bp = it.kind == 3 ? pool.base + (uint64_t)pool.tab[(uint64_t)(it.pad + it.span) * FP_MAX_PIECES + (it.local >> FP_PIECE_LOG)] * FP_PIECE + in_piece : D;
    uint32_t c = 0xffffffffu, adA = 0, adB = 0;
    auto one = [&](uint32_t i) {  // byte i of the item
        uint32_t v = A[i];
        if (it.kind >= 2) {
            const uint32_t x = v ^ Bp[i];
            if (x) v = T[(((x - 1u) << 8) | v) & (FP_TAIL - 1u)];
        }
        if (it.kind) D[i] = (uint8_t)v;
        if (container == 1) c = tab[0][(c ^ v) & 0xff] ^ (c >> 8);
        adA += v;
        adB += adA;
    };
    uint32_t i = lo;
    while (i < hi && (((uintptr_t)(D + i)) & 3)) one(i++);
    // (pieces are aligned, the output is where the caller put it: the pool is read through aligned words, one more
    // than the output has.)  A lane takes 128 bytes at a time: every cache line is requested once, not once per word.
    const uint32_t sa = (uint32_t)(((uintptr_t)(A + i)) & 3) * 8;
    auto word = [&](uint32_t v, uint32_t b) -> uint32_t {  // the true bytes of one word; into the checksum
        if (it.kind >= 2) {
            const uint32_t x = v ^ b;
            if (x) {
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    const uint32_t xa = (v >> (8 * y)) & 0xff, xd = (x >> (8 * y)) & 0xff;
                    if (xd) v = (v & ~(0xffu << (8 * y))) | ((uint32_t)T[(((xd - 1u) << 8) | xa) & (FP_TAIL - 1u)] << (8 * y));
                }
            }
        }
        if (container == 1) {
            c ^= v;
            c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
        } else if (container == 2) {
            adA += v & 0xff; adB += adA;
            adA += (v >> 8) & 0xff; adB += adA;
            adA += (v >> 16) & 0xff; adB += adA;
            adA += v >> 24; adB += adA;
        }
        return v;
    };
    for (; i + 128 <= hi; i += 128) {
        fl_u4a bq[8], aq[8];
        uint32_t a8 = 0, b8 = 0;
        if (it.kind == 3) {
            const fl_u4a* Bq = (const fl_u4a*)(((uintptr_t)(Bp + i)) & ~(uintptr_t)3);
#pragma unroll
            for (int q = 0; q < 8; q++) bq[q] = Bq[q];
            if (sa) b8 = ((const uint32_t*)Bq)[32];
        } else if (it.kind != 1) {
#pragma unroll
            for (int q = 0; q < 8; q++) bq[q] = ((const fl_u4a*)(D + i))[q];
        }
        if (it.kind) {
            const fl_u4a* Aq = (const fl_u4a*)(((uintptr_t)(A + i)) & ~(uintptr_t)3);
#pragma unroll
            for (int q = 0; q < 8; q++) aq[q] = Aq[q];
            if (sa) a8 = ((const uint32_t*)Aq)[32];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            fl_u4a o;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t v, b = 0;
                if (it.kind != 1) b = bq[q].x[w];
                if (it.kind == 3 && sa) b = __builtin_amdgcn_alignbit(w < 3 ? bq[q].x[(w + 1) & 3] : q < 7 ? bq[(q + 1) & 7].x[0] : b8, b, sa);
                if (it.kind) {
                    const uint32_t a0 = aq[q].x[w];
                    const uint32_t a1 = w < 3 ? aq[q].x[(w + 1) & 3] : q < 7 ? aq[(q + 1) & 7].x[0] : a8;
                    v = sa ? __builtin_amdgcn_alignbit(a1, a0, sa) : a0;
                } else {
                    v = b;
                }
                o.x[w] = word(v, b);
            }
            if (it.kind) ((fl_u4a*)(D + i))[q] = o;
        }
    }
    for (; i + 4 <= hi; i += 4) {
        uint32_t v, b = 0;
        if (it.kind == 3) {
            const uint32_t* Bw = (const uint32_t*)(((uintptr_t)(Bp + i)) & ~(uintptr_t)3);
            b = sa ? __builtin_amdgcn_alignbit(Bw[1], Bw[0], sa) : Bw[0];
        } else if (it.kind != 1) {
            b = *(const uint32_t*)(D + i);
        }
        if (it.kind) {
            const uint32_t* Aw = (const uint32_t*)(((uintptr_t)(A + i)) & ~(uintptr_t)3);
            const uint32_t a0 = Aw[0], a1 = sa ? Aw[1] : 0u;
            v = sa ? __builtin_amdgcn_alignbit(a1, a0, sa) : a0;
        } else {
            v = b;
        }
        v = word(v, b);
        if (it.kind) *(uint32_t*)(D + i) = v;
    }
    for (; i < hi; i++) one(i);
    const uint32_t after = it.len - hi;
    if (container == 1) {
        c = (hi > lo) ? ~c : 0u;  // crc of an empty slice is 0
        // x^(8 after): `after` is a multiple of the lane's share but for the item's last lanes -- a few factors
        uint32_t tpow = 0x80000000u;
        for (int j = 0; j < 16; j++)
            if (after & (1u << j)) tpow = fl_crc_mulmod(cc.xpow8[j], tpow);
        c = fl_crc_mulmod(c, tpow);
        c = fl_wave_xor(c);
        if (lane == 0) part[2 * (uint64_t)item_no] = c;
    } else if (container == 2) {
        // (<= 1024 bytes per lane: adB < 2^28) pieces with a = b = 0 start, lanes combined in order
        const uint32_t Bm = (uint32_t)(((uint64_t)adB + (uint64_t)adA * after) % 65521u);
        const uint32_t Am = fl_wave_sum(adA) % 65521u;  // <= 65536 * 255
        const uint32_t Bs = fl_wave_sum(Bm) % 65521u;
        if (lane == 0) part[2 * (uint64_t)item_no] = Am | (Bs << 16);
    }
    if (lane == 0) part[2 * (uint64_t)item_no + 1] = it.len;
}

// The footers of the streams that came out of the spans (one thread each; off = ~0: none), and what the host found:
// status 0, the length, the consumed count.
__global__ __launch_bounds__(64) void k_span_footers(const uint8_t* __restrict__ in, const uint64_t* __restrict__ off,
                                                     uint32_t n, uint32_t flen, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n || off[i] == ~0ull) return;
    for (uint32_t b = 0; b < flen; b++) out[8 * (uint64_t)i + b] = in[off[i] + b];
}
struct fl_span_fin {
    uint64_t total, used;
    uint32_t chunk, pad;
};
__global__ __launch_bounds__(64) void k_span_finish(const fl_span_fin* __restrict__ fin, uint32_t n,
                                                    int32_t* __restrict__ status, uint64_t* __restrict__ out_len,
                                                    uint64_t* __restrict__ consumed) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const fl_span_fin f = fin[i];
    status[f.chunk] = 0;
    out_len[f.chunk] = f.total;
    if (consumed) consumed[f.chunk] = f.used;
}

// The true tails of a stream's spans, in chain order (one workgroup per stream; see above).  tails_a is resolved in
// place.  The tail before the current one is kept in LDS: a span costs one round of loads and one barrier.
__global__ __launch_bounds__(FP_THREADS) void k_span_resolve(const uint32_t* __restrict__ chain,
                                                             const uint32_t* __restrict__ chain_off, uint8_t* tails_a,
                                                             const uint8_t* __restrict__ tails_b) {
    __shared__ uint32_t tl[2][FP_TAIL / 4];
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const uint32_t j0 = chain_off[k], j1 = chain_off[k + 1];
    if (j0 >= j1) return;
    {
        const uint4* t0 = (const uint4*)(tails_a + (uint64_t)chain[j0] * FP_TAIL);
        ((uint4*)tl[0])[2 * tid] = t0[2 * tid];
        ((uint4*)tl[0])[2 * tid + 1] = t0[2 * tid + 1];
    }
    __syncthreads();
    uint32_t cur = 0;
    // (the two fillings of the next span's tail are requested while the current one is resolved)
    uint4 na[2], nb[2];
    if (j0 + 1 < j1) {
        const uint4* pa = (const uint4*)(tails_a + (uint64_t)chain[j0 + 1] * FP_TAIL);
        const uint4* pb = (const uint4*)(tails_b + (uint64_t)chain[j0 + 1] * FP_TAIL);
        na[0] = pa[2 * tid]; na[1] = pa[2 * tid + 1];
        nb[0] = pb[2 * tid]; nb[1] = pb[2 * tid + 1];
    }
    for (uint32_t j = j0 + 1; j < j1; j++) {
        uint4* ta = (uint4*)(tails_a + (uint64_t)chain[j] * FP_TAIL);
        const uint8_t* prev = (const uint8_t*)tl[cur];
        uint32_t* mine = tl[cur ^ 1];
        uint4 a[2] = {na[0], na[1]};
        const uint4 b[2] = {nb[0], nb[1]};
        if (j + 1 < j1) {
            const uint4* pa = (const uint4*)(tails_a + (uint64_t)chain[j + 1] * FP_TAIL);
            const uint4* pb = (const uint4*)(tails_b + (uint64_t)chain[j + 1] * FP_TAIL);
            na[0] = pa[2 * tid]; na[1] = pa[2 * tid + 1];
            nb[0] = pb[2 * tid]; nb[1] = pb[2 * tid + 1];
        }
        bool changed = false;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint32_t aw[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
            const uint32_t bw[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t x = aw[w] ^ bw[w];
                if (x) {  // a byte that differs is history byte ((a ^ b) - 1) << 8 | a of the tail before
                    changed = true;
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        const uint32_t xa = (aw[w] >> (8 * y)) & 0xff, xd = (x >> (8 * y)) & 0xff;
                        if (xd) aw[w] = (aw[w] & ~(0xffu << (8 * y))) | ((uint32_t)prev[(((xd - 1u) << 8) | xa) & (FP_TAIL - 1u)] << (8 * y));  // (masked as in k_span_fix: a ^ b is at most 128 unless the two runs disagree)
                    }
                }
                mine[8 * tid + 4 * q + w] = aw[w];
            }
            a[q] = make_uint4(aw[0], aw[1], aw[2], aw[3]);
        }
        if (changed) {
            ta[2 * tid] = a[0];
            ta[2 * tid + 1] = a[1];
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------ the true tails, all spans at once (round 5)
// k_span_resolve walks a stream's chain span after span (one workgroup, 3.5 us a span: 0.87 ms for 248 spans).  A tail in SYMBOLS
// (below 256: a byte; 256 + h: byte h of the tail BEFORE) is a map from the tail before it to itself, and maps compose: after a
// step with stride D the symbols of the span at chain position j refer to the tail at position j - 2 D (Hillis-Steele: log2 of
// the chain's length steps, every span in every step, 64 KiB of symbols a span and step).  Positions below the stride hold bytes
// only: a stream's first span has no history.
//   S0 from the two planes (a, b): a == b: the byte; else 256 + (((a ^ b) - 1) << 8 | a)
__global__ __launch_bounds__(256) void k_span_rs_init(const uint32_t* __restrict__ chain, const uint32_t* __restrict__ pos,
                                                      const uint8_t* __restrict__ tails_a, const uint8_t* __restrict__ tails_b,
                                                      uint16_t* __restrict__ S0) {
    const uint32_t g = blockIdx.x, i0 = (blockIdx.y * 256u + threadIdx.x) * 8u;
    const uint32_t sp = chain[g];
    const uint2 a = *(const uint2*)(tails_a + (uint64_t)sp * FP_TAIL + i0);
    const uint2 b = pos[g] ? *(const uint2*)(tails_b + (uint64_t)sp * FP_TAIL + i0) : a;
    const uint32_t aw[2] = {a.x, a.y}, bw[2] = {b.x, b.y};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t av = (aw[k >> 2] >> (8 * (k & 3))) & 0xffu, bv = (bw[k >> 2] >> (8 * (k & 3))) & 0xffu;
        const uint32_t x = av ^ bv;
        const uint32_t sy = x ? 256u + ((((x - 1u) << 8) | av) & (FP_TAIL - 1u)) : av;
        if (k & 1) o[k >> 1] |= sy << 16; else o[k >> 1] = sy;
    }
    *(uint4*)(S0 + (uint64_t)g * FP_TAIL + i0) = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void k_span_rs_step(const uint32_t* __restrict__ pos, const uint16_t* __restrict__ Sin,
                                                      uint16_t* __restrict__ Sout, uint32_t D) {
    const uint32_t g = blockIdx.x, i0 = (blockIdx.y * 256u + threadIdx.x) * 8u;
    const uint4 v = *(const uint4*)(Sin + (uint64_t)g * FP_TAIL + i0);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if (pos[g] >= D) {
        const uint16_t* P = Sin + (uint64_t)(g - D) * FP_TAIL;  // (the same stream: its chain is contiguous)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t sy = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            if (sy >= 256u) {
                const uint32_t t = P[sy - 256u];
                w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | (t << (16 * (k & 1)));
            }
        }
    }
    *(uint4*)(Sout + (uint64_t)g * FP_TAIL + i0) = make_uint4(w[0], w[1], w[2], w[3]);
}
// ... and the bytes they have become, where k_span_fix looks for the true tails (plane A's)
__global__ __launch_bounds__(256) void k_span_rs_out(const uint32_t* __restrict__ chain, const uint16_t* __restrict__ S,
                                                     uint8_t* __restrict__ tails_a) {
    const uint32_t g = blockIdx.x, i0 = (blockIdx.y * 256u + threadIdx.x) * 8u;
    const uint4 v = *(const uint4*)(S + (uint64_t)g * FP_TAIL + i0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) o[k >> 2] |= ((w[k >> 1] >> (16 * (k & 1))) & 0xffu) << (8 * (k & 3));
    *(uint2*)(tails_a + (uint64_t)chain[g] * FP_TAIL + i0) = make_uint2(o[0], o[1]);
}
