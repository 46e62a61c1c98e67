// kernels_stream.h -- levels 4..9 for inputs longer than 65535 bytes and for inputs with
// sync-flush points: ONE deflate stream per input, byte-identical to what the reference produces
// when the input goes through Deflate.write / flush / finish (deflate.zig:304-371) with its
// sliding 64 KiB window.
//
// What changes against the chunk path (kernels_lz.h):
//  * match finding runs on overlapping 64 KiB tiles (k_lz_sort<true>, k_lz_match<true>): a tile
//    searches 32 KiB of new positions against 32 KiB of history, which is exactly what the
//    reference's window holds after a slide (SlidingWindow.zig:36-44, Lookup.zig:43-51).  The
//    records go to one per-stream array indexed by the absolute position.
//  * a sync flush (deflate.zig:335-337) cuts the stream into pieces: the tokenizer runs dry at the
//    flush point (no match crosses it, the 3 positions before it are never hashed), the pending
//    tokens go out as a block of their own and an empty stored block follows.  History survives.
//  * inside a piece the lazy-matching automaton (deflate.zig:154-205) is one chain.  It is cut
//    into segments of at most 32768 positions: every segment resolves "anchor -> first anchor
//    beyond the segment" for each of the <= 512 offsets a predecessor can hand over
//    (k_st_parse1), one thread per piece then walks segment to segment (k_st_stitch), and with
//    its entry anchor known each segment marks its anchors and counts its tokens (k_st_parse2).
//  * tokens are numbered through the piece; a block ends every 32768 tokens (deflate.zig:227-230),
//    wherever that falls (k_st_scan, k_st_emit, k_st_blocks).
//  * the raw input slice a block may be stored from (SlidingWindow.zig:119-123) is lost when the
//    window slid since the previous flush: fl_block_plan::no_input.
//
// The block planner, the offset scan and the bit packer (kernels_block.h) are shared.
#pragma once
#include "kernels_lz.h"

#define FL_SEG_ENTRIES 512u  // next anchor <= previous + 254 literals + 258: hand-over offsets < 512

// How many window slides the reference has done when `written` bytes of the stream have gone
// into the window: slide j happens as soon as the window is full for the j-th time
// (65536 + 32768 (j-1) bytes, deflate.zig:306-311).
__device__ __forceinline__ uint32_t fl_slides_when_written(uint32_t written) {
    return written >= 65536u ? (written - 65536u) / FL_SEG + 1u : 0u;
}
// ... and when it visits stream position v: zone[j-1] is the first position visited after slide j
// (the host folds the lookahead rule of SlidingWindow.zig:56-60 and the flush points into it).
__device__ __forceinline__ uint32_t fl_slides_before(uint32_t v, const uint32_t* __restrict__ zone, uint32_t n_slides) {
    uint32_t j = v >= FL_ZONE_START ? (v - FL_ZONE_START) / FL_SEG + 1u : 0u;
    j = min(j, n_slides);
    if (j && v < zone[j - 1]) j--;
    return j;
}

// ------------------------------------------------------------------ k_st_parse1
// One workgroup per segment.  desc[] for every position (as k_lz_tok phase a), the pointer
// table jumped inside 256-position runs (phase b) saved to jmp[], and the exit map
// exitmap[seg][e] = first anchor >= segment end on the path that enters at offset max(e, lo).
__global__ __launch_bounds__(FL_PARSE_THREADS, 8) void k_st_parse1(const fl_chunk* __restrict__ chunks,
                                                                    const fl_piece* __restrict__ pieces,
                                                                    const fl_seg* __restrict__ segs, fl_params prm,
                                                                    const uint32_t* __restrict__ rec_all,
                                                                    uint32_t* __restrict__ desc_all,
                                                                    uint16_t* __restrict__ jmp_all,
                                                                    uint16_t* __restrict__ exitmap) {
    __shared__ uint16_t J[FL_SEG];
    const fl_seg sg = segs[blockIdx.x];
    const fl_piece pc = pieces[sg.piece];
    const fl_chunk ck = chunks[pc.chunk];
    const uint32_t tid = threadIdx.x;
    const uint32_t N = ck.in_len;
    const uint32_t h0 = sg.h0, lo = max(h0, pc.start), h1 = min(h0 + FL_SEG, pc.end);
    const uint32_t rlo = lo - h0, len = h1 - h0;  // window-relative [rlo, len)
    const uint2* rec2 = (const uint2*)rec_all + ck.pos_off;
    uint32_t* desc = desc_all + ck.pos_off;
    uint16_t* jmp = jmp_all + ck.pos_off + h0;

    for (uint32_t base = lo; base < h1; base += FL_PARSE_THREADS * 8) {
        uint2 ra[8], rb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = base + u * FL_PARSE_THREADS + tid;
            ra[u] = p < h1 ? rec2[p] : make_uint2(0u, 0u);
            rb[u] = (p < h1 && p + 1 < N) ? rec2[p + 1] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = base + u * FL_PARSE_THREADS + tid;
            if (p < h1) {
                const uint32_t d = fl_anchor_desc(rec2, p, ra[u], rb[u], prm.good, prm.lazy);
                desc[p] = d;
                J[p - h0] = (uint16_t)(fl_desc_next(d, p) - h0);
            }
        }
    }
    __syncthreads();
    for (int round = 0; round < 8; round++) {
        for (uint32_t r = rlo + tid; r < len; r += FL_PARSE_THREADS) {
            const uint32_t run_end = min((r | 255u) + 1u, len);
            const uint32_t j = J[r];
            if (j < run_end) J[r] = J[j];
        }
        __syncthreads();
    }
    for (uint32_t r = rlo + tid; r < len; r += FL_PARSE_THREADS) jmp[r] = J[r];
    if (tid < FL_SEG_ENTRIES) {
        uint32_t a = max(tid, rlo);
        while (a < len) a = J[a];
        exitmap[(uint64_t)blockIdx.x * FL_SEG_ENTRIES + tid] = (uint16_t)a;
    }
}

// ------------------------------------------------------------------ k_st_stitch
// One thread per piece: the entry anchor of every segment (relative to the segment's h0).
__global__ __launch_bounds__(64) void k_st_stitch(const fl_piece* __restrict__ pieces, uint32_t n_pieces,
                                                  const fl_seg* __restrict__ segs,
                                                  const uint16_t* __restrict__ exitmap,
                                                  uint32_t* __restrict__ entry) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_pieces) return;
    const fl_piece pc = pieces[i];
    uint32_t a = pc.start;  // the tokenizer restarts at the piece start with nothing pending
    for (uint32_t s = 0; s < pc.n_seg; s++) {
        const uint32_t h0 = segs[pc.seg0 + s].h0;
        const uint32_t rel = a - h0;
        entry[pc.seg0 + s] = rel;
        // an entry at or before the segment's first position is slot 0 of the map
        const uint32_t slot = h0 >= pc.start ? rel : 0u;
        if (s + 1 < pc.n_seg) a = h0 + (uint32_t)exitmap[(uint64_t)(pc.seg0 + s) * FL_SEG_ENTRIES + slot];
    }
}

// Pieces of many segments (a 1 GiB stream has 32768) would make that walk the longest thing in
// the pass, so it is done in three short steps over groups of FL_STITCH_GROUP segments: every
// group composes its segments' exit maps for all 512 possible entries (k_st_stitch_a), one thread
// per piece walks the groups (k_st_stitch_b), one thread per group walks its segments from the
// group's now known entry (k_st_stitch_c).
#define FL_STITCH_GROUP 64u
struct fl_sgroup {
    uint32_t piece;
    uint32_t g;  // group number inside the piece
};
// one step of the walk: anchor `a` (stream position) enters segment s of the piece
__device__ __forceinline__ uint32_t fl_stitch_step(const fl_piece& pc, const fl_seg* __restrict__ segs,
                                                   const uint16_t* __restrict__ exitmap, uint32_t s, uint32_t a) {
    const uint32_t h0 = segs[pc.seg0 + s].h0;
    const uint32_t slot = h0 >= pc.start ? a - h0 : 0u;  // an entry at or before the first position is slot 0
    return h0 + (uint32_t)exitmap[(uint64_t)(pc.seg0 + s) * FL_SEG_ENTRIES + slot];
}
__global__ __launch_bounds__(FL_SEG_ENTRIES) void k_st_stitch_a(const fl_piece* __restrict__ pieces,
                                                                const fl_sgroup* __restrict__ groups,
                                                                const fl_seg* __restrict__ segs,
                                                                const uint16_t* __restrict__ exitmap,
                                                                uint16_t* __restrict__ gmap) {
    const fl_sgroup gr = groups[blockIdx.x];
    const fl_piece pc = pieces[gr.piece];
    const uint32_t s0 = gr.g * FL_STITCH_GROUP, s1 = min(s0 + FL_STITCH_GROUP, pc.n_seg);
    if (s1 >= pc.n_seg) return;  // the last group hands over to nobody
    // entries of group 0 all mean "the piece start"; of later groups: h0 of the group's first segment + e
    uint32_t a = gr.g == 0 ? pc.start : segs[pc.seg0 + s0].h0 + threadIdx.x;
    for (uint32_t s = s0; s < s1; s++) a = fl_stitch_step(pc, segs, exitmap, s, a);
    gmap[(uint64_t)blockIdx.x * FL_SEG_ENTRIES + threadIdx.x] = (uint16_t)(a - segs[pc.seg0 + s1].h0);
}
__global__ __launch_bounds__(64) void k_st_stitch_b(const fl_piece* __restrict__ pieces, uint32_t n_pieces,
                                                    const uint32_t* __restrict__ piece_group0,
                                                    const fl_seg* __restrict__ segs,
                                                    const uint16_t* __restrict__ gmap,
                                                    uint32_t* __restrict__ gentry) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_pieces) return;
    const fl_piece pc = pieces[i];
    const uint32_t ng = (pc.n_seg + FL_STITCH_GROUP - 1) / FL_STITCH_GROUP, g0 = piece_group0[i];
    uint32_t a = pc.start;
    for (uint32_t g = 0; g < ng; g++) {
        gentry[g0 + g] = a;
        if (g + 1 < ng) {
            const uint32_t slot = g == 0 ? 0u : a - segs[pc.seg0 + g * FL_STITCH_GROUP].h0;
            a = segs[pc.seg0 + (g + 1) * FL_STITCH_GROUP].h0 + (uint32_t)gmap[(uint64_t)(g0 + g) * FL_SEG_ENTRIES + slot];
        }
    }
}
__global__ __launch_bounds__(64) void k_st_stitch_c(const fl_piece* __restrict__ pieces,
                                                    const fl_sgroup* __restrict__ groups, uint32_t n_groups,
                                                    const fl_seg* __restrict__ segs,
                                                    const uint16_t* __restrict__ exitmap,
                                                    const uint32_t* __restrict__ gentry,
                                                    uint32_t* __restrict__ entry) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_groups) return;
    const fl_sgroup gr = groups[i];
    const fl_piece pc = pieces[gr.piece];
    const uint32_t s0 = gr.g * FL_STITCH_GROUP, s1 = min(s0 + FL_STITCH_GROUP, pc.n_seg);
    uint32_t a = gentry[i];
    for (uint32_t s = s0; s < s1; s++) {
        entry[pc.seg0 + s] = a - segs[pc.seg0 + s].h0;
        if (s + 1 < pc.n_seg) a = fl_stitch_step(pc, segs, exitmap, s, a);
    }
}

// ------------------------------------------------------------------ k_st_parse2
// One workgroup per segment: anchors of the segment (as k_lz_tok phases c-e) from its entry
// anchor, and the number of tokens they emit.  marks_all was cleared by the host: segments of
// neighbouring pieces may share a word.
__global__ __launch_bounds__(FL_PARSE_THREADS, 8) void k_st_parse2(const fl_chunk* __restrict__ chunks,
                                                                    const fl_piece* __restrict__ pieces,
                                                                    const fl_seg* __restrict__ segs,
                                                                    const uint32_t* __restrict__ desc_all,
                                                                    const uint16_t* __restrict__ jmp_all,
                                                                    const uint32_t* __restrict__ entry,
                                                                    uint32_t* __restrict__ marks_all,
                                                                    uint32_t* __restrict__ segtok) {
    __shared__ uint16_t J[FL_SEG];
    __shared__ uint32_t marks[FL_SEG / 32];
    __shared__ uint16_t run_entry[FL_SEG / 256];
    __shared__ uint32_t wsum[FL_PARSE_THREADS / 64];
    const fl_seg sg = segs[blockIdx.x];
    const fl_piece pc = pieces[sg.piece];
    const fl_chunk ck = chunks[pc.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t h0 = sg.h0, lo = max(h0, pc.start), h1 = min(h0 + FL_SEG, pc.end);
    const uint32_t rlo = lo - h0, len = h1 - h0;
    const uint32_t* desc = desc_all + ck.pos_off + h0;
    const uint16_t* jmp = jmp_all + ck.pos_off + h0;
    uint32_t* gmarks = marks_all + ((ck.pos_off + h0) >> 5);

    for (uint32_t r = rlo + tid; r < len; r += FL_PARSE_THREADS) J[r] = jmp[r];
    for (uint32_t i = tid; i < FL_SEG / 32; i += FL_PARSE_THREADS) marks[i] = 0;
    if (tid < FL_SEG / 256) run_entry[tid] = 0xffff;
    __syncthreads();
    if (tid == 0) {  // first anchor of every 256-position run: at most 128 serial steps
        uint32_t a = entry[blockIdx.x];
        while (a < len) {
            run_entry[a >> 8] = (uint16_t)a;
            a = J[a];
        }
    }
    __syncthreads();
    for (uint32_t r = rlo + tid; r < len; r += FL_PARSE_THREADS) J[r] = (uint16_t)(fl_desc_next(desc[r], r));
    __syncthreads();
    if (tid < FL_SEG / 256) {
        uint32_t a = run_entry[tid];
        const uint32_t end = min((tid + 1) << 8, len);
        while (a < end) {
            marks[a >> 5] |= 1u << (a & 31);
            a = J[a];
        }
    }
    __syncthreads();
    uint32_t cnt = 0;
    for (uint32_t r = rlo + tid; r < len; r += FL_PARSE_THREADS) {
        if ((marks[r >> 5] >> (r & 31)) & 1) {
            const uint32_t d = desc[r];
            cnt += d ? ((d >> 23) & 0xff) + 1 : 1;
        }
    }
    for (uint32_t i = tid; i < FL_SEG / 32; i += FL_PARSE_THREADS)
        if (marks[i]) atomicOr(&gmarks[i], marks[i]);
    cnt = fl_wave_sum(cnt);
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < FL_PARSE_THREADS / 64; w++) t += wsum[w];
        segtok[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ k_st_count
// Round 5: when the anchors come from k_lz_parse<true> (kernels_parse.h: the demand-driven tokenizer walking the stream's windows),
// all that is left of k_st_parse2 is its count: the tokens of the segment's marked anchors.
#ifndef FL_CNT_U
#define FL_CNT_U 8u
#endif
__global__ __launch_bounds__(FL_PARSE_THREADS, 8) void k_st_count(const fl_chunk* __restrict__ chunks,
                                                                   const fl_piece* __restrict__ pieces,
                                                                   const fl_seg* __restrict__ segs,
                                                                   const uint32_t* __restrict__ desc_all,
                                                                   const uint32_t* __restrict__ marks_all,
                                                                   uint32_t* __restrict__ segtok) {
    __shared__ uint32_t wsum[FL_PARSE_THREADS / 64];
    const fl_seg sg = segs[blockIdx.x];
    const fl_piece pc = pieces[sg.piece];
    const fl_chunk ck = chunks[pc.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t h0 = sg.h0, lo = max(h0, pc.start), h1 = min(h0 + FL_SEG, pc.end);
    const uint32_t rlo = lo - h0, len = h1 - h0;
    const uint32_t* desc = desc_all + ck.pos_off + h0;
    const uint32_t* gmarks = marks_all + ((ck.pos_off + h0) >> 5);
    uint32_t cnt = 0;
    // a lane per position, 64 consecutive positions per wave and step: the marked lanes' descriptors lie in two or three lines.
    // FL_CNT_U steps' loads are in flight together (the marks of all of them, then the descriptors: a step used to be two
    // dependent round trips to memory, 128 steps a wave)
    for (uint32_t rb = tid & ~63u; rb < len; rb += FL_CNT_U * FL_PARSE_THREADS) {
        uint64_t mk[FL_CNT_U];
        uint32_t d[FL_CNT_U];
#pragma unroll
        for (uint32_t u = 0; u < FL_CNT_U; u++) {
            const uint32_t r0 = rb + u * FL_PARSE_THREADS;
            mk[u] = r0 < len ? ((uint64_t)gmarks[r0 >> 5] | ((uint64_t)gmarks[(r0 >> 5) + 1] << 32)) : 0ull;  // (wave-uniform)
        }
#pragma unroll
        for (uint32_t u = 0; u < FL_CNT_U; u++) {
            const uint32_t r = rb + u * FL_PARSE_THREADS + lane;
            const bool on = ((mk[u] >> lane) & 1ull) && r >= rlo && r < len;
            d[u] = on ? desc[r] : 0xffffffffu;
        }
#pragma unroll
        for (uint32_t u = 0; u < FL_CNT_U; u++)
            if (d[u] != 0xffffffffu) cnt += d[u] ? ((d[u] >> 23) & 0xff) + 1 : 1;
    }
    cnt = fl_wave_sum(cnt);
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < FL_PARSE_THREADS / 64; w++) t += wsum[w];
        segtok[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ k_st_scan
// One wave per piece: index of every segment's first token, and the piece's token count.
__global__ __launch_bounds__(64) void k_st_scan(const fl_piece* __restrict__ pieces,
                                                const uint32_t* __restrict__ segtok,
                                                uint32_t* __restrict__ tokbase, uint32_t* __restrict__ piece_ntok) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    const fl_piece pc = pieces[i];
    uint32_t run = 0;
    for (uint32_t s0 = 0; s0 < pc.n_seg; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t v = s < pc.n_seg ? segtok[pc.seg0 + s] : 0;
        const uint32_t inc = fl_wave_incl_scan(v, lane);
        if (s < pc.n_seg) tokbase[pc.seg0 + s] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) piece_ntok[i] = run;
}

// ------------------------------------------------------------------ k_st_emit
// One workgroup per segment: tokens (token k of a piece at tokens[pos_off + piece.start + k]),
// per-block histograms (added into hist_all, which the host cleared) and the stream position at
// which each full block was flushed (bound[]).
#define FL_STE_SPAN (FL_SEG / FL_EMIT_WAVES)
#define FL_STE_WIN_DW (FL_SEG / 4 + 72)
__global__ __launch_bounds__(FL_EMIT_THREADS, 8) void k_st_emit(const uint8_t* __restrict__ in,
                                                              const fl_chunk* __restrict__ chunks,
                                                              const fl_piece* __restrict__ pieces,
                                                              const fl_seg* __restrict__ segs, fl_params prm,
                                                              const uint32_t* __restrict__ desc_all,
                                                              const uint32_t* __restrict__ marks_all,
                                                              const uint32_t* __restrict__ tokbase,
                                                              uint32_t* __restrict__ tokens_all,
                                                              uint32_t* __restrict__ hist_all,
                                                              uint32_t* __restrict__ bound,
                                                              uint32_t* __restrict__ qgap) {
    __shared__ uint32_t win32[FL_STE_WIN_DW];
    __shared__ uint32_t marks[FL_SEG / 32];
    __shared__ uint32_t hist[2][320];
    __shared__ uint32_t wtot[FL_EMIT_WAVES];
    const fl_seg sg = segs[blockIdx.x];
    const fl_piece pc = pieces[sg.piece];
    const fl_chunk ck = chunks[pc.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = ck.in_len;
    const uint32_t h0 = sg.h0, lo = max(h0, pc.start), h1 = min(h0 + FL_SEG, pc.end);
    const uint32_t rlo = lo - h0, len = h1 - h0;
    const uint8_t* src = in + ck.in_off + h0;
    const uint32_t* desc = desc_all + ck.pos_off + h0;
    const uint32_t* gmarks = marks_all + ((ck.pos_off + h0) >> 5);
    uint32_t* tokens = tokens_all + ck.pos_off + pc.start;

    const uint32_t nleft = N - h0;
    const uint32_t ndw = (min(nleft, FL_STE_WIN_DW * 4u) + 3) >> 2;
    for (uint32_t i = tid; i < FL_STE_WIN_DW; i += FL_EMIT_THREADS)
        win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, nleft) : 0u;
    for (uint32_t i = tid; i < FL_SEG / 32; i += FL_EMIT_THREADS) marks[i] = gmarks[i];
    for (uint32_t i = tid; i < 640; i += FL_EMIT_THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t span0 = wave * FL_STE_SPAN;
    uint32_t cnt = 0;
    for (uint32_t r = 0; r < FL_STE_SPAN / 64; r += 4) {
        uint32_t d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = span0 + (r + u) * 64 + lane;
            d[u] = (p >= rlo && p < len) ? desc[p] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = span0 + (r + u) * 64 + lane;
            const bool mk = p >= rlo && p < len && ((marks[p >> 5] >> (p & 31)) & 1);
            cnt += mk ? (d[u] ? ((d[u] >> 23) & 0xff) + 1 : 1) : 0;
        }
    }
    cnt = fl_wave_sum(cnt);
    if (lane == 0) wtot[wave] = cnt;
    __syncthreads();
    const uint32_t tb = tokbase[blockIdx.x];
    const uint32_t blk0 = tb >> 15;  // block of the piece holding this segment's first token
    uint32_t run = tb;
    for (uint32_t w = 0; w < wave; w++) run += wtot[w];
    for (uint32_t r = 0; r < FL_STE_SPAN / 64; r += 4) {
        uint32_t d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = span0 + (r + u) * 64 + lane;
            d[u] = (p >= rlo && p < len) ? desc[p] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = span0 + (r + u) * 64 + lane;
            const bool mk = p >= rlo && p < len && ((marks[p >> 5] >> (p & 31)) & 1);
            const uint32_t dd = d[u];
            const uint32_t nl = mk ? (dd ? ((dd >> 23) & 0xff) : 1u) : 0u;
            const uint32_t nt = mk ? (dd ? nl + 1 : 1u) : 0u;
            const uint32_t incl = fl_wave_incl_scan(nt, lane);
            uint32_t idx = run + incl - nt;
            run += __shfl(incl, 63, 64);
            for (uint32_t x = 0; x < nl; x++) {
                const uint32_t byte = fl_win_byte(win32, p + x);
                tokens[idx] = FL_TOK_LIT(byte);
                atomicAdd(&hist[(idx >> 15) - blk0][byte], 1u);
                // a literal goes out at the visit of the next position (deflate.zig:214-216)
                if ((idx & (FL_MAX_TOKENS - 1)) == FL_MAX_TOKENS - 1) {
                    bound[pc.first_block + (idx >> 15) + 1] = h0 + p + x + 1;
                    qgap[pc.first_block + (idx >> 15) + 1] = 0;
                }
                idx++;
            }
            if (mk && dd) {
                const uint32_t ll = (dd >> 15) & 0xff, d0 = dd & 0x7fff;
                tokens[idx] = (1u << 23) | (ll << 15) | d0;
                atomicAdd(&hist[(idx >> 15) - blk0][257 + fl_len_index(ll)], 1u);
                atomicAdd(&hist[(idx >> 15) - blk0][286 + fl_dist_code(d0)], 1u);
                // a match of at least `lazy` goes out at its own visit, a shorter one at the next
                if ((idx & (FL_MAX_TOKENS - 1)) == FL_MAX_TOKENS - 1) {
                    const uint32_t adv = (ll + 3 >= prm.lazy) ? 0u : 1u;
                    bound[pc.first_block + (idx >> 15) + 1] = h0 + p + nl + adv;
                    qgap[pc.first_block + (idx >> 15) + 1] = ll + 3 - adv;  // Q1: the match's bytes the window has not advanced over
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 640; i += FL_EMIT_THREADS) {
        const uint32_t v = (&hist[0][0])[i];
        if (v) atomicAdd(&hist_all[(uint64_t)(pc.first_block + blk0 + i / 320) * 320 + i % 320], v);
    }
}

// ------------------------------------------------------------------ k_st_blocks
// One wave per piece: its block table (deflate.zig:268-288): block k holds tokens
// [32768 k, 32768 (k+1)); the last one, possibly empty, goes out at the flush / finish that ends
// the piece; after a sync flush an empty stored block follows (deflate.zig:276-278).
__global__ __launch_bounds__(64) void k_st_blocks(const fl_chunk* __restrict__ chunks,
                                                  const fl_piece* __restrict__ pieces,
                                                  const uint32_t* __restrict__ piece_ntok,
                                                  const uint32_t* __restrict__ bound,
                                                  const uint32_t* __restrict__ qgap, uint32_t repair,
                                                  const uint32_t* __restrict__ zones,
                                                  fl_block_plan* __restrict__ plans) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    const fl_piece pc = pieces[i];
    const fl_chunk ck = chunks[pc.chunk];
    const uint32_t total = piece_ntok[i];
    const uint32_t nblk = total / FL_MAX_TOKENS + 1;
    const uint32_t* zone = zones + ck.zone_off;
    for (uint32_t k = lane; k < pc.n_blocks; k += 64) {
        fl_block_plan* plan = &plans[pc.first_block + k];
        if (k == nblk && (pc.flags & 2)) {  // the sync-flush marker: BFINAL 0, BTYPE 00, LEN 0, NLEN ffff
            plan->valid = 2;  // k_plan leaves it alone
            plan->type = FL_BLOCK_STORED;
            plan->size_bits = 0;
            plan->hdr_nbits = 0;
            plan->final_block = 0;
            plan->in_start = 0;
            plan->in_len = 0;
            plan->tok_start = 0;
            plan->tok_count = 0;
            plan->no_input = 0;
            plan->q1_gap = 0;
            continue;
        }
        if (k >= nblk) {
            plan->valid = 0;
            continue;
        }
        const bool last = k + 1 == nblk;
        // (Q1: a full block that ends in a match is flushed before the window advances over the match; with `repair` the slices
        // are the bytes the tokens cover, the slides still those of the visit in which the block is flushed)
        const uint32_t g0 = k ? qgap[pc.first_block + k] : 0u, g1 = last ? 0u : qgap[pc.first_block + k + 1];
        uint32_t start = k ? bound[pc.first_block + k] : pc.start;
        uint32_t end = last ? pc.end : bound[pc.first_block + k + 1];
        // window start when the block goes out: inside the visit of `end` for a full block; for
        // the last one at the flush / finish call, when every slide the written bytes caused is done
        const uint32_t slides = last ? fl_slides_when_written(pc.end) : fl_slides_before(end, zone, ck.n_slides);
        plan->valid = 1;
        plan->tok_start = pc.start + k * FL_MAX_TOKENS;
        plan->tok_count = last ? total - k * FL_MAX_TOKENS : FL_MAX_TOKENS;
        if (repair) {
            start += g0;
            end += g1;
        }
        plan->q1_gap = repair ? 0u : g1;
        const bool no_input = start < slides * FL_SEG;  // SlidingWindow.zig:40, 119-123: fp went negative
        plan->in_start = start;
        plan->in_len = no_input ? FL_NO_INPUT : end - start;  // (a block without input is never stored)
        plan->final_block = (last && (pc.flags & 1)) ? 1 : 0;
        plan->no_input = no_input ? 1 : 0;
    }
}
