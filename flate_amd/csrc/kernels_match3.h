// kernels_match3.h -- match finder experiment (FLATE_HIP_DBG=2048, levels 4..6): k_lz_match with a rolling
// LDS buffer instead of rebuilt tiles.  Bit-exact on the whole GPU suite, NOT the default: it issues
// fewer instructions (PMC per 64 KiB of text: 825 k VALU / 64 k LDS against 884 k / 174 k) but its
// buffers leave room for only 16 waves per CU instead of 32, and the window walks of the 8-byte
// candidates (a third of the instructions, latency bound) then cost 9 ms instead of 4.5: 30.6 ms per
// GiB against 25.9.  Kept for the measurements DESIGN.md 4 quotes.
//
// Reference path: Deflate.findMatch (deflate.zig:233-266) with SlidingWindow.match
// (SlidingWindow.zig:81-104) over the hash chains of Lookup (Lookup.zig:12-84), for EVERY position:
// rec[p] = { record for the full chain budget, record for chain >> 2 (deflate.zig:241-245) }.
//
// Same decomposition as k_lz_match (kernels_lz.h): S = the positions sorted by (hash, position), the
// chain candidates of a sorted entry are the entries just before it in its bucket, lane = entry,
// loop = candidate number, candidates scored branch-free on their first 8 window bytes, the window
// itself only meets the candidates that agree in all 8.  What differs:
//  * a wave STREAMS through its slice of the sorted array (8192 consecutive entries) and keeps
//    the 8-byte prefixes and positions of the last 128 entries in a rolling LDS buffer, so every
//    entry's prefix is gathered from the window exactly once, and candidate number k of lane l is
//    slot 128 + l - k: consecutive lanes read consecutive 8-byte words, no bank conflicts.
//  * bucket offsets (how many candidates an entry has) come from the hashes of the prefixes that
//    are in registers anyway: no pre-pass over the sorted array, no NQ array in HBM.
//  * a candidate is valid iff its number is <= n; the position rules (distance <= 32768, position
//    0 is the chain's null, the slide zone of whole-stream tiles) shorten n up front by a binary
//    search over the positions, which fall along the chain -- so the loop reads no positions.
//
// 8 waves per workgroup (window 64 KiB + 8 x 1.9 KiB of rolling buffers = 79 KiB: two workgroups
// per CU).  Bound: vector-ALU issue.  No MFMA: byte compares and maxima.
#pragma once
#include "kernels_common.h"
#include "kernels_lz.h"

#define FL_M3_WAVES 8
#define FL_M3_THREADS (64 * FL_M3_WAVES)
#define FL_M3_BACK 128u                       // candidates kept behind the current batch (>= chain)
#define FL_M3_RING (FL_M3_BACK + 64u)
#define FL_M3_SLICE (65536u / FL_M3_WAVES)    // sorted entries per wave

template <bool STREAM>
__global__ __launch_bounds__(FL_M3_THREADS, 4) void k_lz_match3(const uint8_t* __restrict__ in,
                                                               const fl_chunk* __restrict__ chunks,
                                                               const fl_tile* __restrict__ tiles,
                                                               const uint32_t* __restrict__ fpts,
                                                               const uint32_t* __restrict__ n_sorted, fl_params prm,
                                                               const uint16_t* __restrict__ S,
                                                               uint32_t* __restrict__ rec_all) {
    constexpr uint32_t WIN_DW = STREAM ? FL_WIN_DW_STREAM : FL_WIN_DW_CHUNK;
    __shared__ uint32_t win32[WIN_DW];
    __shared__ uint2 rW[FL_M3_WAVES][FL_M3_RING];
    __shared__ uint16_t rS[FL_M3_WAVES][FL_M3_RING];
    const uint32_t c = blockIdx.x;
    const uint32_t w0 = STREAM ? tiles[c].w0 : 0u;
    const uint32_t tgt0 = STREAM ? tiles[c].tgt0 : 0u;
    const fl_chunk ck = chunks[STREAM ? tiles[c].chunk : c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t zone = STREAM ? tiles[c].zone : 65536u;
    const uint32_t N = ck.in_len - w0;
    const uint32_t Mpos = min(N >= 4 ? N - 3 : 0u, 65536u);  // positions with 4 bytes left in the stream
    const uint32_t M = STREAM ? n_sorted[c] : Mpos;            // entries of the sorted array
    const uint32_t* fp = STREAM ? fpts + ck.flush_off : nullptr;
    // a flush point up to 258 bytes past the last position still shortens matches in this window
    const bool has_fl =
        STREAM && ck.n_flush && fl_next_flush(fp, ck.n_flush, w0, ck.in_len) <= w0 + Mpos + 2 + FL_MAX_MATCH;
    const uint8_t* src = in + ck.in_off + w0;
    const uint16_t* Sc = S + (uint64_t)c * FL_CHUNK_STRIDE;
    uint2* rec2 = (uint2*)rec_all + ck.pos_off + w0;
    const uint32_t chain = prm.chain, quarter = prm.chain >> 2, nice = prm.nice;

    fl_prof_mark(8);
    // stage the window in LDS (zero padded)
    const uint32_t ndw = (min(N, WIN_DW * 4u) + 3) >> 2;
    for (uint32_t i = tid; i < WIN_DW; i += FL_M3_THREADS)
        win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, N) : 0u;
    // positions without a hash entry never match (Lookup.zig:24)
    // (with flush points in the stream the host has cleared all records beforehand)
    for (uint32_t p = Mpos + tid; p < min(N, 65536u); p += FL_M3_THREADS) rec2[p] = make_uint2(0u, 0u);
    __syncthreads();
    fl_prof_mark(9);

    const uint32_t slice0 = wave * FL_M3_SLICE;
    if (slice0 >= M) return;  // (no barrier below)
    uint2* rw = rW[wave];
    uint16_t* rs = rS[wave];
    const uint64_t le_mask = ~0ull >> (63 - lane);

    // ---- the 128 entries before the slice: prefixes, positions, and where the bucket that runs
    // into the slice starts (encoded index + 1; an offset >= 128 >= chain needs no exact start)
    uint32_t carry = 0;
    uint32_t hlast = 0xfffffffeu;  // hash of the entry just before the current batch
    if (slice0) {
        const uint32_t ia = slice0 - FL_M3_BACK + lane, ib = ia + 64;  // slice0 >= 8192
        const uint32_t qa = Sc[ia], qb = Sc[ib];
        uint32_t a0, a1, b0, b1;
        fl_lds_load8(win32, qa, a0, a1);
        fl_lds_load8(win32, qb, b0, b1);
        rw[lane] = make_uint2(a0, a1);
        rw[lane + 64] = make_uint2(b0, b1);
        rs[lane] = (uint16_t)qa;
        rs[lane + 64] = (uint16_t)qb;
        const uint32_t ha = fl_hash_le(a0), hb = fl_hash_le(b0);
        uint32_t hpa = __shfl_up(ha, 1, 64), hpb = __shfl_up(hb, 1, 64);
        const uint32_t ha63 = __shfl(ha, 63, 64);
        if (lane == 0) hpb = ha63;
        const uint64_t sa = __ballot(lane > 0 && ha != hpa), sb = __ballot(hb != hpb);
        carry = slice0 - FL_M3_BACK + 1;
        if (sa) carry = slice0 - FL_M3_BACK + 1 + 63u - (uint32_t)__builtin_clzll(sa);
        if (sb) carry = slice0 - 64 + 1 + 63u - (uint32_t)__builtin_clzll(sb);
        hlast = __shfl(hb, 63, 64);
    }
    fl_prof_mark(10);

    const uint32_t slice1 = min(slice0 + FL_M3_SLICE, M);
    uint32_t nx_p = (slice0 + lane < M) ? Sc[slice0 + lane] : 0;
    for (uint32_t i0 = slice0; i0 < slice1; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool valid = i < M;
        const uint32_t p = nx_p;
        const bool active = valid && (!STREAM || p >= tgt0);
        {
            const uint32_t in_ = i + 64;
            nx_p = (i0 + 64 < slice1 && in_ < M) ? Sc[in_] : 0;
        }
        uint32_t p0, p1;
        fl_lds_load8(win32, p, p0, p1);
        fl_lds_order();
        rw[FL_M3_BACK + lane] = make_uint2(p0, p1);
        rs[FL_M3_BACK + lane] = (uint16_t)p;
        fl_lds_order();
        // ---- n = min(offset in the bucket, chain) = how many chain candidates the entry has
        uint32_t n;
        {
            const uint32_t h = valid ? fl_hash_le(p0) : 0xffffffffu;
            uint32_t hp = __shfl_up(h, 1, 64);
            if (lane == 0) hp = hlast;
            hlast = __shfl(h, 63, 64);
            const uint64_t starts = __ballot(valid && (i == 0 || h != hp));
            const uint64_t below = starts & le_mask;
            const uint32_t base1 = i0 + 1;
            const uint32_t st = below ? base1 + 63u - (uint32_t)__builtin_clzll(below) : carry;
            n = valid ? min(i + 1 - st, chain) : 0u;
            if (starts) carry = base1 + 63u - (uint32_t)__builtin_clzll(starts);
        }
        // ---- position rules: q >= 1 (position 0 is the chain's null, deflate.zig:248), p - q <= 32768
        // (deflate.zig:250-251), after a slide nothing at or below the new window start (Lookup.zig:43-51).
        // Positions fall along the chain: the valid candidates are a prefix of the first n.
        {
            uint32_t lowp = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;
            if (STREAM && p >= zone) lowp = max(lowp, FL_MAX_DIST + 1u);
            const uint32_t qn = n ? rs[FL_M3_BACK + lane - n] : 0xffffu;
            const bool trim = n && qn < lowp;
            if (__any(trim)) {
                uint32_t lo = 0, hi = trim ? n : 0u;  // candidate lo is valid (or lo == 0), candidate hi is not
                while (__any(hi - lo > 1)) {
                    if (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (rs[FL_M3_BACK + lane - mid] >= lowp)
                            lo = mid;
                        else
                            hi = mid;
                    }
                }
                if (trim) n = lo;
            }
        }
        if (!active || (prm.dbg & 8)) n = 0;  // (8: timing experiment, wrong output)
        uint32_t maxlen = min(N - p, FL_MAX_MATCH);
        if (STREAM && has_fl) maxlen = min(maxlen, fl_next_flush(fp, ck.n_flush, w0 + p, ck.in_len) - (w0 + p));
        const uint32_t cp = 0xffffu - p;  // key low half = 65535 - (p - q) = q + cp
        uint32_t key = 0, qkey = 0;
        uint32_t pb = 0;     // window bytes p+best-3 .. p+best (valid when best >= 8)
        uint32_t dmask = 0;  // candidates whose 8 prefix bytes all agree; bit b = candidate kdone - b of the tile
        uint32_t bb = 0;     // best score of the current tile: equal-byte flags | 0x40 | 32 - candidate number
        uint32_t kdone = 0;  // candidates of the current tile scored so far
        bool qsnap = false;
        const uint32_t lenmask = maxlen >= 8 ? 0x80808080u : (0x00808080u >> (8 * (7 - maxlen)));
        const uint2* twp = rw + FL_M3_BACK + lane;     // candidate t of the current tile: twp[-t]
        const uint16_t* tsp = rs + FL_M3_BACK + lane;
        // the candidates that agree in 8 bytes meet the window, nearest first (SlidingWindow.zig:81-104)
        auto flush_deep = [&]() {
            if (maxlen <= 8 || (prm.dbg & 4)) dmask = 0;  // (4: timing experiment, wrong output)
            while (__any(dmask != 0)) {
                if (dmask) {
                    const uint32_t b = 31u - (uint32_t)__builtin_clz(dmask);
                    dmask &= ~(1u << b);
                    const uint32_t q = *(tsp - (kdone - b));
                    const uint32_t best = key >> 16;
                    // SlidingWindow.zig:91-98: a candidate that does not extend the best match is
                    // dropped on one compare
                    bool take = maxlen > best;
                    if (take && best >= 8) take = fl_lds_load4(win32, q + best - 3) == pb;
                    if (take) {
                        const uint32_t le = fl_extend_match(win32, p, q, maxlen);
                        const uint32_t kc = (le << 16) | (q + cp);
                        if (kc > key) {  // deflate.zig:254-261
                            key = kc;
                            pb = fl_lds_load4(win32, p + le - 3);
                            if (le >= maxlen || le >= nice) {  // nothing longer possible / stop looking
                                n = 0;
                                dmask = 0;
                            }
                        }
                    }
                }
            }
        };
        // the tile's winner meets the key, then the 8-byte candidates meet the window
        auto tile_end = [&]() {
            if (__any(bb != 0)) {
                if (bb) {
                    const uint32_t q = *(tsp - (32u - (bb & 31u)));
                    const uint32_t le = min(4u + (uint32_t)__popc(bb & 0x80808080u), maxlen);
                    const uint32_t kc = (le << 16) | (q + cp);
                    if (kc > key) {  // deflate.zig:254-261
                        key = kc;
                        if (le >= 8) pb = fl_lds_load4(win32, p + le - 3);
                        if (le >= maxlen) {  // nothing longer possible (le <= 8 < nice here)
                            n = 0;
                            dmask = 0;
                        }
                    }
                }
                bb = 0;
            }
            flush_deep();
        };
        for (uint32_t kb = 0; kb < chain; kb += 32) {
            if (!__any(n > kb)) break;
            kdone = 0;
#pragma unroll
            for (uint32_t kk0 = 1; kk0 <= 32; kk0 += 4) {
                // the chain >> 2 budget (deflate.zig:241-245) ends after candidate `quarter`
                // (a multiple of 4 at every level, deflate.zig:44-49)
                if (kb + kk0 - 1 == quarter) {
                    tile_end();
                    qkey = key;
                    qsnap = true;
                }
                if (!__any(n >= kb + kk0)) break;
                uint2 w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) w[u] = *(twp - (kk0 + u));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t x1 = w[u].y ^ p1;
                    const uint32_t m = ~x1 & (x1 - 1u);  // ones below the lowest differing bit
                    const uint32_t sc = (m & lenmask) | (0x40u | (32u - (kk0 + u)));
                    const uint32_t s = (w[u].x == p0 && n >= kb + kk0 + u) ? sc : 0u;
                    bb = max(bb, s);
                    dmask = __builtin_amdgcn_alignbit(dmask, s, 31);
                }
                kdone = kk0 + 3;
                // When most lanes are waiting for the window anyway (runs, long repeats), one round
                // serves them all: do it now; a match of `nice` bytes then ends the walk early.
                if ((kk0 & 7u) == 5u && __popcll(__ballot(dmask != 0)) >= 40) tile_end();
            }
            tile_end();
            twp -= 32;
            tsp -= 32;
        }
        if (!qsnap) qkey = key;
        if (active) {
            // key -> record: len << 16 | dist - 1, dist = 65535 - low half
            const uint32_t rf = (key >> 16) ? ((key & 0xffff0000u) | (0xfffeu - (key & 0xffffu))) : 0u;
            const uint32_t rq = (qkey >> 16) ? ((qkey & 0xffff0000u) | (0xfffeu - (qkey & 0xffffu))) : 0u;
            rec2[(prm.dbg & 1) ? (i & 1023u) : p] = make_uint2(rf, rq);
        }
        // ---- roll the buffer: the last 128 entries move down by 64
        fl_lds_order();
        {
            const uint2 a = rw[64 + lane];
            const uint16_t sa = rs[64 + lane];
            fl_lds_order();
            rw[lane] = a;
            rw[64 + lane] = make_uint2(p0, p1);
            rs[lane] = sa;
            rs[64 + lane] = (uint16_t)p;
        }
        fl_lds_order();
    }
    fl_prof_mark(11);
}
