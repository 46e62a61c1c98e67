// kernels_match.h -- the match finder of levels 4..9 (second generation).
//
// Reference path: Deflate.findMatch (deflate.zig:233-266) with SlidingWindow.match
// (SlidingWindow.zig:81-104) over the hash chains of Lookup (Lookup.zig:12-84), for EVERY position:
// rec[p] = { record for the full chain budget, record for chain >> 2 (deflate.zig:241-245) }.
//
// Input: S = the chunk's positions sorted by (hash, position) (k_lz_sort): the chain candidates
// of a sorted entry are the entries just before it in its bucket, nearest first.
//
// Work decomposition (one workgroup per chunk / stream tile, the window staged in LDS):
//  * a wave takes a SLICE of 256 consecutive sorted entries at a time.  It loads a tile of the
//    128 entries before the slice and the slice itself with the first 8 window bytes of each
//    (one 8-byte LDS word per entry), so comparing an entry with a candidate is one ds_read_b64.
//  * the walk is cut into UNITS of G (8; 4 at level 4) consecutive candidates.  Round t of an
//    epoch handles candidates G t + 1 .. G t + G of every entry that still has that many: the
//    entries of the slice are ordered by their number of units, so the active ones are a prefix,
//    and 64 of them fill a wave whatever their position in the slice.  That is what the first
//    generation lacked: there a lane was tied to its entry for the whole walk and idled once its
//    (short) chain was done -- lane efficiency 57 % on text, 75-85 % here.
//  * a unit is evaluated branch-free on registers: per candidate two XORs, the trailing-equal-bytes
//    mask, a packed score (equal bytes, then nearest) and a max.  Only the winner of a unit
//    touches the key of its entry.  Units whose winner agrees in all 8 prefix bytes go to a
//    per-round queue; the queue is served by full waves that walk the window (SlidingWindow.match's
//    reject-on-one-compare, then the extension), nearest candidate first, so `nice` ends a walk
//    exactly where the reference ends it (deflate.zig:256-258).
//  * rounds are processed in order, so the state of an entry after round t is the state of the
//    reference's walk after G (t + 1) candidates: the chain >> 2 record is a snapshot.
//  * chains longer than 128 (levels 7..9) run in epochs of 128 candidates, each with its own tile.
//
// Bound: vector-ALU issue (about 8 instructions per candidate); the LDS carries one 8-byte read
// per candidate.  No MFMA: byte compares and maxima.
#pragma once
#include "kernels_common.h"
#include "kernels_lz.h"

// phase accounting of wave 0 of workgroup 0 (tuning aid, compiled in with -DFL_M2_PROF)
#ifdef FL_M2_PROF
#define M2_T0() uint64_t t_prof = __builtin_readcyclecounter()
#define M2_ACC(slot)                                                   \
    do {                                                               \
        const uint64_t t_now = __builtin_readcyclecounter();           \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += t_now - t_prof; \
        t_prof = t_now;                                                \
    } while (0)
#define M2_CNT(slot, v)                                                \
    do {                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += (v); \
    } while (0)
#else
#define M2_T0()
#define M2_ACC(slot)
#define M2_CNT(slot, v)
#endif
#define FL_M2_WAVES 12
#define FL_M2_THREADS (64 * FL_M2_WAVES)
#define FL_M2_SLICE 256u   // sorted entries per wave step
#define FL_M2_BACK 128u    // candidates per epoch = tile entries before the slice
#define FL_M2_TILE (FL_M2_BACK + FL_M2_SLICE)

struct fl_m2_wave {
    uint2 tW[FL_M2_TILE];       // first 8 window bytes of the tile's entries
    uint2 est[FL_M2_SLICE];     // per entry of the slice: x = position | candidates it may look at << 16
                                // (0 = the walk has ended), y = best match so far: len << 16 | 65535 - dist
    uint16_t tS[FL_M2_TILE];    // positions of the tile's entries
    uint16_t dq[FL_M2_SLICE];   // entries of this round whose unit has a candidate equal in 8 bytes
    uint8_t perm[FL_M2_SLICE];  // the slice's entries ordered by units left, most first
    uint32_t cnt[40];           // bins of the counting sort; afterwards cnt[b] = entries with >= b units
};

// trailing-equal-bytes score of one candidate: 0 when the first four bytes differ, else
// (0x80 per further equal byte, low byte first) | 0x20 | tie, cut to the bytes `lenmask` allows
__device__ __forceinline__ uint32_t fl_m2_score(uint2 w, uint32_t p0, uint32_t p1, uint32_t lenmask, uint32_t tie) {
    const uint32_t x0 = w.x ^ p0, x1 = w.y ^ p1;
    const uint32_t m = ~x1 & (x1 - 1u);  // ones below the lowest differing bit
    const uint32_t s = (m & lenmask) | tie;
    return x0 == 0 ? s : 0u;
}

// exact common prefix of the window at p and q, known to be >= len0, capped at maxlen
__device__ __forceinline__ uint32_t fl_extend_from(const uint32_t* win32, uint32_t p, uint32_t q, uint32_t len0,
                                                   uint32_t maxlen) {
    uint32_t len = len0;
    while (len < maxlen) {
        uint32_t a0, a1, b0, b1;
        fl_lds_load8(win32, p + len, a0, a1);
        fl_lds_load8(win32, q + len, b0, b1);
        const uint32_t y0 = a0 ^ b0, y1 = a1 ^ b1;
        if (y0) {
            len += (uint32_t)__builtin_ctz(y0) >> 3;
            break;
        }
        if (y1) {
            len += 4 + ((uint32_t)__builtin_ctz(y1) >> 3);
            break;
        }
        len += 8;
    }
    return min(len, maxlen);
}

struct fl_m2_ctx {
    const uint32_t* win32;
    const uint32_t* fp;
    uint32_t N, w0, in_len, n_flush, zone, nice;
    bool has_fl;
};

template <bool STREAM>
__device__ __forceinline__ uint32_t fl_m2_maxlen(const fl_m2_ctx& cx, uint32_t p) {
    uint32_t maxlen = min(cx.N - p, FL_MAX_MATCH);
    if (STREAM && cx.has_fl) maxlen = min(maxlen, fl_next_flush(cx.fp, cx.n_flush, cx.w0 + p, cx.in_len) - (cx.w0 + p));
    return maxlen;
}
// valid candidates: q >= 1 (position 0 is the chain's null, deflate.zig:248), p - q <= 32768
// (deflate.zig:250-251), beyond the slide zone only the upper half of the window
template <bool STREAM>
__device__ __forceinline__ uint32_t fl_m2_lov(const fl_m2_ctx& cx, uint32_t p) {
    uint32_t lov = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;
    if (STREAM && p >= cx.zone) lov = max(lov, FL_MAX_DIST + 1u);
    return lov;
}

// One round for NB x 64 entries of the slice (perm[b0 ..]): candidates kbase + k0 + 1 .. + G.
// The NB batches are independent chains of LDS reads; written side by side so that the loads of
// one overlap the arithmetic of the other.  Returns the new length of the deep queue.
template <bool STREAM, int G, int NB>
__device__ __forceinline__ uint32_t fl_m2_units(fl_m2_wave& W, const fl_m2_ctx& cx, uint32_t lane, uint32_t a,
                                                uint32_t A, uint32_t b0, uint32_t kbase, uint32_t k0, uint32_t dqn) {
    bool on[NB];
    uint32_t e[NB], slot0[NB];
    uint2 st[NB], cw[NB][G];
#pragma unroll
    for (int h = 0; h < NB; h++) {
        const uint32_t j = b0 + 64 * h + lane;
        on[h] = j < A;
        e[h] = on[h] ? W.perm[j] : 0u;
    }
#pragma unroll
    for (int h = 0; h < NB; h++) {
        st[h] = W.est[e[h]];
        slot0[h] = e[h] + FL_M2_BACK - k0 - G;  // slot of candidate k0 + G
#pragma unroll
        for (int u = 0; u < G; u++) cw[h][u] = W.tW[slot0[h] + G - 1 - u];
    }
    uint32_t p[NB], nrel[NB], maxlen[NB], lov[NB], p0[NB], p1[NB], qc[NB];
    bool live[NB];
#pragma unroll
    for (int h = 0; h < NB; h++) {
        p[h] = st[h].x & 0xffffu;
        const uint32_t n = on[h] ? st[h].x >> 16 : 0u;  // 0: the walk of this entry has ended
        nrel[h] = n > kbase ? n - kbase : 0u;            // epoch-relative candidates allowed
        live[h] = nrel[h] > k0;
        fl_lds_load8(cx.win32, p[h], p0[h], p1[h]);
        // the farthest candidate of this unit that the count allows decides whether every
        // candidate of the unit passes the position bound
        const uint32_t kc = live[h] ? min(nrel[h], k0 + G) : 1u;
        qc[h] = W.tS[e[h] + FL_M2_BACK - kc];
        maxlen[h] = fl_m2_maxlen<STREAM>(cx, p[h]);
        lov[h] = fl_m2_lov<STREAM>(cx, p[h]);
    }
#pragma unroll
    for (int h = 0; h < NB; h++) {
        // (slots below sorted index 0 hold position 0, which fails every bound)
        const bool trunc = live[h] && (qc[h] < lov[h] || a + e[h] < kbase + k0 + G);
        const uint32_t lenmask = maxlen[h] >= 8 ? 0x80808080u : (0x00808080u >> (8 * (7 - maxlen[h])));
        uint32_t best = 0;
        if (__any(trunc)) {
            // some lane's unit reaches below its position bound: check every candidate's position
            const uint16_t* cs = &W.tS[slot0[h]];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const uint32_t q = cs[G - 1 - u];
                uint32_t s = fl_m2_score(cw[h][u], p0[h], p1[h], lenmask, 0x20u | (uint32_t)(G - 1 - u));
                if (q < lov[h] || k0 + 1 + u > nrel[h]) s = 0;
                best = max(best, s);
            }
        } else {
#pragma unroll
            for (int u = 0; u < G; u++)
                best = max(best, fl_m2_score(cw[h][u], p0[h], p1[h], lenmask, 0x20u | (uint32_t)(G - 1 - u)));
        }
        if (!live[h]) best = 0;
        if (best) {
            const uint32_t uw = (G - 1) - (best & 7u);
            const uint32_t le = min(4u + (uint32_t)__popc(best & 0x80808080u), maxlen[h]);
            const uint32_t q = W.tS[slot0[h] + (G - 1) - uw];
            const uint32_t kcand = (le << 16) | (q + 0xffffu - p[h]);  // low half = 65535 - (p - q)
            if (kcand > st[h].y) {  // deflate.zig:254-261
                // nothing longer possible (le <= 8 < nice): the walk ends
                W.est[e[h]] = make_uint2(le >= maxlen[h] ? p[h] : st[h].x, kcand);
            }
        }
        // a candidate equal in all 8 prefix bytes that may run on: the window decides
        const bool deep = best >= 0x80808080u && maxlen[h] > 8;
        const uint64_t dm = __ballot(deep);
        if (deep)
            W.dq[dqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] =
                (uint16_t)e[h];
        dqn += (uint32_t)__popcll(dm);
    }
    return dqn;
}

// Serve 64 entries of the deep queue: SlidingWindow.match against the window for the candidates of
// the unit that agree in 8 bytes, nearest first.
template <bool STREAM, int G>
__device__ __forceinline__ void fl_m2_deep(fl_m2_wave& W, const fl_m2_ctx& cx, uint32_t lane, uint32_t dqn,
                                           uint32_t b0, uint32_t kbase, uint32_t k0) {
    const uint32_t* win32 = cx.win32;
    const uint32_t j = b0 + lane;
    const bool on = j < dqn;
    const uint32_t e = on ? W.dq[j] : 0u;
    const uint2 st = W.est[e];
    const uint32_t slot0 = e + FL_M2_BACK - k0 - G;
    const uint2* cw = &W.tW[slot0];
    const uint16_t* cs = &W.tS[slot0];
    uint2 w[G];
    uint32_t qs[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
        w[u] = cw[G - 1 - u];
        qs[u] = cs[G - 1 - u];
    }
    const uint32_t p = st.x & 0xffffu;
    const uint32_t n = on ? st.x >> 16 : 0u;
    const uint32_t nrel = n > kbase ? n - kbase : 0u;
    // window bytes p .. p+15
    uint32_t p0, p1, pA, pB;
    {
        const uint32_t i = p >> 2, sh = p & 3;
        const uint32_t d0 = win32[i], d1 = win32[i + 1], d2 = win32[i + 2], d3 = win32[i + 3], d4 = win32[i + 4];
        p0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        p1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        pA = __builtin_amdgcn_alignbyte(d3, d2, sh);
        pB = __builtin_amdgcn_alignbyte(d4, d3, sh);
    }
    const uint32_t maxlen = fl_m2_maxlen<STREAM>(cx, p);
    const uint32_t lov = fl_m2_lov<STREAM>(cx, p);
    uint32_t dmask = 0;  // bit u: candidate k0 + 1 + u is valid and agrees in 8 bytes
#pragma unroll
    for (int u = 0; u < G; u++)
        if (((w[u].x ^ p0) | (w[u].y ^ p1)) == 0 && qs[u] >= lov && k0 + 1 + u <= nrel) dmask |= 1u << u;
    if (nrel <= k0) dmask = 0;
    uint32_t key = st.y;
    const uint32_t cp = 0xffffu - p;
    uint32_t pb = 0;  // window bytes p+best-3 .. p+best (valid when best >= 16)
    if ((key >> 16) >= 16) pb = fl_lds_load4(win32, p + (key >> 16) - 3);
    bool stop = false;
    while (__any(dmask != 0)) {
        M2_CNT(43, 1);
        if (dmask) {
            const uint32_t u = (uint32_t)__builtin_ctz(dmask);  // nearest first
            dmask &= dmask - 1;
            const uint32_t q = cs[G - 1 - u];
            const uint32_t bestl = key >> 16;
            // bytes 8 .. 15 of the candidate, and the reference's reject-on-one-compare
            // (SlidingWindow.zig:91-98) once the best match is longer than that
            uint32_t qA, qB;
            fl_lds_load8(win32, q + 8, qA, qB);
            bool take = maxlen > bestl;
            if (take && bestl >= 16) take = fl_lds_load4(win32, q + bestl - 3) == pb;
            if (take) {
                const uint32_t yA = qA ^ pA, yB = qB ^ pB;
                uint32_t le = 8;
                if (yA)
                    le += (uint32_t)__builtin_ctz(yA) >> 3;
                else if (yB)
                    le += 4 + ((uint32_t)__builtin_ctz(yB) >> 3);
                else
                    le = fl_extend_from(win32, p, q, 16, maxlen);
                le = min(le, maxlen);
                const uint32_t kcand = (le << 16) | (q + cp);
                if (kcand > key) {  // deflate.zig:254-261
                    key = kcand;
                    if (le >= 16) pb = fl_lds_load4(win32, p + le - 3);
                    if (le >= maxlen || le >= cx.nice) {  // nothing longer possible / stop looking
                        stop = true;
                        dmask = 0;
                    }
                }
            }
        }
    }
    if (on) W.est[e] = make_uint2(stop ? p : st.x, key);
}

template <bool STREAM, int G>
__global__ __launch_bounds__(FL_M2_THREADS) void k_lz_match2(const uint8_t* __restrict__ in,
                                                             const fl_chunk* __restrict__ chunks,
                                                             const fl_tile* __restrict__ tiles,
                                                             const uint32_t* __restrict__ fpts,
                                                             const uint32_t* __restrict__ n_sorted, fl_params prm,
                                                             const uint16_t* __restrict__ S,
                                                             uint32_t* __restrict__ rec_all) {
    constexpr uint32_t WIN_DW = STREAM ? FL_WIN_DW_STREAM : FL_WIN_DW_CHUNK;
    constexpr uint32_t RPE = FL_M2_BACK / G;  // rounds per epoch
    __shared__ uint32_t win32[WIN_DW];
    __shared__ uint32_t bmask[2048 + 2];  // bit i: sorted entry i starts a bucket
    __shared__ fl_m2_wave wv[FL_M2_WAVES];
    const uint32_t c = blockIdx.x;
    const uint32_t w0 = STREAM ? tiles[c].w0 : 0u;
    const uint32_t tgt0 = STREAM ? tiles[c].tgt0 : 0u;
    const fl_chunk ck = chunks[STREAM ? tiles[c].chunk : c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = ck.in_len - w0;
    const uint32_t Mpos = min(N >= 4 ? N - 3 : 0u, 65536u);  // positions with 4 bytes left in the stream
    const uint32_t M = STREAM ? n_sorted[c] : Mpos;            // entries of the sorted array
    fl_m2_ctx cx;
    cx.win32 = win32;
    cx.fp = STREAM ? fpts + ck.flush_off : nullptr;
    cx.N = N;
    cx.w0 = w0;
    cx.in_len = ck.in_len;
    cx.n_flush = ck.n_flush;
    cx.zone = STREAM ? tiles[c].zone : 65536u;
    cx.nice = prm.nice;
    // a flush point up to 258 bytes past the last position still shortens matches in this window
    cx.has_fl =
        STREAM && ck.n_flush && fl_next_flush(cx.fp, ck.n_flush, w0, ck.in_len) <= w0 + Mpos + 2 + FL_MAX_MATCH;
    const uint8_t* src = in + ck.in_off + w0;
    const uint16_t* Sc = S + (uint64_t)c * FL_CHUNK_STRIDE;
    uint2* rec2 = (uint2*)rec_all + ck.pos_off + w0;
    const uint32_t chain = prm.chain, quarter = prm.chain >> 2;

    fl_prof_mark(8);
    // stage the window in LDS (zero padded)
    const uint32_t ndw = (min(N, WIN_DW * 4u) + 3) >> 2;
    for (uint32_t i = tid; i < WIN_DW; i += FL_M2_THREADS)
        win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, N) : 0u;
    // positions without a hash entry never match (Lookup.zig:24)
    // (with flush points in the stream the host has cleared all records beforehand)
    for (uint32_t p = Mpos + tid; p < min(N, 65536u); p += FL_M2_THREADS) rec2[p] = make_uint2(0u, 0u);
    __syncthreads();
    fl_prof_mark(9);

    // ---- bucket starts: bit i set iff entry i is the first of its hash bucket ----
    {
        const uint32_t ngrp = (M + 63) >> 6;
        for (uint32_t g0 = wave * 4; g0 < ngrp; g0 += FL_M2_WAVES * 4) {
            uint32_t q[4], qp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = ((g0 + u) << 6) + lane;
                q[u] = i < M ? Sc[i] : 0u;
                qp[u] = (i < M && i) ? Sc[i - 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = ((g0 + u) << 6) + lane;
                const uint32_t h = fl_hash_le(fl_lds_load4(win32, q[u]));
                const uint32_t hp = fl_hash_le(fl_lds_load4(win32, qp[u]));
                const uint64_t st = __ballot(i < M && (i == 0 || h != hp));
                if (lane == 0 && g0 + u < ngrp) {
                    bmask[2 * (g0 + u)] = (uint32_t)st;
                    bmask[2 * (g0 + u) + 1] = (uint32_t)(st >> 32);
                }
            }
        }
    }
    __syncthreads();
    fl_prof_mark(10);

    fl_m2_wave& W = wv[wave];
#ifdef FL_M2_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 32; k < 48; k++) g_fl_prof[k] = 0;
#endif
    M2_T0();
    const uint32_t nslices = (M + FL_M2_SLICE - 1) / FL_M2_SLICE;
    // positions of the first tile of this wave's next slice, fetched one slice ahead
    uint32_t nxq[6];
    {
        const int32_t base = (int32_t)(wave * FL_M2_SLICE) - (int32_t)FL_M2_BACK;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int32_t i = base + 64 * r + (int32_t)lane;
            nxq[r] = (wave < nslices && i >= 0 && i < (int32_t)M) ? Sc[i] : 0u;
        }
    }
    for (uint32_t slice = wave; slice < nslices; slice += FL_M2_WAVES) {
        const uint32_t a = slice * FL_M2_SLICE;
        uint32_t tq[6];
#pragma unroll
        for (int r = 0; r < 6; r++) tq[r] = nxq[r];
        {
            const uint32_t sn = slice + FL_M2_WAVES;
            const int32_t base = (int32_t)(sn * FL_M2_SLICE) - (int32_t)FL_M2_BACK;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const int32_t i = base + 64 * r + (int32_t)lane;
                nxq[r] = (sn < nslices && i < (int32_t)M) ? Sc[i] : 0u;
            }
        }
        uint32_t qk[4] = {0, 0, 0, 0};
        bool qsnap = false;
        uint32_t nepoch = 1;
        for (uint32_t ep = 0; ep < nepoch; ep++) {
            const uint32_t kbase = ep * FL_M2_BACK;  // candidates kbase + 1 .. kbase + 128
            fl_lds_order();
            // ---- tile: sorted entries [a - kbase - 128, a - kbase + 256) with their 8 prefix bytes
            // (slots below index 0 hold position 0, which no walk accepts, deflate.zig:248) ----
            if (ep) {
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const int32_t i = (int32_t)a - (int32_t)kbase - (int32_t)FL_M2_BACK + 64 * r + (int32_t)lane;
                    tq[r] = (i >= 0 && i < (int32_t)M) ? Sc[i] : 0u;
                }
            }
#pragma unroll
            for (int r = 0; r < 6; r++) {
                uint32_t a0, a1;
                fl_lds_load8(win32, tq[r], a0, a1);
                W.tS[64 * r + lane] = (uint16_t)tq[r];
                W.tW[64 * r + lane] = make_uint2(a0, a1);
            }
            M2_ACC(33);
            if (ep == 0) {
                // ---- the slice's own entries: position, candidates they may look at ----
                uint32_t maxu = 0;
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    const uint32_t e = (g << 6) + lane, i = a + e;
                    const uint32_t p = tq[2 + g];
                    uint32_t n = 0;
                    if (i < M && (!STREAM || p >= tgt0)) {
                        // bucket offset = distance to the nearest bucket start at or before i, capped at chain
                        uint32_t wd = i >> 5;
                        uint32_t bits = bmask[wd] & (0xffffffffu >> (31 - (i & 31)));
                        uint32_t o;
                        for (;;) {
                            if (bits) {
                                o = i - ((wd << 5) + 31 - (uint32_t)__builtin_clz(bits));
                                break;
                            }
                            if (wd == 0 || i - (wd << 5) >= chain) {
                                o = chain;
                                break;
                            }
                            wd--;
                            bits = bmask[wd];
                        }
                        n = min(o, chain);
                    }
                    W.est[e] = make_uint2(p | (n << 16), 0u);
                    maxu = max(maxu, (n + G - 1) / G);
                }
                maxu = fl_wave_max(maxu);
                nepoch = max(1u, (maxu + RPE - 1) / RPE);
                M2_ACC(32);
            }
            // ---- order the entries by the units they have left in this epoch (most first) ----
            if (lane < 40) W.cnt[lane] = 0;
            fl_lds_order();
            uint32_t ub[4];
#pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                const uint32_t n = W.est[(g << 6) + lane].x >> 16;
                const uint32_t left = n > kbase ? n - kbase : 0u;
                ub[g] = min((left + G - 1) / G, RPE);
                if (ub[g]) atomicAdd(&W.cnt[ub[g]], 1u);
            }
            fl_lds_order();
            {
                // suffix sums over the bins: lane b gets the number of entries with more units than b
                const uint32_t cb = lane <= RPE ? W.cnt[lane] : 0u;
                uint32_t suf = cb;  // sum over lanes >= this one
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_down(suf, d, 64);
                    if (lane + d < 64) suf += t;
                }
                fl_lds_order();
                // cnt[b] = first slot of bin b (entries with more units come first)
                if (lane <= RPE) W.cnt[lane] = suf - cb;
            }
            fl_lds_order();
#pragma unroll
            for (uint32_t g = 0; g < 4; g++)
                if (ub[g]) W.perm[atomicAdd(&W.cnt[ub[g]], 1u)] = (uint8_t)((g << 6) + lane);
            fl_lds_order();
            // now cnt[b] = end of bin b = number of entries with at least b units
            M2_ACC(34);

            // ---- rounds ----
#pragma unroll 1
            for (uint32_t t = 0; t < RPE; t++) {
                const uint32_t A = __builtin_amdgcn_readfirstlane(W.cnt[t + 1]);  // entries with more than t units
                if (A == 0) break;
                const uint32_t k0 = G * t;  // this round: epoch-relative candidates k0 + 1 .. k0 + G
                uint32_t dqn = 0;
                uint32_t b0 = 0;
#pragma unroll 1
                for (; b0 + 64 < A; b0 += 128) dqn = fl_m2_units<STREAM, G, 2>(W, cx, lane, a, A, b0, kbase, k0, dqn);
                if (b0 < A) dqn = fl_m2_units<STREAM, G, 1>(W, cx, lane, a, A, b0, kbase, k0, dqn);
                fl_lds_order();
                M2_ACC(35);
                M2_CNT(40, (A + 63) / 64);
                M2_CNT(41, (dqn + 63) / 64);
                M2_CNT(42, dqn);
#pragma unroll 1
                for (uint32_t d0 = 0; d0 < dqn; d0 += 64) fl_m2_deep<STREAM, G>(W, cx, lane, dqn, d0, kbase, k0);
                fl_lds_order();
                M2_ACC(36);
                // the chain >> 2 budget (deflate.zig:241-245) ends with this round?
                if (kbase + k0 + G == quarter) {
#pragma unroll
                    for (uint32_t g = 0; g < 4; g++) qk[g] = W.est[(g << 6) + lane].y;
                    qsnap = true;
                }
            }
        }
        fl_lds_order();
        // ---- records of the slice's entries ----
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t e = (g << 6) + lane, i = a + e;
            const uint2 st = W.est[e];
            const uint32_t p = st.x & 0xffffu;
            if (i < M && (!STREAM || p >= tgt0)) {
                const uint32_t key = st.y;
                const uint32_t qkey = qsnap ? qk[g] : key;
                // key -> record: len << 16 | dist - 1, dist = 65535 - low half
                const uint32_t rf = (key >> 16) ? ((key & 0xffff0000u) | (0xfffeu - (key & 0xffffu))) : 0u;
                const uint32_t rq = (qkey >> 16) ? ((qkey & 0xffff0000u) | (0xfffeu - (qkey & 0xffffu))) : 0u;
                rec2[p] = make_uint2(rf, rq);
            }
        }
        M2_ACC(37);
    }
    fl_prof_mark(11);
}
