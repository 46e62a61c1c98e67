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
//    (one 8-byte LDS word per entry), so comparing an entry with a candidate is one ds_read_b64,
//    and consecutive lanes read consecutive words (no bank conflicts).
//  * lane = entry, 64 entries at a time; the walk runs in BLOCKS of 32 candidates, branch-free on
//    registers: per candidate two XORs, the trailing-equal-bytes mask, the position bound, a packed
//    score (equal bytes, then nearest), a max, and one bit "agrees in all 8 prefix bytes" shifted
//    into a per-lane mask.  The first generation spent 25-30 instructions per candidate on lane-mask
//    algebra; this is 9.
//  * at the end of a block (and where the chain >> 2 budget ends) the block's winner meets the
//    lane's key, then the lanes serve their masks: SlidingWindow.match against the window (reject
//    on one compare, then the extension), nearest candidate first, so `nice` ends a walk exactly
//    where the reference ends it (deflate.zig:256-258).  Inside a block the order of evaluation does
//    not matter: the key is a maximum, and an 8-byte candidate never beats one that reached `nice`
//    (>= 16).  Blocks are processed in order, so the chain >> 2 record is a snapshot.
//  * chains longer than 128 (levels 7..9) run in epochs of 128 candidates, each with its own tile;
//    the lanes' state stays in registers.
//
// Bound: vector-ALU issue; the LDS carries one 8-byte and one 2-byte read per candidate.  No MFMA:
// byte compares and maxima.
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
#define FL_M2_SLICE 256u   // sorted entries per wave step
#define FL_M2_BACK 128u    // candidates per epoch = tile entries before the slice
#define FL_M2_TILE (FL_M2_BACK + FL_M2_SLICE)
#define FL_M2_WAVES 12
#define FL_M2_THREADS (64 * FL_M2_WAVES)

struct fl_m2_wave {
    uint2 tW[FL_M2_TILE];     // first 8 window bytes of the tile's entries
    uint16_t tS[FL_M2_TILE];  // positions of the tile's entries
};

// exact common prefix of the window at p and q, known to be >= len0, capped at maxlen
__device__ __forceinline__ uint32_t fl_extend_from(const uint32_t* win32, uint32_t p, uint32_t q, uint32_t len0,
                                                   uint32_t maxlen) {
    return fl_extend_len(win32, p, q, len0, maxlen);
}

// the walk of one lane (= one sorted entry)
struct fl_m2_lane {
    uint32_t e;       // entry of the slice
    uint32_t p;       // its position
    uint32_t p0, p1;  // window bytes p .. p+7
    uint32_t n;       // candidates it may look at in total; 0: the walk has ended
    uint32_t key;     // best match so far: len << 16 | 65535 - (number of the candidate along the chain)
    uint32_t maxlen, lov, lenmask;
    uint32_t dm;      // bit b: candidate kdone - b agrees in 8 bytes and waits for the window
};

// Serve the lanes' masks of 8-byte candidates: SlidingWindow.match against the window, nearest
// first.  kdone = epoch-relative number of the last candidate shifted into the masks.
__device__ __forceinline__ void fl_m2_deep(fl_m2_lane& L, const uint32_t* win32, const uint16_t* tS, uint32_t kbase,
                                           uint32_t kdone, uint32_t nice, uint32_t dbg = 0) {
    if (!__any(L.dm != 0)) return;
    if (dbg & 4) { L.dm = 0; return; }  // timing experiment: no window walks (wrong output)
    // window bytes p+8 .. p+15
    uint32_t pA, pB;
    fl_lds_load8(win32, L.p + 8, pA, pB);
    uint32_t pb = 0;  // window bytes p+best-3 .. p+best (valid when best >= 16)
    if ((L.key >> 16) >= 16) pb = fl_lds_load4(win32, L.p + (L.key >> 16) - 3);
    while (__any(L.dm != 0)) {
        M2_CNT(43, 1);
        if (L.dm) {
            const uint32_t b = 31u - (uint32_t)__builtin_clz(L.dm);  // nearest first
            L.dm &= ~(1u << b);
            const uint32_t q = tS[L.e + FL_M2_BACK - (kdone - b)];
            const uint32_t bestl = L.key >> 16;
            // bytes 8 .. 15 of the candidate, and the reference's reject-on-one-compare
            // (SlidingWindow.zig:91-98) once the best match is longer than that
            uint32_t qA, qB;
            fl_lds_load8(win32, q + 8, qA, qB);
            bool take = L.maxlen > bestl;
            if (take && bestl >= 16) take = fl_lds_load4(win32, q + bestl - 3) == pb;
            if (take) {
                const uint32_t yA = qA ^ pA, yB = qB ^ pB;
                uint32_t le = 8;
                if (yA)
                    le += (uint32_t)__builtin_ctz(yA) >> 3;
                else if (yB)
                    le += 4 + ((uint32_t)__builtin_ctz(yB) >> 3);
                else
                    le = fl_extend_from(win32, L.p, q, 16, L.maxlen);
                le = min(le, L.maxlen);
                const uint32_t kcand = (le << 16) | (0xffffu - (kbase + kdone - b));
                if (kcand > L.key) {  // deflate.zig:254-261
                    L.key = kcand;
                    if (le >= 16) pb = fl_lds_load4(win32, L.p + le - 3);
                    if (le >= L.maxlen || le >= nice) {  // nothing longer possible / stop looking
                        L.n = 0;
                        L.dm = 0;
                    }
                }
            }
        }
    }
}

// key -> record (len << 16 | dist - 1).  A key whose low half is 65535 was not beaten in this
// epoch: the record it came with stays.
__device__ __forceinline__ uint32_t fl_m2_record(const fl_m2_lane& L, const uint16_t* tS, uint32_t kbase,
                                                 uint32_t oldrec) {
    const uint32_t le = L.key >> 16, low = L.key & 0xffffu;
    if (le == 0 || low == 0xffffu) return oldrec;
    const uint32_t q = tS[L.e + FL_M2_BACK - ((0xffffu - low) - kbase)];
    return (le << 16) | (L.p - q - 1u);
}

__device__ __forceinline__ uint32_t fl_sel4(const uint32_t (&v)[4], uint32_t g) {
    uint32_t r = v[0];
    r = g == 1 ? v[1] : r;
    r = g == 2 ? v[2] : r;
    r = g == 3 ? v[3] : r;
    return r;
}
__device__ __forceinline__ void fl_put4(uint32_t (&v)[4], uint32_t g, uint32_t x) {
    v[0] = g == 0 ? x : v[0];
    v[1] = g == 1 ? x : v[1];
    v[2] = g == 2 ? x : v[2];
    v[3] = g == 3 ? x : v[3];
}

template <bool STREAM, int G>
__global__ __launch_bounds__(FL_M2_THREADS) void k_lz_match2(
    const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks, const fl_tile* __restrict__ tiles,
    const uint32_t* __restrict__ fpts, const uint32_t* __restrict__ n_sorted, fl_params prm,
    const uint16_t* __restrict__ S, uint32_t* __restrict__ rec_all) {
    constexpr uint32_t NW = FL_M2_WAVES;
    constexpr uint32_t NT = 64 * NW;
    constexpr uint32_t WIN_DW = STREAM ? FL_WIN_DW_STREAM : FL_WIN_DW_CHUNK;
    __shared__ uint32_t win32[WIN_DW];
    __shared__ uint32_t bmask[2048 + 2];  // bit i: sorted entry i starts a bucket
    __shared__ uint16_t gcarry[1024 + 2];  // last bucket start before group g of 64 sorted entries
    __shared__ fl_m2_wave wv[NW];
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
    for (uint32_t i = tid; i < WIN_DW; i += NT) win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, N) : 0u;
    // positions without a hash entry never match (Lookup.zig:24)
    // (with flush points in the stream the host has cleared all records beforehand)
    for (uint32_t p = Mpos + tid; p < min(N, 65536u); p += NT) rec2[p] = make_uint2(0u, 0u);
    __syncthreads();
    fl_prof_mark(9);

    // ---- bucket starts: bit i set iff entry i is the first of its hash bucket ----
    {
        const uint32_t ngrp = (M + 63) >> 6;
        for (uint32_t g0 = wave * 4; g0 < ngrp; g0 += NW * 4) {
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
    // the last bucket start before every group of 64 entries: a running maximum over the groups
    if (wave == 0) {
        const uint32_t ngrp = (M + 63) >> 6;
        uint32_t run = 0;
        for (uint32_t g0 = 0; g0 < ngrp; g0 += 64) {
            const uint32_t g = g0 + lane;
            uint32_t last = 0;
            if (g < ngrp) {
                const uint32_t lo = bmask[2 * g], hi = bmask[2 * g + 1];
                if (hi)
                    last = (g << 6) + 63 - (uint32_t)__builtin_clz(hi);
                else if (lo)
                    last = (g << 6) + 31 - (uint32_t)__builtin_clz(lo);
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(last, d, 64);
                if (lane >= (uint32_t)d) last = max(last, t);
            }
            last = max(last, run);
            if (g < ngrp) gcarry[g + 1] = (uint16_t)last;
            run = __shfl(last, 63, 64);
        }
        if (lane == 0) gcarry[0] = 0;
    }
    __syncthreads();
    fl_prof_mark(10);

    fl_m2_wave& W = wv[wave];
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
    for (uint32_t slice = wave; slice < nslices; slice += NW) {
        const uint32_t a = slice * FL_M2_SLICE;
        uint32_t tq[6];
#pragma unroll
        for (int r = 0; r < 6; r++) tq[r] = nxq[r];
        {
            const uint32_t sn = slice + NW;
            const int32_t base = (int32_t)(sn * FL_M2_SLICE) - (int32_t)FL_M2_BACK;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const int32_t i = base + 64 * r + (int32_t)lane;
                nxq[r] = (sn < nslices && i < (int32_t)M) ? Sc[i] : 0u;
            }
        }
        // the state of the slice's entries: group g of 64, lane = entry
        uint32_t sp[4], sp0[4], sp1[4];  // position, window bytes p .. p+7
        uint32_t sn_[4];                 // candidates the entry may look at in total; 0: the walk has ended
        uint32_t srec[4], sqrec[4];      // records so far: full budget, chain >> 2 budget
        uint32_t sflag[4];               // bit 0: the entry gets a record; bit 1: chain >> 2 snapshot taken
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
                if (ep == 0 && r >= 2) {
                    sp[r - 2] = tq[r];
                    sp0[r - 2] = a0;
                    sp1[r - 2] = a1;
                }
            }
            if (ep == 0) {
                // ---- the slice's own entries: candidates they may look at ----
                uint32_t maxn = 0;
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    const uint32_t i = a + (g << 6) + lane;
                    const uint32_t p = sp[g];
                    uint32_t n = 0;
                    const bool has = i < M && (!STREAM || p >= tgt0);
                    if (has) {
                        // bucket offset = distance to the nearest bucket start at or before i, capped at chain
                        const uint32_t gi = i >> 6;
                        const uint32_t blo = bmask[2 * gi], bhi = bmask[2 * gi + 1];
                        const uint32_t mlo = blo & (lane < 32 ? (0xffffffffu >> (31 - lane)) : 0xffffffffu);
                        const uint32_t mhi = lane < 32 ? 0u : (bhi & (0xffffffffu >> (63 - lane)));
                        uint32_t st;
                        if (mhi)
                            st = (gi << 6) + 63 - (uint32_t)__builtin_clz(mhi);
                        else if (mlo)
                            st = (gi << 6) + 31 - (uint32_t)__builtin_clz(mlo);
                        else
                            st = gcarry[gi];
                        n = min(i - st, chain);
                    }
                    sn_[g] = n;
                    srec[g] = 0;
                    sqrec[g] = 0;
                    sflag[g] = has ? 1u : 0u;
                    maxn = max(maxn, n);
                }
                maxn = fl_wave_max(maxn);
                nepoch = max(1u, (maxn + FL_M2_BACK - 1) / FL_M2_BACK);
            }
            fl_lds_order();

            // ---- the walks: lane = entry, 64 entries at a time ----
#pragma unroll 1
            for (uint32_t g = 0; g < 4; g++) {
                fl_m2_lane L;
                L.e = (g << 6) + lane;
                L.n = fl_sel4(sn_, g);
                uint32_t nrel = L.n > kbase ? L.n - kbase : 0u;  // candidates of this epoch the lane may look at
                const uint32_t kmax = min(fl_wave_max(nrel), FL_M2_BACK);
                if (kmax == 0) continue;
                L.p = fl_sel4(sp, g);
                L.p0 = fl_sel4(sp0, g);
                L.p1 = fl_sel4(sp1, g);
                const uint32_t oldrec = fl_sel4(srec, g);  // the record the earlier epochs left
                L.key = ep ? ((oldrec & 0xffff0000u) | 0xffffu) : 0u;  // (wins every tie: it is nearer)
                uint32_t qrec = fl_sel4(sqrec, g), flag = fl_sel4(sflag, g);
                L.maxlen = min(N - L.p, FL_MAX_MATCH);
                if (STREAM && has_fl) L.maxlen = min(L.maxlen, fl_next_flush(fp, ck.n_flush, w0 + L.p, ck.in_len) - (w0 + L.p));
                // valid candidates: q >= 1 (position 0 is the chain's null, deflate.zig:248), p - q <= 32768
                // (deflate.zig:250-251), beyond the slide zone only the upper half of the window
                L.lov = L.p > FL_MAX_DIST ? L.p - FL_MAX_DIST : 1u;
                if (STREAM && L.p >= zone) L.lov = max(L.lov, FL_MAX_DIST + 1u);
                L.lenmask = L.maxlen >= 8 ? 0x80808080u : (0x00808080u >> (8 * (7 - L.maxlen)));
                L.dm = 0;
                uint32_t kdone = 0;
                uint32_t bb = 0;  // best score of the current block of 32 candidates
                // what a block (or the chain >> 2 budget) ends with: the block's winner meets the key,
                // then the 8-byte candidates meet the window
                auto block_end = [&](uint32_t kb) {
                    if (__any(bb != 0)) {
                        if (bb) {
                            const uint32_t le = min(4u + (uint32_t)__popc(bb & 0x80808080u), L.maxlen);
                            const uint32_t kcand = (le << 16) | (0xffffu - (kbase + kb + 32u - (bb & 31u)));
                            if (kcand > L.key) {  // deflate.zig:254-261
                                L.key = kcand;
                                if (le >= L.maxlen) {  // nothing longer possible (le <= 8 < nice): the walk ends
                                    L.n = 0;
                                    L.dm = 0;
                                }
                            }
                        }
                        bb = 0;
                    }
                    if (L.maxlen <= 8) L.dm = 0;
                    fl_m2_deep(L, win32, W.tS, kbase, kdone, nice, prm.dbg);
                    if (L.n == 0) nrel = 0;
                    if (kbase + kdone == quarter) {  // the chain >> 2 budget (deflate.zig:241-245) ends here
                        qrec = fl_m2_record(L, W.tS, kbase, oldrec);
                        flag |= 2u;
                    }
                };
                // tile words and positions of the first unit; each unit fetches the next one's before it computes
                uint2 cwn[G];
                uint32_t cqn[G];
                {
                    const uint2* cwp = &W.tW[L.e + FL_M2_BACK - G];
                    const uint16_t* cqp = &W.tS[L.e + FL_M2_BACK - G];
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        cwn[u] = cwp[G - 1 - u];
                        cqn[u] = cqp[G - 1 - u];
                    }
                }
#pragma unroll 1
                for (uint32_t kb = 0; kb < kmax; kb += 32) {  // a block: candidates kbase + kb + 1 .. + 32
#pragma unroll
                    for (uint32_t t = 0; t < 32 / G; t++) {
                        const uint32_t k0 = kb + G * t;  // this unit: candidates kbase + k0 + 1 .. + G
                        if (k0 < kmax) {
                            // a lane whose count ends inside the unit needs no mask: what follows in the
                            // tile are entries of other buckets, which differ in their first four bytes
                            // (the hash is a function of those), or slots below sorted index 0, which
                            // hold position 0 and fail the bound; a chain budget ends on a unit boundary
                            const uint32_t lovu = nrel > k0 ? L.lov : 0xffffffffu;
                            uint2 cw[G];
                            uint32_t cq[G];
#pragma unroll
                            for (int u = 0; u < G; u++) {
                                cw[u] = cwn[u];
                                cq[u] = cqn[u];
                            }
                            {
                                const uint32_t kn = min(k0 + G, FL_M2_BACK - G);
                                const uint2* cwp = &W.tW[L.e + FL_M2_BACK - kn - G];
                                const uint16_t* cqp = &W.tS[L.e + FL_M2_BACK - kn - G];
#pragma unroll
                                for (int u = 0; u < G; u++) {
                                    cwn[u] = cwp[G - 1 - u];
                                    cqn[u] = cqp[G - 1 - u];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < G; u++) {
                                // trailing-equal-bytes score: 0 when the first four bytes differ or the
                                // position is out of bounds, else (0x80 per further equal byte, low byte
                                // first) | 0x40 | 31 - number in the block (nearest wins a tie)
                                const uint32_t x0 = cw[u].x ^ L.p0, x1 = cw[u].y ^ L.p1;
                                const uint32_t m = ~x1 & (x1 - 1u);  // ones below the lowest differing bit
                                const uint32_t sc = (m & L.lenmask) | (0x40u | (31u - (G * t + u)));
                                const uint32_t s = (x0 == 0 && cq[u] >= lovu) ? sc : 0u;
                                bb = max(bb, s);
                                L.dm = __builtin_amdgcn_alignbit(L.dm, s, 31);
                            }
                            kdone = k0 + G;
                            if (t + 1 < 32 / G && kbase + kdone == quarter) block_end(kb);
                        }
                    }
                    block_end(kb);
                    if (!__any(nrel > kdone)) break;  // every walk of the group has ended
                }
                // the lane's state waits for the next epoch
                const bool more = L.n > kbase + FL_M2_BACK;
                fl_put4(srec, g, fl_m2_record(L, W.tS, kbase, oldrec));
                fl_put4(sn_, g, more ? L.n : 0u);
                fl_put4(sqrec, g, qrec);
                fl_put4(sflag, g, flag);
            }
        }
        // ---- records of the slice's entries (a walk that ended before the chain >> 2 budget did has
        // the same record twice) ----
#pragma unroll
        for (uint32_t g = 0; g < 4; g++)
            if (sflag[g] & 1u) rec2[sp[g]] = make_uint2(srec[g], (sflag[g] & 2u) ? sqrec[g] : srec[g]);
    }
    fl_prof_mark(11);
}
