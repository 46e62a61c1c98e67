// kernels_parse6.h -- round 5: the tokenizer of the chunk path (levels 4..7) on a SPARSER chain in LDS.
//
// k_lz_parse (kernels_parse.h) walks the reference's own chain L4 (positions with the same 15-bit hash of four bytes)
// candidate after candidate: 5.1 chain steps per byte of text at level 6, 95 % of them with a match of at least 4 bytes
// in hand and 99 % of those rejected by the four-byte filter.  A candidate that can still change the result shares MORE
// bytes with the position than the match in hand (deflate.zig:248-263 keeps only longer ones), so a walk with `len`
// bytes in hand may follow any chain that holds every earlier position sharing len + 1 bytes, in the same order.
// LDS has room for the window and ONE chain (3 bytes per position, 145 KiB per sub-pass).  Here that chain is L6
// (positions with the same hash of SIX bytes, k_lz_links<2, 1>):
//
//   phase B  the automaton of k_lz_parse, segment by segment, speculative + stitched exactly as there; but every call
//            walks L6 at once, as if 5 bytes were in hand: it sees every candidate of 6 bytes and more, nearest first,
//            i.e. the result of the reference's walk whenever that result has at least 6 bytes (the reference's final
//            answer is the nearest candidate of maximal length, or the first one of `nice` bytes: shorter candidates
//            it accepts on the way do not change it).  1.4 steps per byte instead of 5.1.
//   budget   the reference looks at the first `chain` members of the L4 chain (a quarter from `good` bytes on).  That is
//            the bound B on positions of kernels_rank.h: lo = max(1, p - 32768, B[p]) -- loaded when the call starts,
//            applied at the lane's next visit of the slow block (a candidate that passed the filter is only judged
//            there; steps taken beyond the bound in between are wasted, never wrong).
//   phase A  what L6 cannot see: when a call with fewer than 5 bytes in hand finds nothing on L6, the reference's
//            answer is E5 = its first candidate of 5 bytes, else (with nothing in hand) E4 = its first candidate of 4,
//            each with exactly that length (anything longer would be on L6).  Before the automaton runs, with L4 in
//            LDS where L6 will be, a lane per position walks L4 as the reference does (budget counted, 32768 bytes)
//            until the first candidate of 5 bytes and leaves (E4, E5) in global memory; the call reads them with B.
//            A position whose nearest L6 member is a true 6-byte match within the bounds is skipped: its calls find
//            that member.  2.2 steps per byte, but lane-parallel with no automaton around them.
//
// CPU model of exactly this finder, token for token against the oracle on every corpus and level: tools/single_chain_model.c.
// Bound: as k_lz_parse (vector issue + LDS round trips of a pointer chase), with a quarter of the steps.  No MFMA.
#pragma once
#include "kernels_rank.h"

#ifndef P6_ABURST
#define P6_ABURST 8   // phase A: chain steps between two refills
#endif
#define P6_VBEST 5u   // a call walks L6 as if this many bytes were in hand
// Levels whose chain budget reaches this value take k_lz_rank / k_lz_links<2, 1> / k_lz_parse6 (flate_hip.hip), the ones below
// k_lz_chain / k_lz_parse.  MEASURED (round 5, 1 GiB text level 6, profiles/r05_parse_experiments.txt): bit-exact, but
// k_lz_rank 3.75 + k_lz_links 1.28 + k_lz_parse6 41.8 ms against k_lz_chain 1.27 + k_lz_parse 23.3 -- the automaton itself
// got faster (bursts 302 k -> 146 k cycles per wave), phase A and the waits for B / E from memory cost more than that.  So no
// level takes it by default; libflate_hip_parse6.so (-DFL_PARSE6_MIN_CHAIN=16u) keeps it under test.
#ifndef FL_PARSE6_MIN_CHAIN
#define FL_PARSE6_MIN_CHAIN 0xffffffffu
#endif

__global__ __launch_bounds__(PZ_THREADS, PZ_THREADS / 256) void k_lz_parse6(const uint8_t* __restrict__ in,
                                                            const fl_chunk* __restrict__ chunks, fl_params prm,
                                                            const uint16_t* __restrict__ l4_all,
                                                            const uint16_t* __restrict__ l6_all,
                                                            const uint32_t* __restrict__ bnd_all, uint32_t* ent_all,
                                                            const uint32_t* __restrict__ cflag,
                                                            uint32_t* __restrict__ desc_all,
                                                            uint32_t* __restrict__ true_all) {
    __shared__ uint32_t win32[PZ_WIN_DW];
    __shared__ uint16_t prv[PZ_PRV_N];
    __shared__ uint16_t tX[PZ_THREADS];       // exit of a lane's own parse, as soon as it is known
    __shared__ uint16_t tExg[PZ_THREADS];     // exit the path is assumed to take out of a segment
    __shared__ uint16_t tNxt[2][PZ_THREADS];  // segment that exit lands in (pointer jumping, double buffered)
    __shared__ uint16_t tEnt[PZ_THREADS];     // position at which the path enters a segment
    __shared__ uint16_t tMark[PZ_THREADS];    // segment is on the path
    __shared__ uint32_t sh_next_entry;
    constexpr bool small = false, STREAM = false;  // (PZ_SEG_OF of kernels_parse.h asks: other segment sizes are the whole-stream path's, k_lz_parse<true>)
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;
    const uint8_t* src = in + ck.in_off;
    const uint16_t* l4g = l4_all + (uint64_t)c * FL_CHUNK_STRIDE;
    const uint16_t* l6g = l6_all + (uint64_t)c * FL_CHUNK_STRIDE;
    const uint32_t* bndg = bnd_all + (uint64_t)c * FL_CHUNK_STRIDE;
    uint32_t* entg = ent_all + (uint64_t)c * FL_CHUNK_STRIDE;
    uint32_t* descg = desc_all + ck.pos_off;
    uint32_t* trueg = true_all + (ck.pos_off >> 5);
    const uint32_t chain = prm.chain, good = prm.good, lazy = prm.lazy, nice = prm.nice;
    const uint32_t prv_lds = (uint32_t)(size_t)(fl_lds_u32*)prv, win_lds = (uint32_t)(size_t)(fl_lds_u32*)win32;  // LDS byte addresses
    if (cflag[c] == 1u) {
        // one repeated byte (k_lz_rank saw it and built no chains): the anchors as k_lz_parse writes them
        for (uint32_t k = tid; 2 + FL_MAX_MATCH * k < N || k < 1; k += PZ_THREADS) {
            if (k == 0) {
                uint32_t w = 0;
                for (uint32_t p = 0; p < min(N, 2u); p++) {
                    descg[p] = PZ_DESC_LIT;
                    w |= 1u << p;
                }
                if (w) atomicOr(&trueg[0], w);
            }
            const uint32_t a = 2 + FL_MAX_MATCH * k;
            if (a >= N) continue;
            if (N - a >= FL_MIN_MATCH) {
                const uint32_t len = min(N - a, (uint32_t)FL_MAX_MATCH);
                descg[a] = 0x80000000u | ((len - 3u) << 15);  // j = 0, distance 1
                atomicOr(&trueg[a >> 5], 1u << (a & 31u));
            } else {
                for (uint32_t p = a; p < N; p++) {
                    descg[p] = PZ_DESC_LIT;
                    atomicOr(&trueg[p >> 5], 1u << (p & 31u));
                }
            }
        }
        return;
    }

#ifdef PZ_PROF
    uint64_t c_t0 = __builtin_readcyclecounter(), c_tstage = 0, c_tA = 0, c_tspec = 0, c_tstitch = 0, c_slow = 0, c_fast = 0, c_walk = 0, c_tfast = 0, c_tslow = 0;
#endif
    for (uint32_t sub = 0; sub < 2; sub++) {
        const uint32_t t0 = sub ? PZ_TA : 0u;
        if (t0 >= N) break;
        const uint32_t end = min(N, sub ? 65536u : PZ_TA);  // targets [t0, end)
        const uint32_t r0 = sub ? (PZ_TA - FL_MAX_DIST - PZ_MARGIN) : 0u;  // everything below is relative to r0
        const uint32_t S = sub ? PZ_SEG_B : PZ_SEG_A;
        const uint32_t nseg = PZ_SEG_OF(end - t0 + S - 1);
        const uint32_t Nr = N - r0;            // end of the input
        const uint32_t endr = end - r0, t0r = t0 - r0;
        if (sub) __syncthreads();  // the previous sub-pass is done with the LDS tables
#ifdef PZ_PROF
        uint64_t c_ts = __builtin_readcyclecounter();
#endif
        // ---- stage the window bytes (sub-pass B: two thirds of them sit in LDS already, r0 positions further up) ...
        {
            const uint32_t wkeep = sub ? PZ_WIN_DW - (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u : 0u;  // window dwords that stay
            const uint32_t sh16 = (uint32_t)((uintptr_t)(src + r0) & 15);
            const uint32_t ash = sh16 & 3u, dshift = sh16 >> 2;
            const uint4* src16 = (const uint4*)(src + r0 - sh16);
            const uint32_t ngran = (Nr + sh16 + 15) >> 4;  // granules that hold at least one byte of the input
            auto put_win = [&](uint32_t i, uint32_t lo_, uint32_t hi_) {
                uint32_t v = __builtin_amdgcn_alignbyte(hi_, lo_, ash);
                if (4 * i + 4 > Nr) v = 4 * i < Nr ? (v & ((1u << (8 * (Nr - 4 * i))) - 1u)) : 0u;  // zero padding
                if (i < PZ_WIN_DW) win32[i] = v;
            };
            auto put_gran = [&](uint32_t G, const uint4& g, uint32_t nx, uint32_t first) {
                const uint32_t d[5] = {g.x, g.y, g.z, g.w, nx};
#pragma unroll
                for (uint32_t jj = 0; jj < 4; jj++) {
                    const int32_t i = (int32_t)(4 * G + jj) - (int32_t)dshift;
                    if (i >= (int32_t)first) put_win((uint32_t)i, d[jj], d[jj + 1]);
                }
            };
            auto load_gran = [&](uint32_t G, uint4& g, uint32_t& nx) {
                g = G < ngran ? src16[G] : make_uint4(0, 0, 0, 0);
                nx = G + 1 < ngran ? ((const uint32_t*)(src16 + G + 1))[0] : 0u;
            };
            if (!sub) {
                constexpr uint32_t WG = ((PZ_WIN_DW + 6) / 4 + PZ_THREADS - 1) / PZ_THREADS;
                uint4 wg[WG];
                uint32_t wx[WG];
#pragma unroll
                for (uint32_t u = 0; u < WG; u++) load_gran(u * PZ_THREADS + tid, wg[u], wx[u]);
#pragma unroll
                for (uint32_t u = 0; u < WG; u++) put_gran(u * PZ_THREADS + tid, wg[u], wx[u], 0u);
            } else {
                constexpr uint32_t WNEW = (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u;  // window dwords that come from memory
                constexpr uint32_t WNG = ((WNEW + 6) / 4 + PZ_THREADS - 1) / PZ_THREADS;
                const uint32_t Gfirst = (wkeep + dshift) >> 2;
                uint4 wg[WNG];
                uint32_t wx[WNG];
#pragma unroll
                for (uint32_t u = 0; u < WNG; u++) load_gran(Gfirst + u * PZ_THREADS + tid, wg[u], wx[u]);
                constexpr uint32_t WK = ((PZ_WIN_DW - (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u) + PZ_THREADS - 1) / PZ_THREADS;
                uint32_t kw[WK];
#pragma unroll
                for (uint32_t u = 0; u < WK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    kw[u] = i < wkeep ? win32[i + r0 / 4u] : 0u;
                }
                __syncthreads();
#pragma unroll
                for (uint32_t u = 0; u < WK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    if (i < wkeep) win32[i] = kw[u];
                }
#pragma unroll
                for (uint32_t u = 0; u < WNG; u++) put_gran(Gfirst + u * PZ_THREADS + tid, wg[u], wx[u], wkeep);
            }
        }
        // ... and a chain: 16-bit links relative to r0, 0 = none (also everything at or below r0)
        auto stage_links = [&](const uint16_t* lg_) {
            const uint32_t nb_pos = min(Nr, (uint32_t)PZ_PRV_N);  // positions whose links are staged
            const uint4* pv4 = (const uint4*)(lg_ + r0);          // (r0 * 2 bytes is a multiple of 16)
            uint32_t* prv2 = (uint32_t*)prv;
            constexpr uint32_t PG = (PZ_PRV_N / 8 + PZ_THREADS - 1) / PZ_THREADS;
            static_assert(PZ_PRV_N % 8 == 0, "whole 16-byte loads of links");
            uint4 lg[PG];
#pragma unroll
            for (uint32_t u = 0; u < PG; u++) {
                const uint32_t i4 = u * PZ_THREADS + tid;
                lg[u] = (8 * i4 < nb_pos && i4 < PZ_PRV_N / 8) ? pv4[i4] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (uint32_t u = 0; u < PG; u++) {
                const uint32_t i4 = u * PZ_THREADS + tid;
                const uint32_t d[4] = {lg[u].x, lg[u].y, lg[u].z, lg[u].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t i = 4 * i4 + k;     // dword of two links
                    const uint32_t pa = 2 * i + r0;    // absolute position of the low half
                    uint32_t v = d[k];
                    if (pa >= Mpos) v &= 0xffff0000u;  // positions without a hash entry have no link (never written)
                    if (pa + 1 >= Mpos) v &= 0x0000ffffu;
                    uint32_t lo16 = v & 0xffffu, hi16 = v >> 16;
                    lo16 = lo16 > r0 ? lo16 - r0 : 0u;
                    hi16 = hi16 > r0 ? hi16 - r0 : 0u;
                    if (i < PZ_PRV_N / 2) prv2[i] = lo16 | (hi16 << 16);
                }
            }
        };
        stage_links(l4g);
        __syncthreads();
#ifdef PZ_PROF
        c_tstage += __builtin_readcyclecounter() - c_ts;
        c_ts = __builtin_readcyclecounter();
#endif

        // ---- phase A: (E4, E5) of the positions a call with fewer than 5 bytes in hand can start at
        {
            const uint32_t pa_end = min(endr + 8u, Nr);
            const uint32_t npos = pa_end > t0r ? pa_end - t0r : 0u;
            uint32_t i = tid;
            bool active = false;
            uint32_t p = 0, q = 0, cntA = 0, e4 = 0, loA = 0, pv0 = 0, pv1 = 0;
            bool ok5 = false;
            uint32_t nx_l6 = 0, nx_b = 0;  // of the lane's next position, requested one refill ahead
            if (i < npos) {
                nx_l6 = l6g[t0 + i];
                nx_b = bndg[t0 + i];
            }
            for (;;) {
                if (!active && i < npos) {
                    p = t0r + i;
                    const uint32_t l6 = nx_l6, bb = nx_b;
                    i += PZ_THREADS;
                    if (i < npos) {
                        nx_l6 = l6g[t0 + i];
                        nx_b = bndg[t0 + i];
                    }
                    uint32_t ent = 0;
                    bool walk = false;
                    if (p + r0 < Mpos) {
                        fl_lds_load8(win32, p, pv0, pv1);
                        const uint32_t maxlen = min(Nr - p, (uint32_t)FL_MAX_MATCH);
                        ok5 = maxlen >= 5u;
                        loA = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;
                        uint32_t b1 = bb & 0xffffu;
                        b1 = b1 > r0 ? b1 - r0 : 0u;
                        const uint32_t q6 = l6 > r0 ? l6 - r0 : 0u;
                        bool has6 = false;
                        if (maxlen >= 6u && q6 >= max(loA, b1)) {  // (q6 >= 1: a link to nothing is 0)
                            uint32_t c0, c1;
                            fl_lds_load8(win32, q6, c0, c1);
                            has6 = c0 == pv0 && ((c1 ^ pv1) & 0xffffu) == 0u;
                        }
                        if (!has6) {
                            q = prv[p];
                            cntA = chain;
                            e4 = 0;
                            walk = q >= loA;
                        }
                    }
                    if (walk)
                        active = true;
                    else
                        entg[p + r0] = ent;
                }
                if (__ballot(active) == 0ull && __ballot(i < npos) == 0ull) break;
#pragma unroll 1
                for (int s = 0; s < P6_ABURST; s++) {
                    if (__ballot(active) == 0ull) break;
                    if (active) {
                        // one candidate of the reference's walk (deflate.zig:248-263): 4 bytes make it E4, 5 end the walk
                        uint32_t c0, c1;
                        fl_lds_load8(win32, q, c0, c1);
                        const uint32_t nq = prv[q];
                        uint32_t e5 = 0;
                        if (c0 == pv0) {
                            if (!e4) e4 = q;
                            if (ok5 && ((c1 ^ pv1) & 0xffu) == 0u) e5 = q;
                        }
                        cntA--;
                        if (e5 || nq < loA || cntA == 0u) {
                            entg[p + r0] = (e4 ? e4 + r0 : 0u) | ((e5 ? e5 + r0 : 0u) << 16);
                            active = false;
                        } else {
                            q = nq;
                        }
                    }
                }
            }
        }
        __syncthreads();  // L4 has been walked (and the entries are written: same workgroup, same cache)
#ifdef PZ_PROF
        c_tA += __builtin_readcyclecounter() - c_ts;
        c_ts = __builtin_readcyclecounter();
#endif
        stage_links(l6g);
        __syncthreads();
#ifdef PZ_PROF
        c_tstage += __builtin_readcyclecounter() - c_ts;
#endif

        const uint32_t y0 = sub ? sh_next_entry : 0u;  // the sub-pass is entered at this anchor (relative)
#ifdef PZ_ILV
        const uint32_t m = ((tid & 63u) * PZ_WAVES) + (tid >> 6);
#else
        const uint32_t m = tid;                         // this lane's segment
#endif
        const uint32_t seg0 = t0r + m * S;
        const uint32_t seg_end = min(seg0 + S, endr);
        if (y0 >= endr) {  // the path jumps over the whole sub-pass: no anchors (the bitmap is zero already)
            if (tid == 0) sh_next_entry = y0;
            continue;
        }
        const uint32_t me = PZ_SEG_OF(y0 - t0r);
        // per-segment state of the stitch
        uint64_t A = 0, F = 0;        // anchors of the lane's own parse; of the parse from the entry
        uint32_t X = seg_end;         // exit of the lane's own parse
        uint32_t res_entry = PZ_NONE, res_exit = 0, Z = PZ_NONE;
        bool marked = false;
        if (m < PZ_THREADS) tX[m] = (uint16_t)PZ_NONE;
        __syncthreads();

        enum { ST_SPEC = 0, ST_WAIT = 1, ST_FIX = 2, ST_DONE = 3 };
        for (uint32_t round = 0;; round++) {
#ifdef PZ_PROF
            const uint64_t c_tr0 = __builtin_readcyclecounter();
#endif
            uint32_t st = ST_DONE;
            uint32_t a = 0;
            uint64_t stopmask = 0;
            uint32_t y_in = PZ_NONE;
            bool deferred = false;   // this lane's entry is not the one it is resolved for, but may still move: next round
            bool fixing = false;     // this lane parses its segment again in this round ...
            uint32_t ex_used = 0;    // ... and this is the exit the round's path assumed for it
            if (round == 0) {
                if (m < nseg && seg_end > y0) {
                    st = ST_SPEC;
                    a = (m == me) ? y0 : seg0;
                }
            } else {
                // the path, assuming every segment not resolved yet leaves through its own exit
                if (m < nseg) {
                    const uint32_t ex = res_entry != PZ_NONE ? res_exit : X;
                    ex_used = ex;
                    tExg[m] = (uint16_t)ex;
                    tNxt[0][m] = (uint16_t)(ex >= endr ? nseg : PZ_SEG_OF(ex - t0r));
                    tMark[m] = m == me ? 1 : 0;
                    tEnt[m] = m == me ? (uint16_t)y0 : (uint16_t)PZ_NONE;
                }
                __syncthreads();
                uint32_t cur = 0;
                for (uint32_t step = 0; (1u << step) < nseg; step++) {
                    if (m < nseg) {
                        const uint32_t n = tNxt[cur][m];
                        if (n < nseg) {
                            if (tMark[m]) tMark[n] = 1;
                            tNxt[cur ^ 1][m] = tNxt[cur][n];
                        } else {
                            tNxt[cur ^ 1][m] = (uint16_t)nseg;
                        }
                    }
                    __syncthreads();
                    cur ^= 1;
                }
                marked = m < nseg && tMark[m] != 0;
                if (marked) {
                    const uint32_t ex = tExg[m];
                    const uint32_t n0 = ex >= endr ? nseg : PZ_SEG_OF(ex - t0r);
                    if (n0 < nseg) {
                        tEnt[n0] = (uint16_t)ex;
                        tNxt[0][n0] = (uint16_t)m;  // (the segment the path comes from; the jump tables are free now)
                    } else {
                        sh_next_entry = ex;  // (relative to this sub-pass's r0; converted below)
                    }
                }
                __syncthreads();
                if (marked) y_in = tEnt[m];
                const bool need = marked && y_in != res_entry;
                // A segment is parsed again only when the segment the path comes from is settled for the entry IT got
                tMark[m] = need ? 0 : 1;
                __syncthreads();
                const bool work = need && (m == me || tMark[tNxt[0][m]] != 0);
                deferred = need && !work;
                if (!__syncthreads_or(need ? 1 : 0)) break;
                if (work) {
                    st = ST_FIX;
                    a = y_in;
                    stopmask = A;
                    fixing = true;
                }
            }
            // ---- the automaton (deflate.zig:154-205)
            uint64_t amask = 0;
            uint32_t j = 0, plen = 0, pdist = 0;
            uint32_t p = 0, q = 0, cnt = 0, lo = 1, best = 0, bdist = 0, maxlen = 0, pref = 0;
            uint32_t cfl = 0;              // bytes in hand when the call started
            uint32_t recB = 0, recE = 0;   // B and (E4, E5) of the call's position, on their way from memory ...
            bool rpend = false;            // ... since the visit of the slow block that started the call
            uint32_t qh = PZ_NOHIT;
            // A walking lane (cnt != 0) holds what its candidate q needs to be judged: the link nqA = prv[q] and the two
            // aligned window dwords w0A, w1A that hold q's bytes number off .. off + 3 (kernels_parse.h)
            uint32_t nqA = 0, w0A = 0, w1A = 0, xq = 0, offb = win_lds;
#define P6_SET_FILTER(OFF)                                                     \
    do {                                                                       \
        offb = (OFF) + win_lds;                                                \
        pref = pz_lds4(win32, p + (OFF));                                      \
    } while (0)
#define P6_START_CALL(PP, LL)                                                  \
    do {                                                                       \
        p = (PP);                                                              \
        cfl = (LL);                                                            \
        best = max((uint32_t)(LL), P6_VBEST);                                  \
        bdist = 0;                                                             \
        maxlen = min(Nr - p, (uint32_t)FL_MAX_MATCH);                          \
        q = prv[p];                                                            \
        lo = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;                           \
        cnt = (maxlen > best && q >= lo) ? 1u : 0u;                            \
        P6_SET_FILTER(best - 3u);                                              \
        recB = bndg[p + r0];                                                   \
        recE = entg[p + r0];                                                   \
        rpend = true;                                                          \
    } while (0)
#define P6_LOAD_CAND()                                                         \
    do {                                                                       \
        xq = q + offb;                                                         \
        nqA = prv[q];                                                          \
        w0A = win32[(xq - win_lds) >> 2];                                      \
        w1A = win32[((xq - win_lds) >> 2) + 1];                                \
    } while (0)
            // a parse that starts on a position where it has to stop already (FIX only)
            if (st == ST_FIX && ((stopmask >> ((a - seg0) & 63u)) & 1ull)) {
                F = 0;
                res_entry = y_in;
                Z = a;
                res_exit = X;
                st = ST_DONE;
            }
            if (st != ST_DONE) {
                P6_START_CALL(a, 0u);
                if (cnt != 0) P6_LOAD_CAND();
            }
            for (;;) {
                const uint64_t alive = __ballot(st != ST_DONE);
                if (alive == 0) break;
                const uint64_t waiting = __ballot(st == ST_WAIT);
                if (waiting == alive) __builtin_amdgcn_s_sleep(8);  // nothing to do but wait for another wave
                const uint64_t serve = alive & ~waiting;              // (a waiting lane is polled when the others are served)
#ifdef PZ_PROF
                const uint64_t c_ta = __builtin_readcyclecounter();
#endif
                // ---- fast steps: one chain candidate per step, rejected on the four bytes that end at offset `best`
#pragma unroll 1
                for (int b = 0; b < PZ_BURST; b += PZ_UNROLL) {
                    const uint64_t mw = __ballot(cnt != 0);
                    if (mw == 0 || __popcll(serve & ~mw) >= PZ_NEED) break;
#ifdef PZ_PROF
                    c_fast++;
                    c_walk += __popcll(mw);
#endif
                    if (cnt != 0) {
                        // the software-pipelined burst of k_lz_parse without the candidate count: the bound `lo` ends a walk
                        uint32_t nqB, w0B, w1B, a1, a2, xn, t, nqh = 0;
                        uint64_t s_save, s_t, s_hit;
                        asm volatile(
                            "s_mov_b64 %[ssave], exec\n\t"
                            "s_mov_b64 %[shit], 0\n\t"
                            ".rept " PZ_STR(PZ_HALF_UNROLL) "\n\t"
                            "v_lshl_add_u32 %[a1], %[nqA], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqB], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqA], %[offb]\n\t"
                            "v_and_b32 %[a2], -4, %[xn]\n\t"
                            "ds_read_b32 %[w0B], %[a2]\n\t"
                            "ds_read_b32 %[w1B], %[a2] offset:4\n\t"
                            "v_alignbyte_b32 %[t], %[w1A], %[w0A], %[xq]\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqA], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqA], %[lo]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqA]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lp6_done_%=\n\t"
                            "v_lshl_add_u32 %[a1], %[nqB], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqA], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqB], %[offb]\n\t"
                            "v_and_b32 %[a2], -4, %[xn]\n\t"
                            "ds_read_b32 %[w0A], %[a2]\n\t"
                            "ds_read_b32 %[w1A], %[a2] offset:4\n\t"
                            "v_alignbyte_b32 %[t], %[w1B], %[w0B], %[xq]\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqB], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqB], %[lo]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqB]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lp6_done_%=\n\t"
                            ".endr\n\t"
                            ".Lp6_done_%=:\n\t"
                            "s_mov_b64 %[st], exec\n\t"
                            "s_mov_b64 exec, %[ssave]\n\t"
                            : [q] "+v"(q), [nqA] "+v"(nqA), [w0A] "+v"(w0A), [w1A] "+v"(w1A), [xq] "+v"(xq),
                              [qh] "+v"(qh), [nqh] "+v"(nqh), [nqB] "=&v"(nqB), [w0B] "=&v"(w0B), [w1B] "=&v"(w1B),
                              [a1] "=&v"(a1), [a2] "=&v"(a2), [xn] "=&v"(xn), [t] "=&v"(t), [ssave] "=&s"(s_save),
                              [st] "=&s"(s_t), [shit] "=&s"(s_hit)
                            : [lo] "v"(lo), [offb] "v"(offb), [pref] "v"(pref), [prvb] "s"(prv_lds)
                            : "vcc", "scc", "memory");
                        // s_t: the lanes that are still walking (their candidate's data is in the A registers again: an even
                        // number of steps); s_hit: those that stopped on a candidate that passes the filter
                        const uint32_t ln = threadIdx.x & 63u;
                        if (!((s_t >> ln) & 1ull)) {
                            if ((s_hit >> ln) & 1ull) q = nqh;  // (the walk goes on behind the candidate, at its link)
                            cnt = 0;
                        }
                    }
                }
                // ---- slow block
#ifdef PZ_PROF
                const uint64_t c_tb = __builtin_readcyclecounter();
                c_tfast += c_tb - c_ta;
                c_slow++;
#endif
                // (0) what the call asked memory for when it started has arrived: the budget as a bound on positions
                if (rpend && st != ST_DONE) {
                    uint32_t bb = cfl >= good ? recB >> 16 : recB & 0xffffu;  // deflate.zig:241-245
                    bb = bb > r0 ? bb - r0 : 0u;
                    lo = max(lo, bb);
                    rpend = false;
                    if (cnt != 0 && q < lo) cnt = 0;  // the candidate in hand lies beyond it: the walk is over
                }
                if (st != ST_DONE && cnt == 0) {
                    if (qh != PZ_NOHIT) {
                        if (qh >= lo) {
                            // the candidate agrees where it must: its exact common prefix with p
                            uint32_t l = 0;
                            for (;;) {
                                uint32_t a0, a1, b0, b1;
                                fl_lds_load8(win32, p + l, a0, a1);
                                fl_lds_load8(win32, qh + l, b0, b1);
                                const uint64_t x = (uint64_t)(a0 ^ b0) | ((uint64_t)(a1 ^ b1) << 32);
                                if (x) {
                                    l += (uint32_t)__builtin_ctzll(x) >> 3;
                                    break;
                                }
                                l += 8;
                                if (l >= maxlen) break;
                            }
                            l = min(l, maxlen);
                            cnt = q >= lo ? 1u : 0u;   // the walk goes on behind the candidate ...
                            if (l > best) {            // deflate.zig:254-261 (best >= 5: a match)
                                best = l;
                                bdist = p - qh;
                                if (l >= nice || l >= maxlen) {
                                    cnt = 0;  // ... unless the match is good enough / nothing longer is possible
                                } else {
                                    P6_SET_FILTER(l - 3u);
                                }
                            }
                        }  // (else: beyond the reference's budget, and so is everything behind it)
                        qh = PZ_NOHIT;
                    }
                    if (cnt == 0) {
                        // one move per lane and visit; every path through it ends in at most one new call
                        bool start = false;
                        uint32_t sp = 0, sl = 0;
                        if (st == ST_WAIT) {
                            // the lane before has finished its own parse: where does that leave this segment?
                            const uint32_t v = __hip_atomic_load(&tX[m - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (v != PZ_NONE) {
                                st = ST_DONE;
                                if (v >= seg0 && v < seg_end) {
                                    y_in = v;
                                    if ((A >> (v - seg0)) & 1ull) {  // on an anchor of the own parse
                                        F = 0;
                                        res_entry = v;
                                        Z = v;
                                        res_exit = X;
                                    } else {
                                        st = ST_FIX;
                                        a = v;
                                        stopmask = A;
                                        amask = 0;
                                        start = true;
                                        sp = v;
                                    }
                                }
                            }
                        } else {
                            // the call has ended.  Nothing of 6 bytes or more: what the reference's walk ends with then is
                            // its first candidate of 5 bytes, or -- with nothing in hand -- of 4 (phase A)
                            if (!bdist && cfl < P6_VBEST && p + r0 < Mpos) {
                                const uint32_t e4 = recE & 0xffffu, e5 = recE >> 16;
                                if (e5) {
                                    best = 5u;
                                    bdist = p + r0 - e5;
                                } else if (cfl == 0u && e4) {
                                    best = 4u;
                                    bdist = p + r0 - e4;
                                }
                            }
                            // the automaton's next move
                            bool emit = true;  // the pending match goes out (deflate.zig:182-184), or a literal
                            if (bdist) {       // a match, longer than the pending one if there is one
                                if (p != a) j++;  // the pending match's position becomes a literal (deflate.zig:166-168)
                                plen = best;
                                pdist = bdist;
                                emit = plen >= lazy;  // deflate.zig:171-173
                            }
                            if (emit) {
                                uint32_t desc = PZ_DESC_LIT, next = a + 1;
                                if (plen) {
                                    desc = 0x80000000u | (j << 23) | ((plen - 3u) << 15) | (pdist - 1u);
                                    next = a + j + plen;
                                }
                                descg[a + r0] = desc;
                                amask |= 1ull << (a - seg0);
                                a = next;
                                j = 0;
                                plen = 0;
                                const bool meet = a < seg_end && ((stopmask >> ((a - seg0) & 63u)) & 1ull);
                                if (a >= seg_end || meet) {
                                    // the parse leaves the segment or steps on an anchor of the lane's own parse
                                    if (st == ST_SPEC) {
                                        A = amask;
                                        X = a;
                                        __hip_atomic_store(&tX[m], (uint16_t)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        amask = 0;
                                        if (m == me) {  // the entry segment's own parse is the true one
                                            res_entry = y0;
                                            res_exit = a;
                                            Z = y0;
                                            st = ST_DONE;
                                        } else {
                                            st = ST_WAIT;
                                        }
                                    } else {
                                        F = amask;
                                        res_entry = y_in;
                                        Z = meet ? a : PZ_NONE;
                                        res_exit = meet ? X : a;
                                        st = ST_DONE;
                                    }
                                } else {
                                    start = true;
                                    sp = a;
                                }
                            } else {
                                // keep the match, look one position further (deflate.zig:174-178)
                                start = true;
                                sp = a + j + 1u;
                                sl = plen;
                            }
                        }
                        if (start) P6_START_CALL(sp, sl);
                    }
                    if (cnt != 0) P6_LOAD_CAND();  // (every lane that comes out of this block walking has a new q)
                }
#ifdef PZ_PROF
                c_tslow += __builtin_readcyclecounter() - c_tb;
#endif
            }
#undef P6_START_CALL
#undef P6_SET_FILTER
#undef P6_LOAD_CAND
#ifdef PZ_PROF
            if (round == 0) c_tspec += __builtin_readcyclecounter() - c_tr0; else c_tstitch += __builtin_readcyclecounter() - c_tr0;
#endif
            // Every segment parsed again in this round leaves where the round's path assumed: the path stands
            if (round >= 1 && !__syncthreads_or(((fixing && res_exit != ex_used) || deferred) ? 1 : 0)) break;
        }
        // ---- the true anchors of this sub-pass
        if (m < nseg) {
            uint64_t T = 0;
            if (marked) {
                T = F;
                if (Z != PZ_NONE) T |= A & (~0ull << (Z - seg0));
            }
            const uint32_t pa = seg0 + r0, shb = pa & 31u;
            const uint64_t lo64 = T << shb;
            const uint32_t w0 = (uint32_t)lo64, w1 = (uint32_t)(lo64 >> 32), w2 = shb ? (uint32_t)(T >> (64u - shb)) : 0u;
            if (w0) atomicOr(&trueg[pa >> 5], w0);
            if (w1) atomicOr(&trueg[(pa >> 5) + 1], w1);
            if (w2) atomicOr(&trueg[(pa >> 5) + 2], w2);
        }
        __syncthreads();
        // the next sub-pass counts from its own r0
        if (tid == 0 && sub == 0) sh_next_entry = sh_next_entry - (PZ_TA - FL_MAX_DIST - PZ_MARGIN);
    }
#ifdef PZ_PROF
    if ((tid & 63) == 0) {  // (tools/parse_probe.py reads the same slots for k_lz_parse)
        atomicAdd((unsigned long long*)&g_fl_prof[40], (unsigned long long)c_fast);
        atomicAdd((unsigned long long*)&g_fl_prof[41], (unsigned long long)c_walk);
        atomicAdd((unsigned long long*)&g_fl_prof[42], (unsigned long long)c_slow);
        atomicAdd((unsigned long long*)&g_fl_prof[46], (unsigned long long)c_tspec);
        atomicAdd((unsigned long long*)&g_fl_prof[47], (unsigned long long)c_tstitch);
        atomicAdd((unsigned long long*)&g_fl_prof[48], (unsigned long long)(__builtin_readcyclecounter() - c_t0));
        atomicAdd((unsigned long long*)&g_fl_prof[49], 1ull);
        atomicAdd((unsigned long long*)&g_fl_prof[50], (unsigned long long)c_tfast);
        atomicAdd((unsigned long long*)&g_fl_prof[52], (unsigned long long)c_tslow);
        atomicAdd((unsigned long long*)&g_fl_prof[55], (unsigned long long)c_tstage);
        atomicAdd((unsigned long long*)&g_fl_prof[57], (unsigned long long)c_tA);
    }
#endif
}
