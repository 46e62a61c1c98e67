// kernels_walk.h -- the LZ77 match finder + lazy-matching automaton of the chunk path (levels 4..9, inputs of at most
// 65535 bytes), round 4: SPARSE CHAINS.  Round 6: the same kernel over the WINDOWS of long streams at levels 8 and 9
// (k_lz_walk<true, true>, `wk_stream`: see there and k_lz_parse<true> in kernels_parse.h).
//
// The reference's findMatch (deflate.zig:233-266) walks the chain of the positions that share a 15-bit hash of four
// bytes with p, nearest first, at most `chain` of them, and keeps a candidate only if it is LONGER than the match in
// hand.  On text 95 % of its steps happen with a match of 4 bytes or more in hand and 99 % of those candidates fail the
// reference's one-compare reject (SlidingWindow.zig:91-98): measured 5.1 steps per input byte at level 6, 14 at level 9.
// A candidate that can still change the result agrees with p on MORE bytes than the match in hand.  So with a match of
// `len` bytes in hand the walk may follow any chain that holds every earlier position agreeing with p on len + 1
// bytes, in the same (descending) order -- the result is the same, candidate for candidate:
//
//   len < 5        the reference's chain                     L4   (k_lz_links<0>: Lookup.zig:23-51)
//   len 5 or 6     positions with the same hash of 6 bytes   L6   (k_lz_links<2>)
//   len >= 7       positions with the same hash of 8 bytes   L8   (k_lz_links<3>)
//
// What the reference counts down per candidate (`chain`, a quarter of it from `good` on: deflate.zig:241-245) is
// counted as before while the walk is on L4; on L6 / L8 it is checked per ACCEPTED candidate with
// RK[p] = number of earlier positions in p's L4 bucket (k_lz_links<1>): candidate q of p is the (RK[p] - RK[q])-th of
// the reference's walk, and the first one beyond the budget ends the call (everything behind it is farther still).
// Steps per input byte on the benchmark text: 5.1 -> 1.2 at level 6, 14.4 -> 1.4 at level 9; CPU model with the same
// walk, checked token for token against the oracle: tools/multilevel_model.c.
//
//   k_lz_links<W>  one hash table in LDS per launch (as k_lz_chain: DS_MSKOR_RTN exchange of the bucket's head, or
//                  DS_ADD_RTN for the count), 16-bit links to global memory.
//   k_lz_walk      one workgroup per chunk, the whole chunk's bytes in LDS (66 KiB: two workgroups per CU), one LANE per
//                  64-byte segment running the reference's automaton speculatively, stitched as in round 3
//                  (kernels_parse.h).  The links stay in global memory (L2): a step is a 2-byte gather -- measured 0.7-1.5
//                  cycles per lane and CU (tools/ubench/gather.hip), a sixth of the steps of round 3.
//
// No MFMA: pointer hops and byte compares.
#pragma once
#include "kernels_parse.h"

// hashes of the upper levels (any function would do: the chains only have to CONTAIN what matches)
__device__ __forceinline__ uint32_t fl_hash6(uint32_t w0, uint32_t w1) {
    return ((w0 * 0x9E3779B1u) ^ ((w1 & 0xffffu) * 0x85EBCA6Bu)) >> 17;
}
__device__ __forceinline__ uint32_t fl_hash8(uint32_t w0, uint32_t w1) {
    return ((w0 * 0x9E3779B1u) ^ (w1 * 0x85EBCA6Bu)) >> 17;
}

__device__ __forceinline__ uint32_t fl_lds_add_rtn(uint32_t* lds_word, uint32_t value) {
    uint32_t old;
    fl_lds_u32* a = (fl_lds_u32*)lds_word;
    asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(old) : "v"(a), "v"(value) : "memory");
    return old;
}

// ------------------------------------------------------------------ k_lz_links
// WHICH 0: L4 (the reference's chain; also recognises a chunk of one repeated byte: cflag), 1: RK, 2: L6, 3: L8.
// Same structure as k_lz_chain (kernels_parse.h): FL_CHAIN_WAVES waves prepare a block of 1024 positions each and
// take turns with the table, so that it sees the positions in ascending order; a lane that was overtaken (a head
// above its own position, a count that is not the number of earlier bucket members) sends the chunk to the
// one-position-at-a-time path.
// NARR 4: a chunk's four arrays are one block [L4 | L6 | L8 | RK] (levels 8-9); NARR 1: an array of its own (k_lz_parse6's L6).
template <int WHICH, int NARR = 4>
__global__ __launch_bounds__(64 * FL_CHAIN_WAVES, FL_CHAIN_WAVES / 2) void k_lz_links(const uint8_t* __restrict__ in,
                                                                   const fl_chunk* __restrict__ chunks,
                                                                   uint16_t* __restrict__ out_all,
                                                                   uint32_t* __restrict__ cflag) {
    __shared__ uint32_t head32[16384 + 64];  // (+ one word per lane for the exchanges of positions past the end)
    __shared__ uint32_t stg_all[FL_CHAIN_WAVES][FL_CHAIN_STG_DW];
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* sb = stg_all[wave];
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;  // positions with 4 bytes left (Lookup.zig:24)
    if (WHICH == 0 && threadIdx.x == 0) cflag[c] = 0u;
    if (Mpos == 0) return;
    const uint8_t* src = in + ck.in_off;
    // a chunk's four arrays are one block of 4 x 65536 entries: [L4 | L6 | L8 | RK]
    uint16_t* pv = NARR == 1 ? out_all + (uint64_t)c * FL_CHUNK_STRIDE
                             : out_all + (uint64_t)c * (4u * FL_CHUNK_STRIDE) + (WHICH == 0 ? 0u : WHICH == 1 ? 3u : WHICH == 2 ? 1u : 2u) * FL_CHUNK_STRIDE;
    const uint32_t sh = (uint32_t)((uintptr_t)src & 15);
    const uint4* src16 = (const uint4*)(src - sh);  // 16-byte granules; granule g covers chunk bytes 16 g - sh ..
    const uint32_t n_gran = (N + sh + 15) >> 4;     // granules holding at least one byte of the chunk
    if (WHICH == 0) {
        // A chunk of ONE repeated byte needs no chains: k_lz_walk writes its anchors directly.
        if (N >= 64 && !ck.pad_) {  // (pad_ != 0: a WINDOW of a long stream -- k_lz_walk<true, true> enters it anywhere: chains always)
            const uint32_t b0 = src[0] * 0x01010101u;
            bool same = true;
            for (uint32_t g0 = 0; g0 < n_gran; g0 += 64 * FL_CHAIN_WAVES) {
                const uint32_t g = g0 + threadIdx.x;
                if (g < n_gran) {
                    uint4 v = src16[g];
                    const int32_t first = (int32_t)(16 * g) - (int32_t)sh;  // chunk offset of the granule's byte 0
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int bb = 0; bb < 4; bb++) {
                            const int32_t o = first + 4 * k + bb;
                            if (o >= 0 && o < (int32_t)N) m |= 0xffu << (8 * bb);
                        }
                        same = same && ((w[k] ^ b0) & m) == 0;
                    }
                }
                if (__syncthreads_or(same ? 0 : 1)) {
                    same = false;
                    break;
                }
            }
            if (same) {  // (the same verdict in every thread)
                if (threadIdx.x == 0) cflag[c] = 1u;
                return;
            }
        }
    } else {
        if (cflag[c] == 1u) return;  // (written by the <0> launch before this one on the stream)
    }
    {
        uint4* h4 = (uint4*)head32;
        for (uint32_t i = threadIdx.x; i < 4096; i += 64 * FL_CHAIN_WAVES) h4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // block b = chunk bytes [1024 b, 1024 b + 1024) plus what its last position needs (7 bytes more for the 8-byte
    // hash): granules 64 b .. 64 b + 65 (sh + 1023 + 7 < 1056 = 66 granules); lane l loads granule 64 b + l, lanes 0..1 two more
    uint64_t runny = 0;  // (WHICH 0; wave-uniform) some granule of the chunk is 16 times one byte: a run of 323 holds 19 of them
    auto load_block = [&](uint32_t b, uint4& g0, uint4& g1) {
        const uint32_t ga = 64 * b + lane, gb = 64 * b + 64 + lane;
        g0 = ga < n_gran ? src16[ga] : make_uint4(0, 0, 0, 0);
        g1 = (lane < 2 && gb < n_gran) ? src16[gb] : make_uint4(0, 0, 0, 0);
    };
    const uint32_t n_blocks = (Mpos + 1023) >> 10;
    uint4 ga0, ga1;  // the wave's next block, in flight
    load_block(wave, ga0, ga1);
    bool overtaken = false;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += FL_CHAIN_WAVES) {  // (uniform trip count: every wave meets every barrier)
        const uint32_t b = b0 + wave;
        ((uint4*)sb)[lane] = ga0;
        if (lane < 2) ((uint4*)sb)[64 + lane] = ga1;
        if (WHICH == 0) {  // (a granule that is not whole in the chunk may count: the flag only picks the kernel)
            const uint32_t bp = (ga0.x & 0xffu) * 0x01010101u;
            runny |= __ballot(ga0.x == bp && ga0.y == bp && ga0.z == bp && ga0.w == bp);
        }
        if (b + FL_CHAIN_WAVES < n_blocks) load_block(b + FL_CHAIN_WAVES, ga0, ga1);
        fl_lds_order();
        uint32_t hw[16];   // word of the table, or the lane's dummy word for a position past the end
        uint32_t odd = 0;  // bit s: the hash of step s is odd (its entry is the upper half of the word)
        uint32_t val = 0;  // bit s: position of step s exists
#pragma unroll
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t p = (b << 10) + (s << 6) + lane;
            const uint32_t off = (s << 6) + lane + sh;
            const uint32_t d0 = sb[off >> 2], d1 = sb[(off >> 2) + 1];
            const uint32_t v = __builtin_amdgcn_alignbyte(d1, d0, off & 3);
            uint32_t h;
            if (WHICH <= 1) {
                h = fl_hash_le(v);
            } else {
                const uint32_t d2 = sb[(off >> 2) + 2];
                const uint32_t v1 = __builtin_amdgcn_alignbyte(d2, d1, off & 3);
                h = WHICH == 2 ? fl_hash6(v, v1) : fl_hash8(v, v1);
            }
            const bool valid = p < Mpos;
            odd |= (h & 1u) << s;
            val |= (valid ? 1u : 0u) << s;
            hw[s] = valid ? (h >> 1) : 16384u + lane;
        }
        uint32_t old[16];
#pragma unroll 1
        for (uint32_t t = 0; t < FL_CHAIN_WAVES; t++) {
            if (t == wave) {
                // all exchanges are issued before the first result is looked at (no branch around the instruction)
#pragma unroll
                for (uint32_t s = 0; s < 16; s++) {
                    const uint32_t p = (b << 10) + (s << 6) + lane;
                    const uint32_t hs = ((odd >> s) & 1u) << 4;
                    const bool valid = (val >> s) & 1u;
                    if (WHICH == 1)
                        old[s] = fl_lds_add_rtn(&head32[hw[s]], valid ? (1u << hs) : 0u);
                    else
                        old[s] = fl_lds_mskor_rtn(&head32[hw[s]], valid ? (0xffffu << hs) : 0u, valid ? (p << hs) : 0u);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]),
                               "+v"(old[6]), "+v"(old[7]), "+v"(old[8]), "+v"(old[9]), "+v"(old[10]), "+v"(old[11]),
                               "+v"(old[12]), "+v"(old[13]), "+v"(old[14]), "+v"(old[15])
                             :
                             : "memory");
            }
            __syncthreads();
        }
#pragma unroll
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t p = (b << 10) + (s << 6) + lane;
            if ((val >> s) & 1u) {
                const uint32_t o = ((odd >> s) & 1u) ? (old[s] >> 16) : (old[s] & 0xffffu);
                overtaken = overtaken || o > p;  // (a head above the position; a count above the number of positions before it)
                pv[p] = (uint16_t)o;             // 0 = none: position 0 is the chain's null (deflate.zig:248)
            }
        }
    }
#ifdef WK_RANK_CHECK  // (built and measured in round 5: k_lz_links 4.8 -> 5.5 ms per GiB even on one chunk in eight; off)
    if (WHICH == 1 && NARR == 4 && (c & 7u) == 0u) {
        // A count says nothing about the order it was served in (o > p above never fires for it).  The links of the launch
        // before this one do: a position's rank is its predecessor's + 1, and a position without a predecessor has rank 0 --
        // or 1 when position 0, the chain's null, shares its bucket.  Two lanes of one exchange served out of lane order
        // swap their ranks and are seen here.  Every position costs two dependent gathers (all positions of all
        // chunks: 4.8 -> 8.3 ms per GiB for the four arrays; one in eight: 5.5): one chunk in eight is looked at, one position in
        // eight of it -- an LDS that serves out of order does not do so once.
        __syncthreads();  // (this workgroup's ranks are in memory)
        const uint16_t* l4 = out_all + (uint64_t)c * (4u * FL_CHUNK_STRIDE);
        constexpr uint32_t NCK = 65536u / (8u * 64u * FL_CHAIN_WAVES);  // positions per thread: their loads together, then the predecessors'
        uint32_t qq[NCK], rr[NCK], rq[NCK];
#pragma unroll
        for (uint32_t u = 0; u < NCK; u++) {
            const uint32_t p = 8u * (u * 64u * FL_CHAIN_WAVES + threadIdx.x) + ((c >> 3) & 7u);
            qq[u] = p < Mpos ? l4[p] : 0u;
            rr[u] = p < Mpos ? pv[p] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < NCK; u++) rq[u] = qq[u] ? pv[qq[u]] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < NCK; u++) overtaken = overtaken || (qq[u] ? rr[u] != rq[u] + 1u : rr[u] > 1u);
#ifdef WK_RANKCHECK_DEBUG
        if (overtaken) atomicAdd((unsigned long long*)&g_fl_prof[63], 1ull);
        overtaken = false;
#endif
    }
#endif
    // (k_lz_walk<true> takes the chunks with long runs of one byte, k_lz_walk<false> the others: cflag 2 / 0)
    if (WHICH == 0 && __syncthreads_or(runny != 0 ? 1 : 0) && threadIdx.x == 0) cflag[c] = 2u;
#ifdef FL_CHAIN_FORCE_SLOW
    overtaken = true;  // (test builds: exercise the fallback)
#endif
    if (__syncthreads_or(overtaken ? 1 : 0)) {
        // never seen on gfx950: one position at a time, by one lane (Lookup.zig:35-40 as written)
        uint16_t* head16 = (uint16_t*)head32;
        for (uint32_t i = threadIdx.x; i < 16384; i += 64 * FL_CHAIN_WAVES) head32[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t p = 0; p < Mpos; p++) {
                const uint32_t v = fl_load_u32_clamped(src, p, N);
                uint32_t h;
                if (WHICH <= 1) {
                    h = fl_hash_le(v);
                } else {
                    const uint32_t v1 = fl_load_u32_clamped(src, p + 4, N);
                    h = WHICH == 2 ? fl_hash6(v, v1) : fl_hash8(v, v1);
                }
                pv[p] = head16[h];
                head16[h] = WHICH == 1 ? (uint16_t)(head16[h] + 1u) : (uint16_t)p;
            }
        }
    }
}

// ------------------------------------------------------------------ k_lz_walk
#define WK_THREADS 1024
#define WK_SEG 64u
#define WK_WIN_DW ((65536u + 320u) / 4u)  // the chunk's bytes, zero padded (a compare reads up to 258 + 8 + 3 bytes past a position)
#define WK_L4 0u
#define WK_L6 1u
#define WK_L8 2u
#ifndef WK_WAVES_PER_SIMD
#define WK_WAVES_PER_SIMD 8  // two workgroups per CU (64 VGPRs); 4: one (128 VGPRs, nothing spilled)
#endif
#ifndef WK_STREAM_WAVES_PER_SIMD
#define WK_STREAM_WAVES_PER_SIMD 8  // k_lz_walk<true, true>: two workgroups a CU, as the chunks' (one 256 MiB stream of text, level 9: 20.5 ms against 29.6 with one)
#endif
#ifndef WK_BURST
#define WK_BURST 4        // chain steps per trip, at most
#endif
#ifndef WK_PROBE
#define WK_PROBE 8u       // steps on L6 / L8 between two looks at the budget
#endif
#ifndef WK_RUNSKIP
#define WK_RUNSKIP 512u   // bytes of a run skipped per trip, at most
#endif
#ifndef WK_MINWALK
#define WK_MINWALK 1u     // ... fewer when fewer lanes than this still walk (1: never -- 8: text level 9 19.9 GB/s, 1: 22.4)
#endif

#ifdef WK_PROF
#define WK_CNT(var, v) (var) += (v)
#else
#define WK_CNT(var, v)
#endif

// STREAM (round 6): the whole-stream path of levels 8 and 9 on this tokenizer, as k_lz_parse<true> is for levels 4-7 (see there:
// the reference's window after slide j is a chunk, `chunks` holds one fl_chunk per WINDOW, a workgroup walks a group of
// consecutive windows of one stream, the anchor the path leaves a window at is where it enters the next; a second launch
// (fix = 1) parses the groups that started from a guess again from their true entry until a window is left where it was left
// before).  A window's fl_chunk says in pad_ >> 8 how many bytes of the stream lie behind its 65536 (at most 264): the lazy calls
// of the window's last anchor are made AFTER the next slide, with the whole lookahead (deflate.zig:304-321).  Always the DEEP
// instantiation: a stream's windows are not sorted by kind.
struct wk_stream {
    const fl_swin* swins;
    const fl_chunk* schunks;
    const uint32_t* zones;
    uint32_t* gexit;
    uint32_t* gentry;
    uint32_t* wexit;
    uint32_t* dirty;
    uint32_t fix;  // bit 0: the launch from the groups' true entries; >> 8: rounds of the stitch after which a window is given up (0: never)
};
#define WK_TERM 0xffffu  // pointer jumping: the path has left the chunk / the window's targets
// a value every lane holds alike, into a scalar register (what comes out of LDS or a struct in memory lives in a vector register per
// lane otherwise -- and everything computed from it: the window's geometry cost k_lz_walk<true, true> 34 vector registers in scratch)
#define WK_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// lnk: per chunk a block of 4 x 65536 entries, [L4 | L6 | L8 | RK]
template <bool DEEP, bool STREAM = false>
__global__ __launch_bounds__(WK_THREADS, STREAM ? WK_STREAM_WAVES_PER_SIMD : WK_WAVES_PER_SIMD) void k_lz_walk(const uint8_t* __restrict__ in,
                                                                           const fl_chunk* __restrict__ chunks, fl_params prm,
                                                                           const uint16_t* __restrict__ lnk,
                                                                           const uint32_t* __restrict__ cflag,
                                                                           uint32_t* __restrict__ desc_all,
                                                                           uint32_t* __restrict__ true_all, wk_stream sp) {
    __shared__ uint32_t win32[WK_WIN_DW];
    __shared__ uint32_t sh_a, sh_b;           // STREAM: the group's entry / a window's exit
    __shared__ uint16_t tX[WK_THREADS];       // exit of a lane's own parse as soon as it is known, complemented (0xffff: not yet)
    __shared__ uint16_t tExg[WK_THREADS];     // exit the path is assumed to take out of a segment
    __shared__ uint16_t tNxt[2][WK_THREADS];  // segment that exit lands in (pointer jumping, double buffered)
    __shared__ uint16_t tEnt[WK_THREADS];     // position at which the path enters a segment
    __shared__ uint16_t tMark[WK_THREADS];    // segment is on the path
    const uint32_t tid = threadIdx.x;
    const uint32_t chain = prm.chain, good = prm.good, lazy = prm.lazy, nice = prm.nice;
    constexpr uint32_t LITD = STREAM ? 0u : PZ_DESC_LIT;  // descriptor of an anchor that emits one literal (k_st_emit: 0)
    fl_swin sw;
    sw.chunk = 0;
    sw.win0 = blockIdx.x;
    sw.nwin = 1;
    sw.wfirst = 0;
    sw.prev = ~0u;
    fl_chunk sck = chunks[0];
    uint32_t carry = 0;  // STREAM: the anchor at which the path enters the window (window-relative)
    bool guessed = false;
    if (STREAM) {
        sw = sp.swins[blockIdx.x];
        sck = sp.schunks[sw.chunk];
        if (sck.skip) return;
        if (sw.prev != ~0u) {
            if (sp.fix & 1u) {
                // the true entry: where the group before leaves (absolute stream position), read by ONE thread (k_lz_parse<true>)
                if (tid == 0) {
                    sh_a = __hip_atomic_load(&sp.gexit[sw.prev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh_b = sp.gentry[blockIdx.x];
                }
                __syncthreads();
                const uint32_t e = WK_UNI(sh_a), was = WK_UNI(sh_b);
                __syncthreads();
                if (e == was) return;  // parsed from there already
                if (tid == 0) sp.gentry[blockIdx.x] = e;
                carry = e - FL_MAX_DIST * sw.wfirst;
            } else {
                guessed = true;  // (the first window's first target: set below)
            }
        } else if (sp.fix & 1u) {
            return;  // a stream's first group starts at the stream's start
        }
    }
    for (uint32_t wi = 0; wi < sw.nwin; wi++) {
    const uint32_t ws = WK_UNI(sw.wfirst + wi);  // STREAM: the window's number in its stream
    const uint32_t c = WK_UNI(sw.win0 + wi);
    const fl_chunk ck = chunks[c];
    if (!STREAM && ck.skip) return;
    const uint32_t N = WK_UNI(ck.in_len);                           // positions below N have links
    const uint32_t NB = STREAM ? WK_UNI(N + (ck.pad_ >> 8)) : N;    // bytes of the window that exist
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;
    const uint8_t* src = in + ck.in_off;
    const uint64_t pos_off = STREAM ? sck.pos_off + (uint64_t)FL_MAX_DIST * ws : ck.pos_off;
    uint32_t* descg = desc_all + pos_off;
    uint32_t* trueg = true_all + (pos_off >> 5);
    if (N == 0) return;
    // the window's targets [t_first, t_last); the path enters at `carry`
    uint32_t t_first = 0, t_last = N;
    if (STREAM) {
        const uint32_t* zone = sp.zones + sck.zone_off;
        if (ws) t_first = WK_UNI(zone[ws - 1] - FL_MAX_DIST * ws);
        if (ws < sck.n_slides) t_last = WK_UNI(zone[ws] - FL_MAX_DIST * ws);
        if (wi) __syncthreads();  // the window before is done with the LDS tables
        if (guessed && wi == 0) {
            carry = t_first;
            if (tid == 0) sp.gentry[blockIdx.x] = t_first + FL_MAX_DIST * ws;
        }
        if (t_first >= t_last) {
            if (tid == 0) sp.wexit[2 * c] = sp.wexit[2 * c + 1] = ~0u;
            continue;
        }
        if (sp.fix & 1u) {
            // the anchors the first launch left in this window's targets go: bit by bit at the ends (the words there are shared
            // with the windows next to it, which other workgroups may be writing), whole words in between
            const uint64_t b0 = pos_off + t_first, b1 = pos_off + t_last;  // absolute bits [b0, b1)
            for (uint64_t w = (b0 >> 5) + tid; w <= ((b1 - 1) >> 5); w += WK_THREADS) {
                uint32_t keep = 0;
                if (w == (b0 >> 5) && (b0 & 31)) keep |= (1u << (b0 & 31)) - 1u;
                if (w == ((b1 - 1) >> 5) && (b1 & 31)) keep |= ~((1u << (b1 & 31)) - 1u);
                if (keep) atomicAnd(&true_all[w], keep); else true_all[w] = 0u;
            }
            __threadfence();
        }
    }
    // STREAM: a position at or beyond the window's last target is visited AFTER the next slide: the reference has dropped every
    // candidate at or below the next window's start by then (Lookup.zig:43-51) -- relative to this window: at or below 32768
    const uint32_t zt = (STREAM && ws < sck.n_slides) ? t_last : ~0u;
    const uint32_t kind = STREAM ? 2u : cflag[c];
    if (kind != 1u && (kind == 2u) != DEEP) return;  // (the other instantiation's chunk)
    if (kind == 1u && DEEP) return;
    if (kind == 1u) {
        // The chunk is one repeated byte (k_lz_links<0> saw it and built no chains): see k_lz_parse.
        for (uint32_t k = tid; 2 + FL_MAX_MATCH * k < N || k < 1; k += WK_THREADS) {
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
    const uint16_t* lk = lnk + (uint64_t)c * (4u * FL_CHUNK_STRIDE);  // level K: lk[(K << 16) + position]; RK: lk[(3 << 16) + position]
#ifdef WK_PROF
    uint32_t c_iter = 0, c_gath = 0, c_judge = 0, c_meas = 0, c_move = 0, c_rank = 0, c_runs = 0;
    uint64_t c_tstitch = 0, c_tloop = 0, c_tr0 = 0;
    const uint64_t c_t0 = __builtin_readcyclecounter();
#endif
    // ---- the chunk's bytes, zero padded
    {
        const uint32_t ash = (uint32_t)((uintptr_t)src & 3);
        const uint32_t* a32 = (const uint32_t*)(src - ash);
        const uint32_t ndw = (NB + ash + 3) >> 2;  // aligned dwords that hold at least one byte of the input
        constexpr uint32_t WB = (WK_WIN_DW + WK_THREADS - 1) / WK_THREADS;
        uint32_t lo[WB], hi[WB];
#pragma unroll
        for (uint32_t u = 0; u < WB; u++) {
            const uint32_t i = u * WK_THREADS + tid;
            lo[u] = i < ndw ? a32[i] : 0u;
            hi[u] = (ash && i + 1 < ndw) ? a32[i + 1] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < WB; u++) {
            const uint32_t i = u * WK_THREADS + tid;
            uint32_t v = __builtin_amdgcn_alignbyte(hi[u], lo[u], ash);
            if (4 * i + 4 > NB) v = 4 * i < NB ? (v & ((1u << (8 * (NB - 4 * i))) - 1u)) : 0u;
            if (i < WK_WIN_DW) win32[i] = v;
        }
    }
    // A chunk: the segments [64 m, 64 m + 64).  STREAM: the targets from the path's entry on, [carry, t_last), in segments of 32
    // positions when they are at most 32768 (every window but a stream's first and last: all 1024 lanes have one), else of 64;
    // segment m starts at sbase + (m << sg).
    const uint32_t t_end = STREAM ? t_last : N;
    const uint32_t sbase = STREAM ? carry : 0u;
    const uint32_t sg = (STREAM && t_end - sbase <= 32768u) ? 5u : 6u;
    const uint32_t SEGN = 1u << sg;
    const uint32_t nseg = (t_end - sbase + SEGN - 1) >> sg;
    const uint32_t m = tid;  // this lane's segment
    const bool active = m < nseg;
    const uint32_t seg0 = sbase + (m << sg);
    const uint32_t seg_end = min(seg0 + SEGN, t_end);
#define WK_SEG_OF(x) (((x) - sbase) >> sg)
    uint64_t A = 0, F = 0;  // anchors of the lane's own parse; of the parse from the entry
    uint32_t X = seg_end;   // exit of the lane's own parse
    uint32_t res_entry = PZ_NONE, res_exit = 0, Z = PZ_NONE;
    bool marked = false;
    tX[m] = (uint16_t)PZ_NONE;
    __syncthreads();
    // A segment that lies DEEP inside a run of one byte -- bytes [seg0 - 1, seg0 + 64 + 258) all equal -- is parsed in closed
    // form: from any entry e in it the reference finds (258, distance 1) at once (the candidate e - 1 is the head of e's
    // chain and matches 258 bytes >= nice >= lazy) and goes on at e + 258.  Such segments cost no trips, and a change of the
    // path's phase crosses a whole run of them inside ONE round of the stitch (4 KiB of padding took 16 rounds).
    bool deep = false;
    if (DEEP && m >= 1 && seg0 + SEGN + FL_MAX_MATCH <= N && seg0 + SEGN <= t_end) {
        const uint32_t x0 = seg0 - 1u, x1 = seg0 + SEGN + FL_MAX_MATCH;
        const uint32_t bp = (win32[x0 >> 2] >> (8u * (x0 & 3u)) & 0xffu) * 0x01010101u;
        deep = true;
        for (uint32_t x = x0; x < x1; x += 8) {
            uint32_t w0, w1;
            fl_lds_load8(win32, x, w0, w1);
            uint64_t d = (uint64_t)(w0 ^ bp) | ((uint64_t)(w1 ^ bp) << 32);
            if (x1 - x < 8u) d &= (1ull << (8u * (x1 - x))) - 1ull;
            if (d) {
                deep = false;
                break;
            }
        }
    }
    const bool wg_deep = DEEP && __syncthreads_or(deep ? 1 : 0) != 0;
    const uint32_t DEEP_DESC = 0x80000000u | ((uint32_t)(FL_MAX_MATCH - 3) << 15);  // j = 0, 258 bytes, distance 1

    enum { ST_SPEC = 0, ST_WAIT = 1, ST_FIX = 2, ST_DONE = 3 };
    for (uint32_t round = 0;; round++) {
        // (STREAM: a window whose stitch does not settle is given up -- periodic data; see k_lz_parse<true>)
        if (STREAM && (sp.fix >> 8) && round >= (sp.fix >> 8)) {
            if (tid == 0) atomicOr(sp.dirty, 0x80000000u);
            return;
        }
#ifdef WK_PROF
        if (tid == 0) atomicAdd((unsigned long long*)&g_fl_prof[43], 1ull);
#ifdef WK_ROUND_CAP
        if (round > WK_ROUND_CAP) {
            if (tid == 0) atomicAdd((unsigned long long*)&g_fl_prof[46], 1ull);
            break;
        }
#endif
#endif
        uint32_t st = ST_DONE;
        uint32_t a = 0;
        uint64_t stopmask = 0;
        uint32_t y_in = PZ_NONE;
        bool deferred = false; // this lane's entry is not the one it is resolved for, but may still move: next round
        bool fixing = false;   // this lane parses its segment again in this round ...
        uint32_t ex_used = 0;  // ... and this is the exit the round's path assumed for it
        bool moved = false;  // a deep segment got another entry in this round: the path has changed
        if (round == 0) {
            if (active) {
                st = ST_SPEC;
                a = seg0;
            }
            if (DEEP && deep) {  // the own parse in closed form; what is left is the wait for the lane before
                descg[seg0] = DEEP_DESC;
                A = 1ull;
                X = seg0 + FL_MAX_MATCH;
                __hip_atomic_store(&tX[m], (uint16_t)~X, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                st = ST_WAIT;
            }
        } else {
#ifdef WK_PROF
            const uint64_t c_ts0 = __builtin_readcyclecounter();
#endif
            // the path, assuming every segment not resolved yet leaves through its own exit
            if (active) {
                const uint32_t ex = res_entry != PZ_NONE ? res_exit : X;
                ex_used = ex;
                tExg[m] = (uint16_t)ex;
                tNxt[0][m] = (uint16_t)(ex >= t_end ? WK_TERM : WK_SEG_OF(ex));
                tMark[m] = m == 0 ? 1 : 0;
                tEnt[m] = m == 0 ? (uint16_t)seg0 : (uint16_t)PZ_NONE;
            }
            __syncthreads();
            uint32_t cur = 0;
            for (uint32_t step = 0; (1u << step) < nseg; step++) {
                if (active) {
                    const uint32_t n = tNxt[cur][m];
                    if (n != WK_TERM) {
                        if (tMark[m]) tMark[n] = 1;
                        tNxt[cur ^ 1][m] = tNxt[cur][n];
                    } else {
                        tNxt[cur ^ 1][m] = (uint16_t)WK_TERM;
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
            marked = active && tMark[m] != 0;
            if (marked) {
                const uint32_t ex = ex_used;  // (the lane's own, in 32 bits: STREAM exits reach 65536 + 264, tExg holds 16)
                if (ex < t_end) {
                    tEnt[WK_SEG_OF(ex)] = (uint16_t)ex;
                    tNxt[0][WK_SEG_OF(ex)] = (uint16_t)m;  // (the segment the path comes from; the jump tables are free now)
                }
            }
            __syncthreads();
            if (marked) y_in = tEnt[m];
            if (DEEP && wg_deep) {
                // deep segments take their new entries at once and hand their exits on: a run is crossed in this round
                for (uint32_t it = 0; it < 1024u; it++) {
                    bool upd = false;
                    if (deep && marked && y_in != res_entry) {
                        descg[y_in] = DEEP_DESC;
                        F = 1ull << (y_in - seg0);
                        Z = PZ_NONE;
                        res_entry = y_in;
                        res_exit = y_in + FL_MAX_MATCH;  // (< N: the segment is deep)
                        if (res_exit < t_end) {
                            tEnt[WK_SEG_OF(res_exit)] = (uint16_t)res_exit;
                            tNxt[0][WK_SEG_OF(res_exit)] = (uint16_t)m;
                        }
                        upd = true;
                        moved = true;
                    }
                    if (!__syncthreads_or(upd ? 1 : 0)) break;
                    const uint32_t e2 = tEnt[m];
                    if (active && e2 != PZ_NONE && e2 != y_in) {  // (handed on by a deep segment: on the path now)
                        y_in = e2;
                        marked = true;
                    }
                    __syncthreads();
                }
            }
            const bool need = marked && y_in != res_entry;
            // A segment is parsed again only when the segment the path comes from is settled for the entry IT got: else that
            // one's exit -- this one's entry -- may still move.  (A run of one byte is entered 258 bytes further in every
            // round, 16 rounds for 4 KiB of padding: everything behind the run was parsed again in every one of them.)
            tMark[m] = need ? 0 : 1;
            __syncthreads();
            const bool work = need && (m == 0 || tMark[tNxt[0][m]] != 0);
            deferred = need && !work;
#ifdef WK_PROF
            c_tstitch += __builtin_readcyclecounter() - c_ts0;
#endif
            // (moved: a deep segment has taken another entry than the path above assumed -- the marks of what lay behind its old
            // exit are stale; in a window that is one run to its end nobody else asks for the next round)
            if (!__syncthreads_or((need || moved) ? 1 : 0)) break;
            if (work) {
                st = ST_FIX;
                a = y_in;
                stopmask = A;
                fixing = true;
            }
        }
        // ---- the automaton (deflate.zig:154-205) over calls of the match finder
        uint64_t amask = 0;
        uint32_t j = 0, plen = 0, pdist = 0;
        // the call in progress: position p, best = match in hand (bdist its distance, 0: none accepted in this call),
        // K = level walked, q = the chain member in hand (the candidate is q - off), last = every candidate from here up
        // has been looked at
        uint32_t p = 0, best = 0, bdist = 0, maxlen = 0, lo = 1, last = 0, K = WK_L4, cnt = 0, budget = 0, q = 0, pref = 0, fo = 0;
        uint32_t off = 0;     // OFFSET MODE (L8 only): the walk follows the chain of p + off, off = best - 7 -- its members, moved
                              // back by off, share with p the LAST 8 of the best + 1 bytes a better candidate must share (records,
                              // markup: the first 8 are shared by thousands); the filter then looks at the first four bytes
        uint32_t prun = 0;    // bytes equal to the first from p on (capped at maxlen), if at least 4, else 0
        uint32_t pend_l = 0;  // length of a candidate of L6 / L8 that waits for the ranks (0: a probe; bit 31: a target inside a run)
        // what a lane waits for: MOVE the automaton's next move (between calls), WALK the link in flight = its next
        // candidate, MEAS the exact length of the candidate in hand, RANK the two ranks in flight, RUN the candidate
        // lies, like p, in a run of one repeated byte
        enum { CS_MOVE = 0, CS_WALK = 1, CS_MEAS = 2, CS_RANK = 3, CS_IDLE = 4, CS_RUN = 5 };
        uint32_t g_link = 0, g_rp = 0, g_rq = 0;  // values of the gathers in flight
        const uint32_t RUNT = 0x80000000u;
        const uint32_t PROBE_RUN = 0x40000000u;  // pend_l of a probe at the top of a run of p's own byte: within the budget, the run is looked at
        auto allsame8 = [&](uint32_t x, uint32_t& pat) {  // the 8 bytes at x are one repeated byte (pat = that byte, four times)
            uint32_t w0, w1;
            fl_lds_load8(win32, x, w0, w1);
            pat = (w0 & 0xffu) * 0x01010101u;
            return w0 == pat && w1 == pat;
        };
        auto offset_of = [&](uint32_t len) -> uint32_t {
            if (len < 8u || prun != 0) return 0u;
            if (STREAM && p + len + 1u > N) return 0u;  // (p + off + 8 <= N: a position whose 8-byte hash was made of the window's own bytes)
            uint32_t pat;
            return allsame8(p + len - 7u, pat) ? 0u : len - 7u;  // (a chain of run positions is the worst there is)
        };
        auto measure = [&](uint32_t cand) -> uint32_t {  // exact common prefix of cand and p, capped at maxlen
            uint32_t l = 0;
            for (;;) {
                WK_CNT(c_meas, 1);
                uint32_t a0, a1, b0, b1, a2, a3, b2, b3;
                fl_lds_load8(win32, p + l, a0, a1);
                fl_lds_load8(win32, cand + l, b0, b1);
                fl_lds_load8(win32, p + l + 8u, a2, a3);  // (four loads in flight: 16 bytes a step)
                fl_lds_load8(win32, cand + l + 8u, b2, b3);
                const uint64_t x = (uint64_t)(a0 ^ b0) | ((uint64_t)(a1 ^ b1) << 32);
                const uint64_t y = (uint64_t)(a2 ^ b2) | ((uint64_t)(a3 ^ b3) << 32);
                if (x) {
                    l += (uint32_t)__builtin_ctzll(x) >> 3;
                    break;
                }
                l += 8;
                if (l >= maxlen) break;
                if (y) {
                    l += (uint32_t)__builtin_ctzll(y) >> 3;
                    break;
                }
                l += 8;
                if (l >= maxlen) break;
            }
            return min(l, maxlen);
        };
        // bytes equal to pattern `bp` right below position `from`, counted down to `floor` at most: the lowest position t >= floor
        // with [t, from) all that byte
        auto scan_down = [&](uint32_t from, uint32_t floor, uint32_t bp) -> uint32_t {
            uint32_t t = from;
            // (16 bytes a step while that many are left: two loads in flight -- a scan is a chain of LDS round trips)
            while (t >= floor + 32u) {
                uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
                fl_lds_load8(win32, t - 8u, a0, a1);
                fl_lds_load8(win32, t - 16u, b0, b1);
                fl_lds_load8(win32, t - 24u, c0, c1);
                fl_lds_load8(win32, t - 32u, d0, d1);
                const uint64_t xa = (uint64_t)(a0 ^ bp) | ((uint64_t)(a1 ^ bp) << 32);
                const uint64_t xb = (uint64_t)(b0 ^ bp) | ((uint64_t)(b1 ^ bp) << 32);
                const uint64_t xc = (uint64_t)(c0 ^ bp) | ((uint64_t)(c1 ^ bp) << 32);
                const uint64_t xd = (uint64_t)(d0 ^ bp) | ((uint64_t)(d1 ^ bp) << 32);
                if (xa) return t - ((uint32_t)__builtin_clzll(xa) >> 3);
                if (xb) return t - 8u - ((uint32_t)__builtin_clzll(xb) >> 3);
                if (xc) return t - 16u - ((uint32_t)__builtin_clzll(xc) >> 3);
                if (xd) return t - 24u - ((uint32_t)__builtin_clzll(xd) >> 3);
                t -= 32u;
            }
            while (t >= floor + 16u) {
                uint32_t a0, a1, b0, b1;
                fl_lds_load8(win32, t - 8u, a0, a1);
                fl_lds_load8(win32, t - 16u, b0, b1);
                const uint64_t xa = (uint64_t)(a0 ^ bp) | ((uint64_t)(a1 ^ bp) << 32);
                const uint64_t xb = (uint64_t)(b0 ^ bp) | ((uint64_t)(b1 ^ bp) << 32);
                if (xa) return t - ((uint32_t)__builtin_clzll(xa) >> 3);
                if (xb) return t - 8u - ((uint32_t)__builtin_clzll(xb) >> 3);
                t -= 16u;
            }
            while (t > floor) {
                const uint32_t step = min(8u, t - floor);
                uint32_t w0, w1;
                fl_lds_load8(win32, t - step, w0, w1);  // bytes t - step .. t - step + 7: the first `step` of them count
                uint64_t x = (uint64_t)(w0 ^ bp) | ((uint64_t)(w1 ^ bp) << 32);
                if (step < 8u) x &= (1ull << (8u * step)) - 1ull;
                if (x) {
                    t -= (uint32_t)(__builtin_clzll(x) - (64 - 8 * (int)step)) >> 3;  // the equal bytes right below t
                    break;
                }
                t -= step;
            }
            return t;
        };
        // a parse that starts on a position where it has to stop already (FIX only)
        if (st == ST_FIX && ((stopmask >> ((a - seg0) & 63u)) & 1ull)) {
            F = 0;
            res_entry = y_in;
            Z = a;
            res_exit = X;
            st = ST_DONE;
        }
        uint32_t cs = st != ST_DONE ? CS_MOVE : CS_IDLE;
        // The walk leaves a run of p's own byte (pattern bp) in which nothing (more) can help: on below what is known to be run,
        // from position `from` down; what is skipped counts as looked at.  Round 6: on L8 with p itself at eight bytes of that
        // byte the chain's next member below the run's start is the top of the run BEFORE it -- found in the window when at
        // most eight other bytes lie between (sparse data: one), no link fetched: members of other strings with the same hash
        // are left out, they share fewer than K bytes with p and K - 1 are in hand (tools/multilevel_model.c, `shortcuts`).
        auto leave_run = [&](uint32_t from, uint32_t bp) {
            const uint32_t t = scan_down(from, from > WK_RUNSKIP ? from - WK_RUNSKIP : 0u, bp);
            cs = CS_WALK;
            if (K == WK_L4) {
                const uint32_t need = from - t;
                if (cnt < need) cs = CS_MOVE;
                cnt -= min(cnt, need);
                last = t;
            } else {
                cnt = cnt > 2u ? cnt - 2u : 0u;  // (a candidate of p's bucket is asked for its rank every few runs)
            }
            if (cs != CS_WALK) return;
            q = t;
            constexpr uint32_t KB = 8u;  // (L8 only: the member found holds eight bytes of the run's byte, like every member (R) is handed)
            if (K == WK_L8 && off == 0 && prun >= KB && t >= 2u) {
                const uint8_t* wb = (const uint8_t*)win32;
                const uint32_t b = bp & 0xffu;
                if (wb[t - 1u] != b) {  // (the run starts at t: the scan did not end on its own limit)
                    uint32_t e = t - 1u;
                    while (e > 0u && t - e <= 8u && wb[e - 1u] != b) e--;
                    if (e >= KB + lo && wb[e - 1u] == b) {
                        uint32_t w0, w1;
                        fl_lds_load8(win32, e - KB, w0, w1);
                        if (w0 == bp && w1 == bp) {
                            // the member e - KB, as if its link had arrived: (W)'s budget step, then the run or the ranks
                            q = e - KB;
                            if (cnt) cnt--;
                            if (cnt == 0) {
                                WK_CNT(c_gath, 2);
                                pend_l = PROBE_RUN;
                                g_rp = lk[(3u << 16) + p];
                                g_rq = lk[(3u << 16) + q];
                                cs = CS_RANK;
                            } else {
                                pend_l = 0;
                                cs = CS_RUN;
                            }
                            return;
                        }
                    }
                }
            }
            WK_CNT(c_gath, 1);
            g_link = lk[(K << 16) + t];
        };
        bool first_call = true;  // the next move is the first call of the parse (at a, nothing pending)
#ifdef WK_PROF
        const uint64_t c_tl0 = __builtin_readcyclecounter();
#endif
        // Lanes do the same thing at the same time: a wave's trip through this loop is (M) the automaton's move for the
        // lanes between calls, (W) up to WK_BURST chain steps for the lanes that walk -- a step is the arrival of a
        // link, the bounds, the four-byte filter, the request of the next link --, (R) runs, (J) the exact length of the
        // candidates that passed the filter and what follows from it.
        for (;;) {
            const uint64_t alive = __ballot(cs != CS_IDLE);
            if (alive == 0) break;
#ifdef WK_TRIP_CAP
            if (c_iter > WK_TRIP_CAP) {  // (debug builds: a parse that does not end is cut off and counted)
                if ((tid & 63) == 0) atomicAdd((unsigned long long*)&g_fl_prof[45], 1ull);
                break;
            }
#endif
            if (__ballot(st == ST_WAIT) == alive) __builtin_amdgcn_s_sleep(8);  // nothing to do but wait for another wave
            WK_CNT(c_iter, 1);
            // ---- (M) the automaton's move: one per lane and trip; every path through it ends in at most one new call
            if (cs == CS_MOVE) {
                WK_CNT(c_move, 1);
                bool start = false;
                uint32_t sp = 0, sl = 0;
                if (st == ST_WAIT) {
                    // the lane before has finished its own parse: where does that leave this segment?
                    const uint32_t vx = __hip_atomic_load(&tX[m - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t v = ~vx & 0xffffu;  // (stored as its complement: an exit is 1 .. 65535, 0xffff says "not yet")
                    if (vx != PZ_NONE) {
                        st = ST_DONE;
                        cs = CS_IDLE;
                        if (v >= seg0 && v < seg_end) {
                            y_in = v;
                            if ((A >> (v - seg0)) & 1ull) {  // on an anchor of the own parse
                                F = 0;
                                res_entry = v;
                                Z = v;
                                res_exit = X;
                            } else if (DEEP && deep) {  // (closed form)
                                descg[v] = DEEP_DESC;
                                F = 1ull << (v - seg0);
                                res_entry = v;
                                Z = PZ_NONE;
                                res_exit = v + FL_MAX_MATCH;
                            } else {
                                st = ST_FIX;
                                a = v;
                                stopmask = A;
                                amask = 0;
                                j = 0;
                                plen = 0;
                                start = true;
                                sp = v;
                            }
                        }
                    }
                } else if (first_call) {
                    first_call = false;
                    start = true;
                    sp = a;
                } else {
                    // the call has ended: the automaton's next move
                    bool emit = true;  // the pending match goes out (deflate.zig:182-184), or a literal
                    if (bdist) {       // a match, longer than the pending one if there is one
                        if (p != a) j++;  // the pending match's position becomes a literal (deflate.zig:166-168)
                        plen = best;
                        pdist = bdist;
                        emit = plen >= lazy;  // deflate.zig:171-173
                    }
                    if (emit) {
                        uint32_t desc = LITD, next = a + 1;
                        if (plen) {
                            desc = 0x80000000u | (j << 23) | ((plen - 3u) << 15) | (pdist - 1u);
                            next = a + j + plen;
                        }
                        descg[a] = desc;
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
                                // (STREAM: an exit beyond 65535 -- a match of the window's last anchors that ends in the bytes behind
                                // the window -- is handed on as 1: no segment but the first holds that, and 65536 would read "not yet")
                                __hip_atomic_store(&tX[m], (uint16_t)~((STREAM && a > 0xffffu) ? 1u : a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                amask = 0;
                                if (m == 0) {  // the first segment's own parse is the true one
                                    res_entry = seg0;
                                    res_exit = a;
                                    Z = seg0;
                                    st = ST_DONE;
                                    cs = CS_IDLE;
                                } else {
                                    st = ST_WAIT;
                                }
                            } else {
                                F = amask;
                                res_entry = y_in;
                                Z = meet ? a : PZ_NONE;
                                res_exit = meet ? X : a;
                                st = ST_DONE;
                                cs = CS_IDLE;
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
                if (start) {
                    // a call of the match finder at sp with a match of sl bytes in hand (deflate.zig:233-245)
                    p = sp;
                    best = sl;
                    bdist = 0;
                    maxlen = min(NB - p, (uint32_t)FL_MAX_MATCH);
                    lo = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;
                    if (STREAM && p >= zt) lo = max(lo, (uint32_t)FL_MAX_DIST + 1u);
                    budget = sl >= good ? (chain >> 2) : chain;
                    K = sl < 5u ? WK_L4 : (sl < 7u ? WK_L6 : WK_L8);
                    if (STREAM && p + 8u > N) K = WK_L4;  // (the hashes of 6 and 8 bytes of a window's last positions were made of zeros)
                    cnt = K == WK_L4 ? budget : WK_PROBE;
                    last = p;
                    prun = 0;
                    off = 0;
                    cs = CS_MOVE;  // (no hash entry / nothing longer is possible: the call finds nothing)
                    if (p < Mpos && maxlen > sl) {
                        uint32_t w0, w1;
                        fl_lds_load8(win32, p, w0, w1);
                        const uint32_t bp = (w0 & 0xffu) * 0x01010101u;
                        if (w0 == bp) {  // p starts with four equal bytes: a run of how many?
                            uint32_t r = 4;
                            {
                                const uint32_t x = w1 ^ bp;
                                r += x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
                            }
                            while (r >= 8u && r < maxlen) {  // (r < 8: the run has ended; else r is a multiple of 8 here)
                                uint32_t w2, w3;
                                fl_lds_load8(win32, p + r, w0, w1);
                                fl_lds_load8(win32, p + r + 8u, w2, w3);  // (two loads in flight: 16 bytes a step)
                                const uint64_t x = (uint64_t)(w0 ^ bp) | ((uint64_t)(w1 ^ bp) << 32);
                                const uint64_t y = (uint64_t)(w2 ^ bp) | ((uint64_t)(w3 ^ bp) << 32);
                                if (x) {
                                    r += (uint32_t)__builtin_ctzll(x) >> 3;
                                    break;
                                }
                                r += 8;
                                if (r >= maxlen) break;
                                if (y) {
                                    r += (uint32_t)__builtin_ctzll(y) >> 3;
                                    break;
                                }
                                r += 8;
                            }
                            prun = min(r, maxlen);
                        }
                        off = offset_of(sl);
                        fo = off ? 0u : (sl ? sl - 3u : 0u);
                        pref = pz_lds4(win32, p + fo);
                        WK_CNT(c_gath, 1);
                        g_link = lk[(K << 16) + p + off];  // the top of the chain is the position's own link
                        cs = CS_WALK;
                    }
                }
            }
            // ---- (W) chain steps
#pragma unroll 1
            for (int it = 0; it < WK_BURST; it++) {
                const uint64_t walkers = __ballot(cs == CS_WALK);
                if (walkers == 0) break;
                if (it != 0 && (uint32_t)__popcll(walkers) < WK_MINWALK) break;  // (the others have waited long enough)
                if (cs == CS_WALK) {
                    WK_CNT(c_judge, 1);
                    q = g_link;
                    if (q < lo + off || (K == WK_L4 && cnt == 0)) {
                        cs = CS_MOVE;  // the call has ended
                    } else {
                        const uint32_t qc = q - off;
                        bool next = true;  // the walk goes on with the link of q
                        const bool fresh = qc < last;  // (else: looked at before the walk changed chains)
                        if (fresh) {
                            if (cnt) cnt--;
                            if (K == WK_L4) last = qc;
                        }
                        uint32_t pat;
                        const bool run8 = allsame8(q, pat);  // the chain member lies in a run of one byte
                        const bool same = run8 && prun != 0 && off == 0 && pat == (win32[p >> 2] >> (8u * (p & 3u)) & 0xffu) * 0x01010101u;
                        if (run8 && !same) {
                            // ... and p + off does not start with that byte four times: a collision of the hash with a crowded
                            // bucket (zero padding: thousands of members).  No member of the run can be what the walk looks for.
                            cs = CS_RUN;
                            pend_l = fresh ? 1u : 2u;
                            next = false;
                        } else if (fresh) {
                            if (same && K != WK_L4 && cnt == 0) {
                                // Round 6: a walk on L6 / L8 that has just left a run of p's byte with nothing accepted (cnt = 0, see
                                // (R)) asks the ranks BEFORE it enters the next one: the member lies in p's L4 bucket (at least four
                                // bytes of the run's byte at both), and beyond the budget the call ends as the reference's does.
                                // Without it such a walk was never counted down -- sparse zeros at level 9 crossed all 330 runs of
                                // the window in every call where the reference looks at the nearest 10 to 42 (0.47 GB/s).
                                WK_CNT(c_gath, 2);
                                pend_l = PROBE_RUN;
                                g_rp = lk[(3u << 16) + p];
                                g_rq = lk[(3u << 16) + qc];
                                cs = CS_RANK;
                                next = false;
                            } else if (same) {
                                cs = CS_RUN;
                                pend_l = 0;
                                next = false;
                            } else if (pz_lds4(win32, qc + fo) == pref) {
                                cs = CS_MEAS;
                                next = false;
                            } else if (K != WK_L4 && cnt == 0 && pz_lds4(win32, qc) == pz_lds4(win32, p)) {
                                // A walk on L6 / L8 is not counted down, and one that finds nothing better would go on to the
                                // end of the window.  Every WK_PROBE steps a candidate of p's own L4 bucket (same first four
                                // bytes) is asked for its rank: beyond the budget ends the call.
                                WK_CNT(c_gath, 2);
                                pend_l = 0;
                                g_rp = lk[(3u << 16) + p];
                                g_rq = lk[(3u << 16) + qc];
                                cs = CS_RANK;
                                next = false;
                            }
                        }
                        if (next) {
                            WK_CNT(c_gath, 1);
                            g_link = lk[(K << 16) + q];
                        }
                    }
                }
            }
            // ---- (R) runs of one repeated byte (zero padding, sparse data): the reference's own worst case.  p starts with
            // r = prun bytes b, then another byte (or the end of what can match); the candidate q (off = 0) starts with 8
            // bytes b and lies in a run [s, E).  A position q' of that run matches p over min(r, E - q') bytes, more only at
            // q* = E - r, where the match may go on behind the runs.  With c bytes in hand the walk would take, one after
            // the other, every q' from min(q, max(E - c - 1, q*)) down to max(s, q*) -- each a byte longer than the last --
            // and nothing else of the run.  So it goes to the LAST of them at once (the first that reaches `nice`, or where
            // the budget or the distance ends, if that comes before), looks at that one as at any candidate, and below it
            // skips what is left of the run, WK_RUNSKIP bytes per trip at most.  Everything skipped lies in p's bucket and
            // counts against the budget as if it had been looked at; on L6 / L8 the ranks say where the budget ends.
            // CPU model of exactly this, checked against the oracle on run-heavy inputs: tools/multilevel_model.c.
            if (cs == CS_RUN && pend_l != 0) {
                // a run of another byte (see (W)): what is known to be run is skipped; it counts as looked at if it has not
                // been before (pend_l 1)
                WK_CNT(c_runs, 1);
                uint32_t pat;
                (void)allsame8(q, pat);
                const uint32_t t = scan_down(q, q > WK_RUNSKIP ? q - WK_RUNSKIP : 0u, pat);
                cs = CS_WALK;
                if (pend_l == 1u) {
                    if (K == WK_L4) {
                        const uint32_t need = q - t;
                        if (cnt < need) cs = CS_MOVE;
                        cnt -= min(cnt, need);
                        last = t;
                    } else {
                        cnt = 0;  // (the next candidate of p's bucket is asked for its rank)
                    }
                }
                pend_l = 0;
                if (cs == CS_WALK) {
                    q = t;
                    WK_CNT(c_gath, 1);
                    g_link = lk[(K << 16) + t];
                }
            } else if (cs == CS_RUN) {
                WK_CNT(c_runs, 1);
                const uint32_t bp = (win32[p >> 2] >> (8u * (p & 3u)) & 0xffu) * 0x01010101u;
                const uint32_t c = best, r = prun, cap = max(c, r) + 1u;
                uint32_t d = 8;  // bytes b from q on, counted up to cap + 1
                while (d <= cap) {
                    uint32_t w0, w1, w2, w3;
                    fl_lds_load8(win32, q + d, w0, w1);
                    fl_lds_load8(win32, q + d + 8u, w2, w3);
                    const uint64_t x = (uint64_t)(w0 ^ bp) | ((uint64_t)(w1 ^ bp) << 32);
                    const uint64_t y = (uint64_t)(w2 ^ bp) | ((uint64_t)(w3 ^ bp) << 32);
                    if (x) {
                        d += (uint32_t)__builtin_ctzll(x) >> 3;
                        break;
                    }
                    d += 8;
                    if (d > cap) break;
                    if (y) {
                        d += (uint32_t)__builtin_ctzll(y) >> 3;
                        break;
                    }
                    d += 8;
                }
                int32_t target = -1, hi = -1;
                if (d <= cap) {  // the run ends at E = q + d
                    const int32_t E = (int32_t)(q + d), qstar = E - (int32_t)r;
                    hi = max(E - (int32_t)c - 1, qstar);
                    hi = min(hi, (int32_t)q);
                    if (qstar <= hi) {
                        const uint32_t lo_u = scan_down(q, (uint32_t)max(qstar, 0), bp);  // max(s, q*)
                        if (hi >= (int32_t)lo_u) target = max((int32_t)lo_u, min(hi, E - (int32_t)nice));
                    }
                }
                // (below q* every position matches exactly r bytes: the first one met helps if less than that is in hand)
                if (target < 0 && c < r && d > r) {
                    target = (int32_t)q;
                    hi = (int32_t)q;
                }
                if (target < 0) {
                    // nothing here can help
                    leave_run(q, bp);
                } else {
                    if (target < (int32_t)lo) target = (int32_t)lo;  // (beyond the distance nothing is looked at)
                    if (K == WK_L4) {
                        if (cnt < q - (uint32_t)target) target = (int32_t)(q - cnt);
                        cnt -= q - (uint32_t)target;
                    }
                    if (target > hi) {
                        cs = CS_MOVE;  // budget / distance end among positions that cannot help
                    } else {
                        if (K == WK_L4) last = (uint32_t)target; else cnt = 0;
                        q = (uint32_t)target;
                        g_link = (uint32_t)hi;  // (kept for the case that the budget ends inside the run)
                        pend_l = RUNT;
                        cs = CS_MEAS;
                    }
                }
            }
            // ---- (J) the exact length of the candidates in hand; ranks that have arrived
            if (cs == CS_RANK || cs == CS_MEAS) {
                const uint32_t qc = q - off;
                bool accept = false, left = false;  // left: leave_run has said where the walk goes on
                uint32_t l = pend_l & ~(RUNT | PROBE_RUN);
                if (cs == CS_RANK) {
                    WK_CNT(c_rank, 1);
                    const uint32_t dt = g_rp - g_rq;
                    cs = CS_WALK;
                    cnt = WK_PROBE;
                    if (dt <= budget) {
                        if (pend_l == PROBE_RUN) cs = CS_RUN;  // within the budget: the run, in the next trip's (R)
                        accept = l != 0;  // (0: a probe)
                    } else {
                        cs = CS_MOVE;  // beyond what the reference looks at, and so is everything behind it
                        if (pend_l & RUNT) {
                            // ... but the budget ends inside this run (its members have consecutive ranks): the last one within
                            // it is the one the reference ends on
                            const uint32_t t2 = qc + (dt - budget);
                            if (t2 <= g_link) {
                                const uint32_t l2 = measure(t2);
                                if (l2 > best) {
                                    best = l2;
                                    bdist = p - t2;
                                }
                            }
                        }
                    }
                    pend_l = 0;
                } else {
                    const bool runt = (pend_l & RUNT) != 0;
                    l = measure(qc);
                    if (l >= FL_MIN_MATCH && l > best) {  // deflate.zig:254-261
                        if (K == WK_L4) {
                            accept = true;
                            pend_l = 0;
                        } else {
                            pend_l = l | (runt ? RUNT : 0u);
                            WK_CNT(c_gath, 2);
                            g_rp = lk[(3u << 16) + p];
                            g_rq = lk[(3u << 16) + qc];
                            cs = CS_RANK;
                        }
                    } else if (runt) {
                        // (round 6) the run's best position did not beat what is in hand: below it every position of the run
                        // matches exactly prun <= best bytes -- the walk leaves the run at once
                        pend_l = 0;
                        leave_run(qc, (win32[p >> 2] >> (8u * (p & 3u)) & 0xffu) * 0x01010101u);
                        left = true;
                    } else {
                        pend_l = 0;
                        cs = CS_WALK;  // the walk goes on behind the candidate
                    }
                }
                if (accept) {
                    best = l;
                    bdist = p - qc;
                    last = qc;
                    cs = CS_WALK;
                    if (l >= nice || l >= maxlen) {
                        cs = CS_MOVE;  // good enough / nothing longer is possible
                    } else {
                        uint32_t K2 = l < 5u ? WK_L4 : (l < 7u ? WK_L6 : WK_L8);
                        if (STREAM && p + 8u > N) K2 = WK_L4;
                        const uint32_t off2 = offset_of(l);
                        if (K2 != K || off2 != off) {  // on to another chain, from its top (what is done is skipped there)
                            K = K2;
                            off = off2;
                            q = p + off;
                            if (K != WK_L4) cnt = WK_PROBE;
                        }
                        fo = off ? 0u : l - 3u;
                        pref = pz_lds4(win32, p + fo);
                    }
                }
                if (cs == CS_WALK && !left) {
                    WK_CNT(c_gath, 1);
                    g_link = lk[(K << 16) + q];
                }
            }
        }
#ifdef WK_PROF
        c_tloop += __builtin_readcyclecounter() - c_tl0;
        if (round == 0) c_tr0 = c_tloop;
#endif
        // Every segment parsed again in this round leaves where the round's path assumed: the path stands, and with
        // it the marks and entries found above -- no round to confirm it.
        if (round >= 1 && !__syncthreads_or(((fixing && res_exit != ex_used) || deferred || moved) ? 1 : 0)) break;
    }
    // ---- the true anchors
    if (active) {
        uint64_t T = 0;
        if (marked) {
            T = F;
            if (Z != PZ_NONE) T |= A & (~0ull << (Z - seg0));
        }
        // (segments are 64 positions: two whole words of the bitmap the host has cleared; STREAM: the words at the ends of the
        // window's targets are shared with the windows next to it)
        if (STREAM) {  // (a segment need not start on a word boundary)
            const uint32_t sh = seg0 & 31u;
            const uint64_t lo64 = T << sh;
            const uint32_t w0 = (uint32_t)lo64, w1 = (uint32_t)(lo64 >> 32), w2 = sh ? (uint32_t)(T >> (64u - sh)) : 0u;
            if (w0) atomicOr(&trueg[seg0 >> 5], w0);
            if (w1) atomicOr(&trueg[(seg0 >> 5) + 1], w1);
            if (w2) atomicOr(&trueg[(seg0 >> 5) + 2], w2);
        } else {
            if ((uint32_t)T) trueg[seg0 >> 5] = (uint32_t)T;
            if ((uint32_t)(T >> 32)) trueg[(seg0 >> 5) + 1] = (uint32_t)(T >> 32);
        }
    }
    if (STREAM) {
        // where the path leaves the window: the exit of the one segment on the path that leaves the targets
        __syncthreads();
        if (marked && res_exit >= t_end) sh_b = res_exit;
        __syncthreads();
        const uint32_t ex = WK_UNI(sh_b);
        const uint32_t was = sp.wexit[2 * c];
        __syncthreads();
        if (tid == 0) {
            sp.wexit[2 * c] = ex;
            sp.wexit[2 * c + 1] = ~0u;
        }
        if ((sp.fix & 1u) && was == ex) return;  // from here on the first launch's anchors stand
        carry = ex - FL_MAX_DIST;  // (a window that is not the stream's last is left at or beyond 65274)
        if (wi + 1 == sw.nwin && tid == 0) {
            // the group's exit; in a fix launch: a NEW one -- the group behind has to be parsed again from it
            __hip_atomic_store(&sp.gexit[blockIdx.x], ex + FL_MAX_DIST * ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sp.fix & 1u) atomicAdd(sp.dirty, 1u);
        }
    }
#ifdef WK_PROF
    if ((tid & 63) == 0) {
        atomicAdd((unsigned long long*)&g_fl_prof[40], (unsigned long long)c_iter);
        atomicAdd((unsigned long long*)&g_fl_prof[41], (unsigned long long)(__builtin_readcyclecounter() - c_t0));
        atomicAdd((unsigned long long*)&g_fl_prof[42], 1ull);
        atomicAdd((unsigned long long*)&g_fl_prof[47], (unsigned long long)c_runs);
        atomicAdd((unsigned long long*)&g_fl_prof[48], (unsigned long long)c_meas);
        atomicAdd((unsigned long long*)&g_fl_prof[49], (unsigned long long)c_judge);
        atomicAdd((unsigned long long*)&g_fl_prof[50], (unsigned long long)c_move);
        atomicMax((unsigned long long*)&g_fl_prof[51], (unsigned long long)c_iter);
        atomicAdd((unsigned long long*)&g_fl_prof[52], (unsigned long long)c_tstitch);
        atomicAdd((unsigned long long*)&g_fl_prof[53], (unsigned long long)c_tloop);
        atomicAdd((unsigned long long*)&g_fl_prof[54], (unsigned long long)c_tr0);
    }
    atomicAdd((unsigned long long*)&g_fl_prof[55], (unsigned long long)fl_wave_sum(c_gath));
    atomicAdd((unsigned long long*)&g_fl_prof[56], (unsigned long long)fl_wave_sum(c_rank));
#endif
#undef WK_SEG_OF
    }  // windows
}
