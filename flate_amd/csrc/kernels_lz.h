// kernels_lz.h -- the hash-order match finder (round 1-2) and helpers shared by the tokenizer kernels.
// k_lz_sort and k_lz_match serve, as <true> instances, the tiles of the whole-stream path (kernels_stream.h: inputs
// longer than 65535 bytes at levels 4..9).  The chunk path (inputs of at most 65535 bytes) no longer uses them:
// levels 4-7 take kernels_parse.h (round 3), levels 8-9 kernels_walk.h (round 4); the <false> code paths below are
// what is left of the chunk-path use and are not instantiated.
//
// Reference path: Deflate.tokenize / findMatch (deflate.zig:154-266),
// SlidingWindow.match (SlidingWindow.zig:81-104), Lookup (Lookup.zig:12-84).
//
// The reference walks a hash chain sequentially.  The chain content does not
// depend on the parse: every position is inserted exactly once, in ascending
// order (deflate.zig:207-211,236; Lookup.zig:55-72), so chain[p] is simply the
// nearest earlier position with the same 15-bit hash:
//
//   k_lz_sort   stable LSD radix sort of the positions by hash: the candidates of a
//               position are its predecessors in its bucket, nearest first.
//   k_lz_match  for EVERY position, the longest-match record the reference's findMatch
//               would return, for the full chain budget and for chain >> 2
//               (deflate.zig:241-245); window staged in LDS.  Two instantiations:
//               plain windows, and windows the sort marks runny (runs, padding, records).
//
// Bounds: k_lz_match is bound by vector-ALU issue, k_lz_sort by the CU's memory pipe
// (one scattered gather and one scattered store per element) and issue; no MFMA.
#pragma once
#include "kernels_common.h"

// Lookup.zig:59-63,82-84: big-endian 4-byte word * 0x9E3779B1 >> 17
__device__ __forceinline__ uint32_t fl_hash_le(uint32_t le_word) {
    return (__builtin_bswap32(le_word) * 0x9E3779B1u) >> 17;
}

// peers of this lane: lanes (within `valid`) holding the same NB-bit digit
template <int NB>
__device__ __forceinline__ uint64_t fl_match_any(uint32_t d, uint64_t valid) {
    uint64_t peers = valid;
#pragma unroll
    for (int bit = 0; bit < NB; bit++) {
        const bool s = (d >> bit) & 1;
        const uint64_t m = __ballot(s);
        peers &= s ? m : ~m;
    }
    return peers;
}

// in-place exclusive scan of cnt[0..n) (n <= 256) by one wave
__device__ __forceinline__ void fl_wave_excl_scan_lds(uint32_t* cnt, uint32_t n, uint32_t lane) {
    const uint32_t per = (n + 63) / 64;  // <= 4
    uint32_t v[4];
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = lane * per + k;
        v[k] = i < n ? cnt[i] : 0;
        s += v[k];
    }
    const uint32_t incl = fl_wave_incl_scan(s, lane);
    uint32_t run = incl - s;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = lane * per + k;
        if (i < n) cnt[i] = run;
        run += v[k];
    }
}

// smallest sync-flush point > pos, else n (fp ascending).  A flush at F ends the lookahead of
// everything before it: matches stop at F and positions F-3 .. F-1 never enter the hash table
// (deflate.zig:196-203 runs the tokenizer dry, Lookup.zig:23-27 needs 4 bytes).
__device__ __forceinline__ uint32_t fl_next_flush(const uint32_t* __restrict__ fp, uint32_t n_flush, uint32_t pos,
                                                  uint32_t n) {
    uint32_t lo = 0, hi = n_flush;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (fp[mid] > pos)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo < n_flush ? fp[lo] : n;
}

// ------------------------------------------------------------------ k_lz_sort
// One workgroup (16 waves) per chunk.  Output: S[c][0..M) = the positions 0..M-1
// (M = in_len - 3: those with 4 bytes left, Lookup.zig:24) sorted by (hash, position).
//
// Stable LSD radix sort, 8 + 7 bits.  Wave w owns elements [4096 w, 4096 (w+1)) of each
// pass and keeps its own digit counters, so the scatter is stable without atomics on the
// destination.  Pass 1 scatters 16-bit positions inside LDS (128 KiB); pass 2 scatters to
// global memory as 128 sequential 2-byte write streams which the L2 / Infinity Cache merge
// into whole lines because only one chunk per CU is in flight.  A scattered vector-memory
// instruction costs the CU's memory pipe per lane, so pass 2 issues exactly two of them per
// element batch: the 8-byte gather that recomputes the hash and the 2-byte store.
#define FL_SORT_WAVES 16
#define FL_SORT_THREADS (64 * FL_SORT_WAVES)
#define FL_SORT_SLICE 4096u

__device__ __forceinline__ uint32_t fl_load_u32_clamped(const uint8_t* src, uint32_t p, uint32_t N) {
    // bytes p..p+3 of the chunk, zero beyond N.  Touches only aligned dwords that contain at
    // least one valid byte: never another chunk's bytes in the key, never memory past the buffer.
    if (p >= N) return 0;
    const uint32_t nb = min(N - p, 4u);
    const uint32_t sh = (uint32_t)((uintptr_t)(src + p) & 3);
    const uint32_t* w = (const uint32_t*)(src + p - sh);
    const uint32_t lo = w[0];
    const uint32_t hi = (sh + nb > 4) ? w[1] : 0u;
    uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sh);
    if (nb < 4) v &= (1u << (8 * nb)) - 1;
    return v;
}

struct __attribute__((packed, aligned(4))) fl_u32x2 {
    uint32_t a, b;
};
// bytes p..p+3 (p <= N - 4) with a single 8-byte load whenever both dwords are inside the chunk
__device__ __forceinline__ uint32_t fl_gather_u32(const uint8_t* src, uint32_t p, uint32_t N) {
    if (p + 8 <= N) {
        const uint32_t sh = (uint32_t)((uintptr_t)(src + p) & 3);
        const fl_u32x2 w = *(const fl_u32x2*)(src + p - sh);
        return __builtin_amdgcn_alignbyte(w.b, w.a, sh);
    }
    return fl_load_u32_clamped(src, p, N);
}

// Exclusive scan of a [waves][ndig] counter table in (digit major, wave minor) order, in
// place, by the whole workgroup.  PER = entries per thread (ndig * 16 / 1024).
template <int NDIG, int PER>
__device__ __forceinline__ void fl_scan_counters(uint32_t (*cnt)[NDIG], uint32_t* wsum, uint32_t tid) {
    const uint32_t lane = tid & 63, wave = tid >> 6;
    uint32_t v[PER];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t e = tid * PER + k;  // e = d * 16 + w
        v[k] = cnt[e & 15][e >> 4];
        s += v[k];
    }
    const uint32_t incl = fl_wave_incl_scan(s, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t e = tid * PER + k;
        cnt[e & 15][e >> 4] = run;
        run += v[k];
    }
    __syncthreads();
}

// STREAM: the unit is a tile of a long stream (fl_tile): the window starts w0 bytes into the
// chunk, N is what is left of the stream from there and at most 65536 positions are sorted.
template <bool STREAM>
__global__ __launch_bounds__(FL_SORT_THREADS) void k_lz_sort(const uint8_t* __restrict__ in,
                                                              const fl_chunk* __restrict__ chunks,
                                                              const fl_tile* __restrict__ tiles,
                                                              const uint32_t* __restrict__ fpts,
                                                              uint32_t* __restrict__ n_sorted,
                                                              uint16_t* __restrict__ S, uint32_t* __restrict__ cflag) {
    __shared__ uint16_t tmp[65536];
    __shared__ uint32_t cnt1[FL_SORT_WAVES][256];
    __shared__ uint32_t cnt2[FL_SORT_WAVES][128];
    __shared__ uint32_t wsum[FL_SORT_WAVES];
    const uint32_t c = blockIdx.x;
    const uint32_t w0 = STREAM ? tiles[c].w0 : 0u;
    const fl_chunk ck = chunks[STREAM ? tiles[c].chunk : c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = ck.in_len - w0;
    const uint32_t M = min(N >= 4 ? N - 3 : 0u, 65536u);  // positions with 4 bytes left in the stream
    const uint8_t* src = in + ck.in_off + w0;
    uint16_t* So = S + (uint64_t)c * FL_CHUNK_STRIDE;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t slice0 = wave * FL_SORT_SLICE;
    // sync-flush points inside the window take the 3 positions before them out of the table
    const uint32_t* fp = STREAM ? fpts + ck.flush_off : nullptr;
    const bool has_fl = STREAM && ck.n_flush && fl_next_flush(fp, ck.n_flush, w0, ck.in_len) <= w0 + M + 2;
    auto hashed = [&](uint32_t p) -> bool {
        if (p >= M) return false;
        if (!STREAM || !has_fl) return true;
        return fl_next_flush(fp, ck.n_flush, w0 + p, ck.in_len) - (w0 + p) >= 4;
    };
    __shared__ uint32_t n_hashed;
    if (tid == 0) n_hashed = 0;
    // A window of one repeated byte (a run of zeros, say) needs no sort: k_lz_match writes its records
    // directly (every position matches its predecessor over the full length).  It is one iff every
    // 4-byte word equals the first one.
    const uint32_t word0 = M ? fl_load_u32_unaligned(src) : 0u;
    bool same = cflag != nullptr && M >= 64 && !(STREAM && ck.n_flush);

    // (looked for before the counting pass -- whose LDS atomics would all hit one counter -- whenever the
    // first words of every wave's slice already agree)
    if (__syncthreads_and(!same || slice0 + lane >= M || fl_load_u32_unaligned(src + slice0 + lane) == word0) && same) {
        for (uint32_t p = tid; p < M; p += FL_SORT_THREADS) same = same && fl_load_u32_unaligned(src + p) == word0;
        if (STREAM)  // the lookahead of the last targets (k_lz_match stages 288 bytes past the 65536 positions)
            for (uint32_t q = 65536u + tid; q < min(N, 65536u + 288u); q += FL_SORT_THREADS)
                same = same && src[q] == (uint8_t)word0;
        if (__syncthreads_and(same)) {
            if (tid == 0) {
                cflag[c] = 1u;
                if (STREAM) n_sorted[c] = M;
            }
            return;
        }
    }
    fl_prof_mark(0);
    for (uint32_t i = tid; i < FL_SORT_WAVES * 256; i += FL_SORT_THREADS) (&cnt1[0][0])[i] = 0;
    for (uint32_t i = tid; i < FL_SORT_WAVES * 128; i += FL_SORT_THREADS) (&cnt2[0][0])[i] = 0;
    __syncthreads();
    // ---- pass 1 count: low 8 bits of the hash ----
    for (uint32_t r = 0; r < FL_SORT_SLICE / 64; r += 8) {
        uint32_t w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = slice0 + (r + u) * 64 + lane;
            w[u] = p < M ? fl_load_u32_unaligned(src + p) : 0;
        }
        uint32_t nv = 0;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (hashed(slice0 + (r + u) * 64 + lane)) {
                atomicAdd(&cnt1[wave][fl_hash_le(w[u]) & 255], 1u);
                nv++;
            }
        if (STREAM) {
            nv = fl_wave_sum(nv);
            if (lane == 0 && nv) atomicAdd(&n_hashed, nv);
        }
    }
    __syncthreads();
    const uint32_t Ms = STREAM ? n_hashed : M;  // entries of the sorted array
    if (STREAM && tid == 0) n_sorted[c] = Ms;
    // "runny" windows (one low hash digit holds a sixteenth of the positions: runs, padding, repeated
    // records, short periods) go to the RJ variant of k_lz_match
    {
        bool big = false;
        if (tid < 256) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < FL_SORT_WAVES; w++) t += cnt1[w][tid];
            big = t >= max(Ms >> 4, 128u);
        }
        const int runny = __syncthreads_or(big);
        if (cflag != nullptr && tid == 0) cflag[c] = runny ? 2u : 0u;
    }
    fl_prof_mark(1);
    fl_scan_counters<256, 4>(cnt1, wsum, tid);
    fl_prof_mark(2);
    // ---- pass 1 scatter into LDS; count pass 2's digits per destination slice ----
    uint32_t wn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t p = slice0 + u * 64 + lane;
        wn[u] = p < M ? fl_load_u32_unaligned(src + p) : 0;
    }
    for (uint32_t r = 0; r < FL_SORT_SLICE / 64; r += 4) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = wn[u];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // next group's loads fly while this one is ranked
            const uint32_t p = slice0 + (r + 4 + u) * 64 + lane;
            wn[u] = (r + 4 < FL_SORT_SLICE / 64 && p < M) ? fl_load_u32_unaligned(src + p) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = slice0 + (r + u) * 64 + lane;
            const bool valid = hashed(p);
            const uint32_t h = fl_hash_le(w[u]);
            const uint32_t d = h & 255;
            const uint64_t peers = fl_match_any<8>(d, __ballot(valid));
            const uint32_t rank = __popcll(peers & lt_mask), np = __popcll(peers);
            if (valid) {
                const uint32_t dst = cnt1[wave][d] + rank;
                tmp[dst] = (uint16_t)p;
                atomicAdd(&cnt2[dst >> 12][h >> 8], 1u);
            }
            fl_lds_order();
            if (valid && rank == np - 1) cnt1[wave][d] += np;
            fl_lds_order();
        }
    }
    __syncthreads();
    fl_prof_mark(3);
    fl_scan_counters<128, 2>(cnt2, wsum, tid);
    fl_prof_mark(4);
    // ---- pass 2 scatter: high 7 bits, to global memory ----
    uint32_t ppn[4], a0n[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = slice0 + u * 64 + lane;
        ppn[u] = e < Ms ? tmp[e] : 0;
        a0n[u] = e < Ms ? fl_gather_u32(src, ppn[u], N) : 0;
    }
    for (uint32_t r = 0; r < FL_SORT_SLICE / 64; r += 4) {
        uint32_t pp[4], a0[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            pp[u] = ppn[u];
            a0[u] = a0n[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {  // next group's gathers fly while this one is ranked
            const uint32_t e = slice0 + (r + 4 + u) * 64 + lane;
            const bool okn = r + 4 < FL_SORT_SLICE / 64 && e < Ms;
            ppn[u] = okn ? tmp[e] : 0;
            a0n[u] = okn ? fl_gather_u32(src, ppn[u], N) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t e = slice0 + (r + u) * 64 + lane;
            const bool valid = e < Ms;
            const uint32_t d = fl_hash_le(a0[u]) >> 8;
            const uint64_t peers = fl_match_any<7>(d, __ballot(valid));
            const uint32_t rank = __popcll(peers & lt_mask), np = __popcll(peers);
            if (valid) So[cnt2[wave][d] + rank] = (uint16_t)pp[u];
            fl_lds_order();
            if (valid && rank == np - 1) cnt2[wave][d] += np;
            fl_lds_order();
        }
    }
    fl_prof_mark(5);
}

// ------------------------------------------------------------------ k_lz_match
#define FL_MATCH_WAVES 16
#define FL_MATCH_THREADS (64 * FL_MATCH_WAVES)
#define FL_KB 32                 // candidates per tile
#define FL_TILE (FL_KB + 64)     // FL_KB back + 64 lanes

__device__ __forceinline__ uint32_t fl_lds_load4(const uint32_t* win32, uint32_t off) {
    const uint32_t i = off >> 2;
    return __builtin_amdgcn_alignbyte(win32[i + 1], win32[i], off);  // (v_alignbyte_b32 shifts by the low two bits of its third operand)
}
// window bytes off..off+3 and off+4..off+7
__device__ __forceinline__ void fl_lds_load8(const uint32_t* win32, uint32_t off, uint32_t& w0, uint32_t& w1) {
    const uint32_t i = off >> 2, sh = off;  // (v_alignbyte_b32 shifts by the low two bits)
    const uint32_t d0 = win32[i], d1 = win32[i + 1], d2 = win32[i + 2];
    w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
    w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
}

// exact common prefix of the window at p and q, known to be >= len0, capped at maxlen
__device__ __forceinline__ uint32_t fl_extend_len(const uint32_t* win32, uint32_t p, uint32_t q, uint32_t len0,
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
// ... known to be >= 8
__device__ __forceinline__ uint32_t fl_extend_match(const uint32_t* win32, uint32_t p, uint32_t q, uint32_t maxlen) {
    uint32_t len = 8;
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

// rec[c][p] = { record for the full chain budget, record for chain >> 2 }
//             (0 = no match, else len << 16 | dist-1)
//
// Lane = one sorted entry, loop = its chain candidates (the preceding entries of its hash
// bucket, nearest first).  The first 8 bytes of every entry are gathered from the LDS window
// when a tile of sorted entries is loaded, so a candidate whose first 8 bytes settle the
// comparison (the common case) costs ~15 VALU ops on registers; the window is read again only
// to extend a match beyond 8 bytes.
//
// How the reference's walk (deflate.zig:233-266) is encoded branch-free:
//  * a lane may look at min(bucket offset, chain) candidates.  Positions fall along the
//    chain, so "candidate number <= n" is the same as "position >= position of candidate n";
//    that bound is merged with the window limit p - 32768 and the null position 0 into one
//    per-lane lower bound `lov` (entries of other buckets that follow in the tile differ in
//    their first four bytes, because the hash is a function of those bytes).
//  * the best match so far is one 32-bit key  len << 16 | (65535 - dist):  a longer match
//    wins, and for equal length the nearer one, which is the one the reference met first.
//
// STREAM (whole-stream passes): the unit is a tile (fl_tile).  Only entries at window positions
// >= tgt0 are searched; the window carries 288 bytes of lookahead past its 65536 positions so
// that a match of the last target can run its full 258 bytes; N is what is left of the stream.
// After the reference slides its window (deflate.zig:291-294, Lookup.zig:43-51) everything at or
// below the new window start is gone from the chains: positions visited after slide number j
// (those at buffer offsets >= 65274 = 65536 - min_lookahead, SlidingWindow.zig:56-60, provided
// the window did fill up, i.e. N >= 65536) may only use candidates above window offset 32768.
#define FL_WIN_DW_CHUNK (16384 + 8)
#define FL_WIN_DW_STREAM (16384 + 72)
#define FL_ZONE_START (65536u - (FL_MAX_MATCH + 4u))  // deflate.zig:159-163 min_lookahead = 262
// BF: the candidates of a tile are scored branch-free (two XORs, the trailing-equal-bytes mask, the
// position bound, a packed score "equal bytes, then nearest", a max, and one bit "agrees in all 8
// prefix bytes"); only the tile's winner meets the lane's key, at the end of the tile.  Same
// result: the key is a maximum.
// RJ: the variant for windows k_lz_sort has marked "runny" (cflag 2: one hash digit holds a sixteenth of
// the positions -- runs, padding, repeated records, short periods).  It adds the byte filter for groups of candidates
// and the wave-wide compare described at flush_deep_rj below; plain windows run the plain variant,
// whose code these additions would slow by 10 % (measured on the benchmark text: 26.0 -> 28.9 ms).
template <bool STREAM, bool BF, bool RJ = false>
__global__ __launch_bounds__(FL_MATCH_THREADS, 8) void k_lz_match(const uint8_t* __restrict__ in,
                                                                  const fl_chunk* __restrict__ chunks,
                                                                  const fl_tile* __restrict__ tiles,
                                                                  const uint32_t* __restrict__ fpts,
                                                                  const uint32_t* __restrict__ n_sorted, fl_params prm,
                                                                  const uint16_t* __restrict__ S,
                                                                  uint32_t* __restrict__ NQ,
                                                                  uint32_t* __restrict__ rec_all,
                                                                  const uint32_t* __restrict__ cflag) {
    constexpr uint32_t WIN_DW = STREAM ? FL_WIN_DW_STREAM : FL_WIN_DW_CHUNK;
    __shared__ uint32_t win32[WIN_DW];
    __shared__ uint2 tW[FL_MATCH_WAVES][FL_TILE];
    __shared__ uint16_t tS[FL_MATCH_WAVES][FL_TILE];
    __shared__ uint32_t wlast[FL_MATCH_WAVES];
    const uint32_t c = blockIdx.x;
    const uint32_t w0 = STREAM ? tiles[c].w0 : 0u;
    const uint32_t tgt0 = STREAM ? tiles[c].tgt0 : 0u;
    const fl_chunk ck = chunks[STREAM ? tiles[c].chunk : c];
    if (ck.skip) return;
    if (((cflag != nullptr && cflag[c] == 2u) ? true : false) != RJ) return;  // the other variant's window
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
    uint32_t* NQc = NQ + (uint64_t)c * FL_CHUNK_STRIDE;
    uint2* rec2 = (uint2*)rec_all + ck.pos_off + w0;
    const uint32_t chain = prm.chain, quarter = prm.chain >> 2, nice = prm.nice;

    fl_prof_mark(8);
    // positions without a hash entry never match (Lookup.zig:24)
    // (with flush points in the stream the host has cleared all records beforehand)
    for (uint32_t p = Mpos + tid; p < min(N, 65536u); p += FL_MATCH_THREADS) rec2[p] = make_uint2(0u, 0u);
    if (cflag != nullptr && cflag[c] == 1u) {
        // The window is one repeated byte (k_lz_sort saw it and sorted nothing).  The nearest chain
        // candidate of p is p - 1, it matches over the whole lookahead, and a match that long ends the
        // walk (deflate.zig:254-258): every position's record is (maxlen, distance 1), for both chain
        // budgets -- unless p - 1 is the chain's null (position 0, deflate.zig:248) or went with a slide.
        for (uint32_t p = tgt0 + tid; p < Mpos; p += FL_MATCH_THREADS) {
            uint32_t lov = 1u;
            if (STREAM && p >= zone) lov = FL_MAX_DIST + 1u;
            const uint32_t r = (p >= 1u && p - 1u >= lov) ? (min(N - p, FL_MAX_MATCH) << 16) : 0u;
            rec2[p] = make_uint2(r, r);
        }
        return;
    }
    // stage the window in LDS (zero padded)
    const uint32_t ndw = (min(N, WIN_DW * 4u) + 3) >> 2;
    for (uint32_t i = tid; i < WIN_DW; i += FL_MATCH_THREADS)
        win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, N) : 0u;
    __syncthreads();
    fl_prof_mark(9);

    // ---- per entry: n = min(bucket offset, chain) = how many chain candidates it may look at,
    // and qn = the position of candidate number n.  NQ[i] = n | qn << 16.
    // Wave w owns sorted indices [4096 w, 4096 (w+1)).
    {
        const uint32_t slice0 = wave * 4096u;
        // last bucket start at or before the end of this wave's slice: walk back from the end
        // (buckets are short, so this almost always ends in the first step)
        uint32_t last_start = 0;  // encoded i + 1, 0 = none
        {
            const uint32_t hi = min(slice0 + 4096u, M);
            for (uint32_t e = hi; e > slice0 && last_start == 0; e = e > 64 ? e - 64 : 0) {
                const int32_t i = (int32_t)e - 1 - (int32_t)lane;  // candidates e-1 .. e-64
                const bool valid = i >= (int32_t)slice0;
                const uint32_t h = valid ? fl_hash_le(fl_lds_load4(win32, Sc[i])) : 0;
                const uint32_t hp = (valid && i > 0) ? fl_hash_le(fl_lds_load4(win32, Sc[i - 1])) : ~0u;
                const uint32_t st = (valid && (i == 0 || h != hp)) ? (uint32_t)i + 1 : 0;
                last_start = fl_wave_max(st);
                if (e <= slice0 + 64) break;
            }
        }
        if (lane == 0) wlast[wave] = last_start;
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t w = 0; w < wave; w++) carry = max(carry, wlast[w]);
        const uint64_t le_mask = ~0ull >> (63 - lane);
        uint32_t spn[4];
        uint32_t hlast = 0xfffffffeu;  // hash of the entry just before the current round
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = slice0 + u * 64 + lane;
            spn[u] = i < M ? Sc[i] : 0;
        }
        if (slice0 > 0 && slice0 <= M) hlast = fl_hash_le(fl_lds_load4(win32, Sc[slice0 - 1]));
        for (uint32_t r = 0; r < 64; r += 4) {
            uint32_t sp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) sp[u] = spn[u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = slice0 + (r + 4 + u) * 64 + lane;
                spn[u] = (r + 4 < 64 && i < M) ? Sc[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = slice0 + (r + u) * 64 + lane;
                const bool valid = i < M;
                const uint32_t h = valid ? fl_hash_le(fl_lds_load4(win32, sp[u])) : 0xffffffffu;
                // (DPP wave_shr:1 -- lane i reads lane i - 1; lane 0 keeps `old`: the previous round's last hash)
                const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp((int)hlast, (int)h, 0x138, 0xf, 0xf, false);
                hlast = (uint32_t)__builtin_amdgcn_readlane((int)h, 63);
                // bucket start (encoded index + 1) at or before this lane: the highest start
                // flag among lanes <= lane, else the one carried in from earlier rounds
                const uint64_t starts = __ballot(valid && (i == 0 || h != hp));
                const uint64_t below = starts & le_mask;
                const uint32_t base1 = slice0 + (r + u) * 64 + 1;
                const uint32_t st = below ? base1 + 63u - (uint32_t)__builtin_clzll(below) : carry;
                if (valid) {
                    const uint32_t n = min(i + 1 - st, chain);
                    // (BF needs no position of candidate n: what follows candidate n in the tile is another
                    // bucket -- other first four bytes --, the end of the chain budget, or a slot below
                    // sorted index 0, which holds position 0 and fails the position bound)
                    const uint32_t qn = BF ? 0u : (n ? Sc[i - n] : 0xffffu);
                    NQc[i] = n | (qn << 16);
                }
                if (starts) carry = base1 + 63u - (uint32_t)__builtin_clzll(starts);
            }
        }
        __syncthreads();  // NQ is read back by other waves below
    }
    fl_prof_mark(10);

    const uint32_t nbatch = (M + 63) >> 6;
    uint16_t* ts = tS[wave];
    uint2* tw = tW[wave];

    // software pipeline: the next batch's own entry is fetched while this one is searched
    uint32_t nx_p = 0, nx_nq = 0xffff0000u, nx_tq0 = 0, nx_tq1 = 0;
    {
        const uint32_t i = (wave << 6) + lane;
        if (wave < nbatch && i < M) {
            nx_p = Sc[i];
            nx_nq = NQc[i];
        }
        const int32_t ia = (int32_t)(wave << 6) - FL_KB + (int32_t)lane, ib = ia + 64;
        nx_tq0 = (wave < nbatch && ia >= 0 && ia < (int32_t)M) ? Sc[ia] : 0;
        nx_tq1 = (wave < nbatch && lane < FL_TILE - 64 && ib >= 0 && ib < (int32_t)M) ? Sc[ib] : 0;
    }
    for (uint32_t batch = wave; batch < nbatch; batch += FL_MATCH_WAVES) {
        const uint32_t i0 = batch << 6, i = i0 + lane;
        const uint32_t p = nx_p;
        const bool active = i < M && (!STREAM || p >= tgt0);
        uint32_t n = active ? (nx_nq & 0xffff) : 0;  // candidates left to look at (loop bound only)
        // valid candidates: q >= 1 (position 0 is the chain's null, deflate.zig:248),
        // p - q <= 32768 (deflate.zig:250-251) and not beyond candidate n
        uint32_t lov = max(max(p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u, 1u), nx_nq >> 16);
        if (STREAM && p >= zone) lov = max(lov, FL_MAX_DIST + 1u);
        if (n == 0) lov = 0x7fffffffu;
        // positions of the first tile's entries (slots lane and lane + 64 < FL_TILE), fetched a batch ahead
        uint32_t tq0 = nx_tq0, tq1 = nx_tq1;
        {
            const uint32_t bn = batch + FL_MATCH_WAVES;
            const uint32_t in_ = (bn << 6) + lane;
            const bool okn = bn < nbatch && in_ < M;
            nx_p = okn ? Sc[in_] : 0;
            nx_nq = okn ? NQc[in_] : 0xffff0000u;
            const int32_t ia = (int32_t)(bn << 6) - FL_KB + (int32_t)lane, ib = ia + 64;
            nx_tq0 = (bn < nbatch && ia >= 0 && ia < (int32_t)M) ? Sc[ia] : 0;
            nx_tq1 = (bn < nbatch && lane < FL_TILE - 64 && ib >= 0 && ib < (int32_t)M) ? Sc[ib] : 0;
        }
        uint32_t p0, p1;
        fl_lds_load8(win32, p, p0, p1);
        uint32_t maxlen = min(N - p, FL_MAX_MATCH);
        if (STREAM && has_fl) maxlen = min(maxlen, fl_next_flush(fp, ck.n_flush, w0 + p, ck.in_len) - (w0 + p));
        const uint32_t cp = 0xffffu - p;  // key low half = 65535 - (p - q) = q + cp
        uint32_t key = 0, qkey = 0;
        // A candidate can beat the lane's best only if its first best+1 bytes agree with p's.
        // The first four are compared through `bad`; `mk` selects the bytes of the second
        // prefix word that must agree as well (0 while there is no match yet, all ones once
        // the best is >= 7: then the whole word must agree and the window decides).
        uint32_t mk = 0;
        uint32_t pb = 0;     // window bytes p+best-3 .. p+best (valid when best >= 8)
        uint32_t dmask = 0;  // candidates of the current tile whose 8 prefix bytes all agree
        bool qsnap = false;
        // Window work is deferred to the end of the tile: there every lane with such a candidate
        // extends one per step, instead of the whole wave stalling whenever a single lane has
        // one.  The key is a max, so the order of evaluation inside a tile does not matter,
        // except for `nice` (deflate.zig:256-258), which only cuts off candidates after it.
        auto flush_deep = [&]() {
            while (__any(dmask != 0)) {
                if (dmask) {
                    const uint32_t kk = (uint32_t)__builtin_ctz(dmask) + 1;  // nearest first
                    dmask &= dmask - 1;
                    const uint32_t q = ts[FL_KB + lane - kk];
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
                            mk = ~0u;
                            pb = fl_lds_load4(win32, p + le - 3);
                            if (le >= maxlen || le >= nice) {  // nothing longer possible / stop looking
                                n = 0;
                                lov = 0x7fffffffu;
                                dmask = 0;
                            }
                        }
                    }
                }
            }
        };
        // ---- BF ----
        const uint32_t lenmask = maxlen >= 8 ? 0x80808080u : (0x00808080u >> (8 * (7 - maxlen)));
        uint32_t bb = 0;     // best score of the current tile: equal-byte flags | 0x40 | 32 - candidate number
        uint32_t kdone = 0;  // candidates of the current tile scored so far; dmask bit b = candidate kdone - b
        // Runs and repeated records (every candidate agrees in 8 bytes, nearly every lane of the wave already
        // holds a match of >= 8): such a group of four candidates is filtered instead of scored -- a
        // candidate can only beat a match of `best` bytes if its byte number `best` agrees
        // (SlidingWindow.zig:91-93), one byte load per candidate.  What passes goes into dmask like an
        // 8-byte candidate, but its prefix has not been looked at: fcand remembers which candidates of the
        // tile were filtered, and those candidates are compared from their first byte.
        bool runny = false;    // the wave has been served early once in this batch (uniform)
        uint32_t fcand = 0;    // bit t - 1: candidate t of the current tile was filtered, not scored (uniform)
        auto flush_deep_rj = [&]() {
            // an 8-byte candidate of a lane that can match at most 8 bytes has nothing to add to its score
            // (a filtered one has no score yet: bit b of dmask = candidate kdone - b)
            if (maxlen <= 8) dmask &= (fcand && kdone) ? (__brev(fcand) >> (32u - kdone)) : 0u;
            for (;;) {
                const uint64_t act = __ballot(dmask != 0);
                if (!act) break;
                if (__popcll(act) <= 2 && __any(__popc(dmask) >= 3)) {
                    // One or two lanes left, with several candidates (the start of a run walking back through
                    // the end of the previous one: every candidate a byte longer than the last): the whole wave
                    // compares one pair at a time, lane l the bytes 4 l .. 4 l + 3 -- the common prefix
                    // in one step however long it is, instead of 8 bytes per trip on one lane.
                    for (uint64_t rem = act; rem; rem &= rem - 1) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(rem);
                        const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)l);
                        const uint32_t ML = (uint32_t)__builtin_amdgcn_readlane((int)maxlen, (int)l);
                        const uint32_t off = 4u * lane;
                        const uint32_t pw = off < ML ? fl_lds_load4(win32, P + off) : 0u;  // p's bytes: once per lane served
                        for (;;) {  // all of this lane's candidates before the next lane is served
                            uint32_t q_l = ~0u;
                            if (lane == l) {
                                if (maxlen <= (key >> 16)) dmask = 0;  // nothing can beat the match it holds
                                if (dmask) {
                                    const uint32_t b = 31u - (uint32_t)__builtin_clz(dmask);  // nearest first
                                    dmask &= ~(1u << b);
                                    q_l = ts[FL_KB + lane - (kdone - b)];
                                }
                            }
                            const uint32_t Q = (uint32_t)__builtin_amdgcn_readlane((int)q_l, (int)l);
                            if (Q == ~0u) break;
                            uint32_t x = 0;
                            if (off < ML) x = pw ^ fl_lds_load4(win32, Q + off);
                            const uint64_t diff = __ballot(x != 0);
                            uint32_t lcp;
                            if (diff) {
                                const uint32_t fl_ = (uint32_t)__builtin_ctzll(diff);
                                const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)fl_);
                                lcp = 4u * fl_ + ((uint32_t)__builtin_ctz(xf) >> 3);
                            } else if (ML > 256u) {  // bytes 256, 257 (FL_MAX_MATCH = 258)
                                const uint32_t xt = fl_lds_load4(win32, P + 256u) ^ fl_lds_load4(win32, Q + 256u);
                                lcp = 256u + (xt ? ((uint32_t)__builtin_ctz(xt) >> 3) : 4u);
                            } else {
                                lcp = ML;
                            }
                            lcp = min(lcp, ML);
                            if (lane == l) {
                                const uint32_t kc = (lcp << 16) | (q_l + cp);
                                if (lcp >= FL_MIN_MATCH && kc > key) {  // deflate.zig:254-261
                                    key = kc;
                                    if (lcp >= maxlen || lcp >= nice) {  // nothing longer possible / stop looking
                                        n = 0;
                                        lov = 0x7fffffffu;
                                        dmask = 0;
                                    }
                                }
                            }
                        }
                        if (lane == l && (key >> 16) >= 8u) pb = fl_lds_load4(win32, p + (key >> 16) - 3);
                    }
                    continue;
                }
                if (dmask) {
                    const uint32_t b = 31u - (uint32_t)__builtin_clz(dmask);  // nearest first
                    dmask &= ~(1u << b);
                    const uint32_t t = kdone - b;
                    const uint32_t q = ts[FL_KB + lane - t];
                    const uint32_t best = key >> 16;
                    // SlidingWindow.zig:91-98: a candidate that does not extend the best match is
                    // dropped on one compare
                    bool take = maxlen > best;
                    if (take && best >= 8) take = fl_lds_load4(win32, q + best - 3) == pb;
                    // (a filtered candidate's prefix has not been looked at: compare from the first byte;
                    // fewer than 4 equal bytes -- another 4-gram of the bucket, or another bucket -- is no match)
                    const bool filtered = (fcand >> (t - 1u)) & 1u;
                    if (take) {
                        const uint32_t le = fl_extend_len(win32, p, q, filtered ? 0u : 8u, maxlen);
                        const uint32_t kc = (le << 16) | (q + cp);
                        if (le >= FL_MIN_MATCH && kc > key) {  // deflate.zig:254-261
                            key = kc;
                            pb = fl_lds_load4(win32, p + le - 3);
                            if (le >= maxlen || le >= nice) {  // nothing longer possible / stop looking
                                n = 0;
                                lov = 0x7fffffffu;
                                dmask = 0;
                            }
                        }
                    }
                }
            }
        };
        auto flush_deep_plain = [&]() {
            if (maxlen <= 8) dmask = 0;
            while (__any(dmask != 0)) {
                if (dmask) {
                    const uint32_t b = 31u - (uint32_t)__builtin_clz(dmask);  // nearest first
                    dmask &= ~(1u << b);
                    const uint32_t q = ts[FL_KB + lane - (kdone - b)];
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
                                lov = 0x7fffffffu;
                                dmask = 0;
                            }
                        }
                    }
                }
            }
        };
        auto flush_deep_bf = [&]() {
            if (RJ)
                flush_deep_rj();
            else
                flush_deep_plain();
        };
        // the tile's winner meets the key, then the 8-byte candidates meet the window
        auto tile_end_bf = [&]() {
            if (__any(bb != 0)) {
                if (bb) {
                    const uint32_t q = ts[FL_KB + lane - (32u - (bb & 31u))];
                    const uint32_t le = min(4u + (uint32_t)__popc(bb & 0x80808080u), maxlen);
                    const uint32_t kc = (le << 16) | (q + cp);
                    if (kc > key) {  // deflate.zig:254-261
                        key = kc;
                        if (le >= 8) pb = fl_lds_load4(win32, p + le - 3);
                        if (le >= maxlen) {  // nothing longer possible (le <= 8 < nice here)
                            n = 0;
                            lov = 0x7fffffffu;
                            dmask = 0;
                        }
                    }
                }
                bb = 0;
            }
            flush_deep_bf();
        };
        for (uint32_t kb = 0; kb < chain; kb += FL_KB) {
            if (!__any(n > kb)) break;
            kdone = 0;
            fcand = 0;
            // tile = sorted entries [i0 - kb - FL_KB, i0 - kb + 64) with their first 8 bytes
            fl_lds_order();
            {
                uint32_t a0, a1;
                fl_lds_load8(win32, tq0, a0, a1);
                ts[lane] = (uint16_t)tq0;
                tw[lane] = make_uint2(a0, a1);
                if (lane < FL_TILE - 64) {
                    fl_lds_load8(win32, tq1, a0, a1);
                    ts[lane + 64] = (uint16_t)tq1;
                    tw[lane + 64] = make_uint2(a0, a1);
                }
            }
            if (__any(n > kb + FL_KB)) {  // the following tile's positions fly during this tile's search
                const int32_t ia = (int32_t)i0 - (int32_t)kb - 2 * FL_KB + (int32_t)lane, ib = ia + 64;
                tq0 = (ia >= 0 && ia < (int32_t)M) ? Sc[ia] : 0;
                tq1 = (lane < FL_TILE - 64 && ib >= 0 && ib < (int32_t)M) ? Sc[ib] : 0;
            }
            fl_lds_order();
            // candidate kk of this tile sits in slot FL_KB + lane - kk; walk the slots downwards
            const uint16_t* tsp = ts + FL_KB + lane;
            const uint2* twp = tw + FL_KB + lane;
#pragma unroll
            for (uint32_t kk0 = 1; kk0 <= FL_KB; kk0 += 4) {
                // the chain >> 2 budget (deflate.zig:241-245) ends after candidate `quarter`
                // (a multiple of 4 at every level, deflate.zig:44-49)
                if (kb + kk0 - 1 == quarter) {
                    if (BF)
                        tile_end_bf();
                    else
                        flush_deep();
                    qkey = key;
                    qsnap = true;
                }
                if (!__any(n >= kb + kk0)) break;
                tsp -= 4;
                twp -= 4;
                if (RJ && BF && runny &&
                    __popcll(__ballot(n >= kb + kk0 && key < (8u << 16))) <= 8) {
                    // (nearly) every lane that still walks holds a match of >= 8 bytes: filter (see above; the
                    // few lanes with a shorter match filter on their byte number `best` just the same)
                    const uint32_t bo = min(key >> 16, maxlen - 1u);
                    const uint8_t* win8 = (const uint8_t*)win32;
                    const uint32_t pbyte = win8[p + bo];
                    fcand |= 0xfu << (kk0 - 1u);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t q = tsp[3 - u];
                        // (a lane whose walk has ended has lov = 0x7fffffff)
                        const uint32_t s = (q >= lov && win8[q + bo] == pbyte) ? 0x80000000u : 0u;
                        dmask = __builtin_amdgcn_alignbit(dmask, s, 31);
                    }
                    kdone = kk0 + 3;
                    if ((kk0 & 7u) == 5u && __popcll(__ballot(dmask != 0)) >= 40) tile_end_bf();
                    continue;
                }
                if (BF) {
                    // (the compiler turns the conditions into branches that skip the remaining loads of a
                    // candidate whose first four bytes differ; measured faster than forcing them straight)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t q = tsp[3 - u];
                        const uint2 w = twp[3 - u];
                        const uint32_t x0 = w.x ^ p0, x1 = w.y ^ p1;
                        const uint32_t m = ~x1 & (x1 - 1u);  // ones below the lowest differing bit
                        const uint32_t sc = (m & lenmask) | (0x40u | (32u - (kk0 + u)));
                        // (a lane whose walk has ended has lov = 0x7fffffff)
                        const uint32_t s = (x0 == 0 && q >= lov) ? sc : 0u;
                        bb = max(bb, s);
                        dmask = __builtin_amdgcn_alignbit(dmask, s, 31);
                    }
                    kdone = kk0 + 3;
                    // When most lanes are waiting for the window anyway (runs, long repeats), one round
                    // serves them all: do it now; a match of `nice` bytes then ends the walk early.
                    if (RJ && (kk0 & 7u) == 5u && __popcll(__ballot(dmask != 0)) >= 40) {  // (every 8 candidates)
                        tile_end_bf();
                        if (RJ) runny = true;
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t q = tsp[3 - u];
                    const uint2 w = twp[3 - u];
                    const uint32_t x = w.y ^ p1;
                    // bad != 0: first four bytes differ (a colliding 4-gram or another bucket),
                    // or the candidate is below the lane's lower bound
                    const uint32_t bad = (w.x ^ p0) | ((q - lov) >> 31);
                    if (((x & mk) | bad) == 0) {
                        // longer than the best so far (deflate.zig:254) as far as the 8 prefix bytes tell
                        uint32_t tz;
                        asm("v_ffbl_b32 %0, %1" : "=v"(tz) : "v"(x));  // ~0 for x == 0
                        const uint32_t le = min(min(4u + (tz >> 3), 8u), maxlen);
                        const uint32_t kc = (le << 16) | (q + cp);
                        if (kc > key) {
                            key = kc;
                            if (le >= 8) pb = fl_lds_load4(win32, p + le - 3);
                            if (le >= maxlen) {  // nothing longer possible (le <= 8 < nice here)
                                n = 0;
                                lov = 0x7fffffffu;
                                dmask = 0;
                            }
                            mk = le >= 7 ? ~0u : ((1u << (8 * (le - 3))) - 1u);
                        }
                        // all 8 prefix bytes agree and the match may go on: the window decides, later
                        if (x == 0 && maxlen > 8) dmask |= 1u << (kk0 + u - 1);
                    }
                }
                // When most lanes are waiting for the window anyway (runs, long repeats), one round
                // serves them all: do it now; a match of `nice` bytes then ends the walk early.
                if (__popcll(__ballot(dmask != 0)) >= 40) flush_deep();
            }
            if (BF)
                tile_end_bf();
            else
                flush_deep();
        }
        if (!qsnap) qkey = key;
        if (active) {
            // key -> record: len << 16 | dist - 1, dist = 65535 - low half
            const uint32_t rf = (key >> 16) ? ((key & 0xffff0000u) | (0xfffeu - (key & 0xffffu))) : 0u;
            const uint32_t rq = (qkey >> 16) ? ((qkey & 0xffff0000u) | (0xfffeu - (qkey & 0xffffu))) : 0u;
            rec2[p] = make_uint2(rf, rq);
        }
    }
    fl_prof_mark(11);
}

// ------------------------------------------------------------------ anchor descriptors
// desc[p]: what an anchor at p emits.  0 = one literal, next anchor p + 1.
// Otherwise bit31 | j << 23 | (len - 3) << 15 | dist - 1: j literals p .. p+j-1, then a
// match (len, dist) at p + j; next anchor p + j + len.
#define FL_PARSE_THREADS 1024

__device__ __forceinline__ uint32_t fl_desc_next(uint32_t d, uint32_t p) {
    if (!d) return p + 1;
    return p + ((d >> 23) & 0xff) + ((d >> 15) & 0xff) + 3;
}

// deflate.zig:154-194 seen from a position visited with no pending match.  ra / rb are the
// records of p and p + 1.
__device__ __forceinline__ uint32_t fl_anchor_desc(const uint2* __restrict__ rec2, uint32_t p, uint2 ra, uint2 rb,
                                                   uint32_t good, uint32_t lazy) {
    const uint32_t r = ra.x;  // findMatch(pos, lh, 0): full budget
    if (!r) return 0;
    uint32_t len = r >> 16, dist0 = r & 0x7fff, j = 0, q = p;
    while (len < lazy) {  // deflate.zig:171-178: keep the match, look one position further
        const uint32_t sel = len >= good ? 1u : 0u;  // deflate.zig:242-245
        uint32_t r2;
        if (j == 0) {
            r2 = sel ? rb.y : rb.x;
        } else {
            const uint2 rr = rec2[q + 1];
            r2 = sel ? rr.y : rr.x;
        }
        const uint32_t l2 = r2 >> 16;
        if (l2 <= len) break;  // deflate.zig:182-184: no better match, the pending one goes out
        len = l2;              // deflate.zig:166-168: better match, the pending one becomes a literal
        dist0 = r2 & 0x7fff;
        j++;
        q++;
    }
    return 0x80000000u | (j << 23) | ((len - 3) << 15) | dist0;
}

// ------------------------------------------------------------------ token emission helpers
// (k_lz_emit in kernels_parse.h; k_st_emit in kernels_stream.h)
#define FL_EMIT_WAVES 16
#define FL_EMIT_THREADS (64 * FL_EMIT_WAVES)

__device__ __forceinline__ uint32_t fl_win_byte(const uint32_t* win32, uint32_t off) {
    return (win32[off >> 2] >> (8 * (off & 3))) & 0xff;
}

// ------------------------------------------------------------------ parts of a chunk in the emitters
// k_lz_emit (kernels_parse.h) handles a chunk in parts of 8192 positions; wave w owns positions
// [h0 + 512 w, h0 + 512 (w + 1)) of a part.  (k_lz_tok -- parse and emit of the round-2 chunk path in one kernel over the
// records of k_lz_match -- went with round 4: levels 8 and 9 take kernels_walk.h.)
#ifndef FL_TOK_PART
#define FL_TOK_PART 8192u
#endif
#define FL_TOK_SPAN (FL_TOK_PART / 16u)  // positions per wave per part
#define FL_TOK_R (FL_TOK_SPAN / 64u)     // positions per lane per part
#define FL_TOK_LOOK 256u                 // literals of an anchor may reach this far past its part (j < 256)
#define FL_TOK_WIN_DW ((FL_TOK_PART + FL_TOK_LOOK) / 4u + 2u)

