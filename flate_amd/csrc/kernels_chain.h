// kernels_chain.h -- the match finder of the chunk path (levels 4..9, inputs of at most 65535
// bytes), third generation: the reference's own data structure, hash chains, walked by lanes.
//
// Reference path: Lookup.add / Lookup.prev (Lookup.zig:23-51), Deflate.findMatch
// (deflate.zig:233-266), SlidingWindow.match (SlidingWindow.zig:81-104).
//
//   k_lz_chain  builds the chains: delta[p] = p - (nearest earlier position with the same hash),
//               0xffff when there is none (Lookup.zig:43-51: head[h] = 0 is the null, so position 0
//               never is a candidate).  The chains do not depend on the parse (every position is
//               inserted exactly once, in ascending order: deflate.zig:207-211, 236), so one wave
//               per chunk inserts 64 consecutive positions per step into a 32768-entry head table
//               in LDS.  Nothing waits for a result: the LDS unit runs one wave's DS instructions
//               in order, so "read head, write head" of consecutive steps pipeline.
//   k_lz_walk   for EVERY position the record findMatch would return, for the full chain budget
//               and for chain >> 2 (deflate.zig:241-245).  Lane = one position, walking its chain
//               exactly as the reference does: one byte compare at offset `best` rejects a
//               candidate (SlidingWindow.zig:91-98), the chain hop is one 16-bit LDS read.  Lanes
//               that finish take the next position of their wave, so the lanes stay busy whatever
//               the chain lengths are; positions are handed out in ascending order, which keeps
//               the record stores of a wave within a few hundred bytes of each other.
//
// Against the second generation (sort by hash, walk sorted tiles): no sort, no bucket-offset
// pre-pass, no position-scattered 8-byte record stores (40 GB written per GiB), lane efficiency
// independent of the bucket-size mix.  Both kernels are bound by vector-ALU issue and LDS latency;
// no MFMA (byte compares and pointer hops).
#pragma once
#include "kernels_common.h"
#include "kernels_lz.h"

#define FL_NO_PREV 0xffffu

// counters of wave 0 of workgroup 0 (tuning aid, compiled in with -DFL_WALK_PROF)
#ifdef FL_WALK_PROF
#define WK_CNT(slot, v)                                                             \
    do {                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] += (uint64_t)(v); \
    } while (0)
#else
#define WK_CNT(slot, v)
#endif

// ------------------------------------------------------------------ k_lz_chain
// One wave per chunk; LDS = the head table (64 KiB) + a 2 x 1 KiB staging buffer for the input.
#define FL_CHAIN_STG_DW 264  // 1024 bytes + 16 (alignment shift) + 4 (hash of the last position) rounded up

__global__ __launch_bounds__(64) void k_lz_chain(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks,
                                                 uint16_t* __restrict__ delta_all) {
    __shared__ uint16_t head[32768];
    __shared__ uint32_t stg[2][FL_CHAIN_STG_DW];
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;  // positions with 4 bytes left (Lookup.zig:24)
    if (Mpos == 0) return;
    const uint8_t* src = in + ck.in_off;
    uint16_t* dl = delta_all + (uint64_t)c * FL_CHUNK_STRIDE;
    const uint32_t sh = (uint32_t)((uintptr_t)src & 15);
    const uint4* src16 = (const uint4*)(src - sh);  // 16-byte granules; granule g covers chunk bytes 16 g - sh ..
    const uint32_t n_gran = (N + sh + 15) >> 4;     // granules holding at least one byte of the chunk
    {
        uint4* h4 = (uint4*)head;
        for (uint32_t i = lane; i < 4096; i += 64) h4[i] = make_uint4(0, 0, 0, 0);
    }
    // block b = chunk bytes [1024 b, 1024 b + 1024) plus what its last position needs: granules
    // 64 b .. 64 b + 65 (sh + 1023 + 3 < 1056 = 66 granules); lane l loads granule 64 b + l, lanes 0..1 two more
    auto load_block = [&](uint32_t b, uint4& g0, uint4& g1) {
        const uint32_t ga = 64 * b + lane, gb = 64 * b + 64 + lane;
        g0 = ga < n_gran ? src16[ga] : make_uint4(0, 0, 0, 0);
        g1 = (lane < 2 && gb < n_gran) ? src16[gb] : make_uint4(0, 0, 0, 0);
    };
    const uint32_t n_blocks = (Mpos + 1023) >> 10;
    uint4 ga0, ga1, gb0, gb1;  // two blocks in flight
    load_block(0, ga0, ga1);
    if (n_blocks > 1) load_block(1, gb0, gb1);
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint32_t* sb = stg[b & 1];
        ((uint4*)sb)[lane] = ga0;
        if (lane < 2) ((uint4*)sb)[64 + lane] = ga1;
        ga0 = gb0;
        ga1 = gb1;
        if (b + 2 < n_blocks) load_block(b + 2, gb0, gb1);
        fl_lds_order();
#pragma unroll 4
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t p = (b << 10) + (s << 6) + lane;
            const bool valid = p < Mpos;
            const uint32_t off = (s << 6) + lane + sh;
            const uint32_t v = __builtin_amdgcn_alignbyte(sb[(off >> 2) + 1], sb[off >> 2], off & 3);
            const uint32_t h = fl_hash_le(v);
            uint32_t old = 0, chk = p;
            if (valid) old = head[h];
            fl_lds_order();
            if (valid) head[h] = (uint16_t)p;
            fl_lds_order();
            if (valid) chk = head[h];
            // lanes of this step that share a hash: whichever store won, the others see it
            uint64_t dup = __ballot(chk != p);
            while (dup) {
                const uint32_t l0 = (uint32_t)__builtin_ctzll(dup);
                const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)l0);
                const uint64_t grp = __ballot(valid && h == hk);  // ascending lanes = ascending positions
                const uint64_t below = grp & ((1ull << lane) - 1ull);
                if (valid && h == hk) {
                    if (below) old = (b << 10) + (s << 6) + 63u - (uint32_t)__builtin_clzll(below);
                    if ((grp >> lane) == 1ull) head[h] = (uint16_t)p;  // the last one stays in the table
                }
                dup &= ~grp;
            }
            fl_lds_order();
            if (valid) dl[p] = old ? (uint16_t)(p - old) : (uint16_t)FL_NO_PREV;
        }
        fl_lds_order();
    }
}

// ------------------------------------------------------------------ k_lz_walk
// One workgroup (16 waves) per chunk, one workgroup per CU: the whole window (64 KiB + the
// zero padding a 258-byte compare may touch) and a ring of 40960 chain entries stay in LDS.
#define FL_WALK_WAVES 16
#define FL_WALK_THREADS (64 * FL_WALK_WAVES)
#define FL_WALK_WIN_DW (16384 + 72)
#define FL_WALK_BLK 4096u                // ring refill unit (positions)
#define FL_WALK_RING (10u * FL_WALK_BLK)  // entries: 32768 of history + two blocks of positions in flight
#define FL_WALK_REFILL 20                // lanes without work before the wave hands out new positions
#define FL_WALK_VERIFY 16                // lanes waiting for a full compare before the wave serves them

// ring slot of position x (x < 65536 < 2 * FL_WALK_RING)
__device__ __forceinline__ uint32_t fl_ring_slot(uint32_t x) { return min(x, x - FL_WALK_RING); }

__global__ __launch_bounds__(FL_WALK_THREADS, 1) void k_lz_walk(const uint8_t* __restrict__ in,
                                                                const fl_chunk* __restrict__ chunks, fl_params prm,
                                                                const uint16_t* __restrict__ delta_all,
                                                                uint32_t* __restrict__ rec_all) {
    __shared__ uint32_t win32[FL_WALK_WIN_DW];
    __shared__ uint16_t ring[FL_WALK_RING];
    __shared__ uint32_t ctl_loaded;  // blocks of the chain array copied into the ring so far
    __shared__ uint32_t ctl_lock;
    __shared__ uint32_t ctl_done[16];  // finished positions per block
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;
    const uint8_t* src = in + ck.in_off;
    const uint16_t* dl = delta_all + (uint64_t)c * FL_CHUNK_STRIDE;
    uint2* rec2 = (uint2*)rec_all + ck.pos_off;
    const uint8_t* win8 = (const uint8_t*)win32;
    const uint32_t chain = prm.chain, nice = prm.nice;
    const uint32_t qmark = chain - (chain >> 2);  // cnt after `chain >> 2` candidates (deflate.zig:241-245)
    const uint32_t n_blk = (Mpos + FL_WALK_BLK - 1) / FL_WALK_BLK;

    fl_prof_mark(8);
    {
        const uint32_t ndw = (N + 3) >> 2;
        for (uint32_t i = tid; i < FL_WALK_WIN_DW; i += FL_WALK_THREADS)
            win32[i] = i < ndw ? fl_load_u32_clamped(src, 4 * i, N) : 0u;
        // positions without a hash entry never match (Lookup.zig:24)
        for (uint32_t p = Mpos + tid; p < N; p += FL_WALK_THREADS) rec2[p] = make_uint2(0u, 0u);
        const uint32_t first = min(n_blk, FL_WALK_RING / FL_WALK_BLK) * FL_WALK_BLK;  // multiple of 8 entries
        const uint4* d4 = (const uint4*)dl;
        for (uint32_t i = tid; i < first / 8; i += FL_WALK_THREADS) ((uint4*)ring)[i] = d4[i];
        if (tid < 16) ctl_done[tid] = 0;
        if (tid == 0) {
            ctl_loaded = first / FL_WALK_BLK;
            ctl_lock = 0;
        }
    }
    __syncthreads();
    fl_prof_mark(9);
    if (Mpos == 0) return;

    // this wave's positions: strips of 64, strip numbers wave, wave + 16, ...
    const uint32_t n_strips = (Mpos + 63) >> 6;
    const uint32_t total = n_strips > wave ? ((n_strips - wave + FL_WALK_WAVES - 1) / FL_WALK_WAVES) << 6 : 0u;
    uint32_t cur = 0;  // sequence numbers handed out so far
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

    // lane state
    enum { ST_IDLE = 0, ST_WALK = 1, ST_VERIFY = 2, ST_DONE = 3 };
    uint32_t st = ST_IDLE;
    uint32_t p = 0, q = 0, lov = 0, cnt = 0, best = 0, key = 0, qkey = 0, pbyte = 0, maxlen = 0, p0 = 0, p1 = 0, dsave = 0;

    // after candidate q has been looked at: the chain >> 2 snapshot, then the hop (deflate.zig:248-263)
    auto advance = [&](uint32_t d) {
        cnt--;
        if (cnt == qmark) qkey = key;
        const int32_t nq = (int32_t)q - (int32_t)d;  // FL_NO_PREV makes it negative
        st = (nq >= (int32_t)lov && cnt != 0) ? ST_WALK : ST_DONE;
        q = (uint32_t)nq;
    };

    for (;;) {
        uint64_t m_walk = __ballot(st == ST_WALK);
        uint64_t m_ver = __ballot(st == ST_VERIFY);
        const uint64_t m_free = ~(m_walk | m_ver);  // idle or done
        // ---- finished lanes store their records; free lanes take the wave's next positions
        const bool feed = cur < total;
        WK_CNT(32, 1);
        WK_CNT(35, __popcll(m_walk));
        if ((m_walk | m_ver) == 0 || __popcll(m_free) >= (feed ? FL_WALK_REFILL : 48)) {
            uint64_t m_done = __ballot(st == ST_DONE);
            WK_CNT(33, 1);
            WK_CNT(36, __popcll(m_free));
            if (st == ST_DONE) {
                if (cnt > qmark) qkey = key;  // the walk ended inside the chain >> 2 budget
                rec2[p] = make_uint2(key, qkey);
                st = ST_IDLE;
            }
            const uint32_t pblk = p >> 12;
            while (m_done) {  // count them per block (a wave's positions span one or two blocks)
                const uint32_t l0 = (uint32_t)__builtin_ctzll(m_done);
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)pblk, (int)l0);
                const uint64_t g = m_done & __ballot(pblk == b0);
                if (lane == l0) atomicAdd(&ctl_done[b0], (uint32_t)__popcll(g));
                m_done &= ~g;
            }
            if (feed) {
                uint32_t loaded = __hip_atomic_load(&ctl_loaded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                loaded = (uint32_t)__builtin_amdgcn_readfirstlane((int)loaded);
                // the ring lacks the block of the next position: copy it in once every position that
                // still needs the block it replaces is finished (all blocks <= loaded - 2)
                const uint32_t first_np = ((wave + FL_WALK_WAVES * (cur >> 6)) << 6) + (cur & 63);
                if ((first_np >> 12) >= loaded && loaded < n_blk) {
                    uint32_t dn = FL_WALK_BLK;
                    if (lane + 2 <= loaded && lane < 16)
                        dn = __hip_atomic_load(&ctl_done[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__ballot(dn != FL_WALK_BLK) == 0) {
                        uint32_t got = 1;
                        if (lane == 0) got = atomicCAS(&ctl_lock, 0u, 1u);
                        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                        if (got == 0) {
                            uint32_t l2 = __hip_atomic_load(&ctl_loaded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            l2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l2);
                            if (l2 == loaded) {
                                const uint4* d4 = (const uint4*)(dl + (uint64_t)loaded * FL_WALK_BLK);
                                uint4* r4 = (uint4*)(ring + (loaded - FL_WALK_RING / FL_WALK_BLK) * FL_WALK_BLK);
#pragma unroll
                                for (uint32_t k = 0; k < FL_WALK_BLK / 8 / 64; k++) r4[k * 64 + lane] = d4[k * 64 + lane];
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                if (lane == 0)
                                    __hip_atomic_store(&ctl_loaded, loaded + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) __hip_atomic_store(&ctl_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            loaded = __hip_atomic_load(&ctl_loaded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            loaded = (uint32_t)__builtin_amdgcn_readfirstlane((int)loaded);
                        }
                    } else if ((m_walk | m_ver) == 0) {
                        __builtin_amdgcn_s_sleep(8);
                        WK_CNT(38, 1);
                    }
                }
                const uint32_t r = (uint32_t)__popcll(m_free & lt_mask);
                const uint32_t i = cur + r;
                const uint32_t np = ((wave + FL_WALK_WAVES * (i >> 6)) << 6) + (i & 63);
                const bool mine = ((m_free >> lane) & 1) && i < total && (np >> 12) < loaded;
                cur += (uint32_t)__popcll(__ballot(mine));
                if (mine && np < Mpos) {
                    p = np;
                    const uint32_t d = ring[fl_ring_slot(p)];
                    lov = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;  // deflate.zig:248-251: not the null, not too far
                    fl_lds_load8(win32, p, p0, p1);
                    maxlen = min(N - p, (uint32_t)FL_MAX_MATCH);
                    best = 0;
                    key = 0;
                    qkey = 0;
                    cnt = chain;
                    pbyte = p0 & 0xffu;
                    const int32_t nq = (int32_t)p - (int32_t)d;
                    q = (uint32_t)nq;
                    st = nq >= (int32_t)lov ? ST_WALK : ST_DONE;
                }
            } else if (__ballot(st != ST_IDLE) == 0) {
                break;
            }
            m_walk = __ballot(st == ST_WALK);
        }
        // ---- full compares for the lanes whose candidate passed the one-byte test
        if (__popcll(m_ver) >= FL_WALK_VERIFY || (m_ver && !m_walk)) {
            WK_CNT(34, 1);
            WK_CNT(37, __popcll(m_ver));
            if (st == ST_VERIFY) {
                uint32_t a0, a1;
                fl_lds_load8(win32, q, a0, a1);
                const uint32_t x0 = a0 ^ p0, x1 = a1 ^ p1;
                uint32_t le;
                if (x0)
                    le = (uint32_t)__builtin_ctz(x0) >> 3;
                else if (x1)
                    le = 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
                else
                    le = maxlen > 8 ? fl_extend_match(win32, p, q, maxlen) : 8;
                le = min(le, maxlen);
                st = ST_WALK;
                if (le > best && le >= FL_MIN_MATCH) {  // deflate.zig:254-261
                    best = le;
                    key = (le << 16) | (p - q - 1);
                    if (le >= nice || le >= maxlen)
                        st = ST_DONE;  // good enough / nothing longer possible
                    else
                        pbyte = win8[p + le];
                }
                if (st != ST_DONE) advance(dsave);
            }
        }
        // ---- one chain step for the walking lanes (SlidingWindow.zig:91-98, Lookup.zig:37-39)
        if (st == ST_WALK) {
            const uint32_t cb = win8[q + best];
            const uint32_t d = ring[fl_ring_slot(q)];
            if (cb == pbyte) {
                st = ST_VERIFY;
                dsave = d;
            } else {
                advance(d);
            }
        }
    }
    fl_prof_mark(11);
}
