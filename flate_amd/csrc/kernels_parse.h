// kernels_parse.h -- the LZ77 tokenizer of the chunk path (levels 4..9, inputs of at most 65535 bytes),
// round 3: demand driven.  The reference calls findMatch only where its lazy-matching automaton
// actually goes (0.42 calls per byte of text, deflate.zig:154-205) and rejects most candidates on
// one compare (SlidingWindow.zig:91-98); the round-2 kernels computed two records for EVERY position
// (2 calls per byte, no reject).  Here the automaton itself runs on the GPU:
//
//   k_lz_chain  hash chains (Lookup.zig:23-51): prev[p] = nearest earlier position with the same
//               hash, 0 = none.  They do not depend on the parse (every position is inserted exactly
//               once, in ascending order: deflate.zig:207-211, 236), so one wave per chunk inserts
//               64 consecutive positions per step into a head table in LDS.
//   k_lz_parse  one workgroup per chunk, one LANE per 64-byte (or 32-byte) segment of the input.
//               Phase 0: every lane runs the reference's automaton from the start of its segment as
//               if that were a position visited with no pending match ("anchor") -- findMatch walks
//               the chain in LDS exactly as deflate.zig:233-266 does, candidate after candidate, with
//               the reference's early reject.  A parse started at an arbitrary position falls in step
//               with the true parse within a few bytes (measured: 6 bytes on average, never more than
//               128 on the benchmark text), because both are at an anchor whenever a token of each
//               ends at the same position.  Stitch rounds: the true path is followed from segment to
//               segment by pointer jumping over the lanes' exits; a segment the path enters at a
//               position its lane did not visit is parsed again from there until it meets the lane's
//               own anchors (or leaves the segment).  Rounds repeat until every segment on the path
//               has been resolved for the entry it really gets -- at that point nothing is
//               speculative any more: the marked anchors are exactly the reference's.
//               Output: one descriptor per anchor (what it emits) and the bitmap of true anchors.
//   k_lz_emit   tokens, per-block histograms and the 32768-token block cut (deflate.zig:213-230,
//               268-288; block_writer.zig:444-462) by prefix sums over the anchors.
//
// LDS holds, per sub-pass, the window bytes and the chain links of every position a call of the
// sub-pass can touch: targets [0, 49152) need positions [0, 49152 + 256 + 266), targets [49152, 65536)
// need [16320, 65536) -- 3 bytes per position, 145 KiB, one workgroup per CU.
//
// Bounds: k_lz_parse is bound by vector-ALU issue and LDS latency (pointer chasing), k_lz_chain by
// LDS latency of one wave; no MFMA (byte compares and pointer hops).
#pragma once
#include "kernels_common.h"
#include "kernels_lz.h"

// ------------------------------------------------------------------ k_lz_chain
// LDS = the head table (32768 x 16 bit = 64 KiB) + a 1 KiB staging buffer per wave.
// "Exchange head[h] with p" is one LDS instruction per 64 positions: DS_MSKOR_RTN_B32 (D = (D & ~mask) | value,
// returns the old word) on the 32-bit word that holds two 16-bit heads.  Lanes that hit the same word in
// one instruction are served in lane order and one wave's DS instructions in program order (measured:
// tools/ubench/ub2.hip, profiles/r03_ubench.txt), which is the order of the reference's insertions, so the
// value a lane gets back IS its chain link -- also for positions of one step that share a hash.  The
// kernel does not rely on it: a lane that was overtaken receives a position above its own, and any such
// lane makes the wave redo the chunk one position at a time.
#define FL_CHAIN_STG_DW 264  // 1024 bytes + 16 (alignment shift) + 4 (hash of the last position) rounded up

typedef __attribute__((address_space(3))) uint32_t fl_lds_u32;
__device__ __forceinline__ uint32_t fl_lds_mskor_rtn(uint32_t* lds_word, uint32_t mask, uint32_t value) {
    uint32_t old;
    fl_lds_u32* a = (fl_lds_u32*)lds_word;  // (the 32-bit LDS address)
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=v"(old) : "v"(a), "v"(mask), "v"(value) : "memory");
    return old;
}

// Levels whose chain budget reaches this value keep the match finder of kernels_lz.h (flate_hip.hip).
#ifndef FL_BULK_MIN_CHAIN
#define FL_BULK_MIN_CHAIN 1024u
#endif

// FL_CHAIN_WAVES (8) waves per chunk.  Wave w prepares block 8 k + w (1024 positions: loads, staging, hashes) while the others
// prepare theirs; the exchanges themselves are issued block after block -- wave 0, barrier, wave 1, barrier, ... --
// each wave waiting for its results before the barrier, so that the table sees the positions in ascending order.
#ifndef FL_CHAIN_WAVES
#define FL_CHAIN_WAVES 8
#endif
#ifndef FL_CHAIN_TB
#define FL_CHAIN_TB 1  // blocks per wave and turn
#endif
template <bool FLUSH = false>  // FLUSH: windows of a stream with sync-flush points (the other instantiation pays nothing for them)
__global__ __launch_bounds__(64 * FL_CHAIN_WAVES, 4) void k_lz_chain(const uint8_t* __restrict__ in,
                                                                   const fl_chunk* __restrict__ chunks,
                                                                   uint16_t* __restrict__ prev_all,
                                                                   uint32_t* __restrict__ cflag,
                                                                   uint32_t* __restrict__ marks_all,
                                                                   const uint32_t* __restrict__ fpts) {
    __shared__ uint32_t head32[16384 + 64];  // (+ one word per lane for the exchanges of positions past the end)
    __shared__ uint32_t stg_all[FL_CHAIN_WAVES][FL_CHAIN_STG_DW];
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    if (marks_all) {
        // the chunk's anchor bitmap (k_lz_parse ORs into it): cleared here, not by a memset between the passes of the host path
        uint4* mk = (uint4*)(marks_all + (uint64_t)c * (FL_CHUNK_STRIDE / 32u));
        for (uint32_t i = threadIdx.x; i < FL_CHUNK_STRIDE / 128u; i += 64 * FL_CHAIN_WAVES) mk[i] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* sb = stg_all[wave];
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;  // positions with 4 bytes left (Lookup.zig:24)
    if (threadIdx.x == 0) cflag[c] = 0u;
    if (Mpos == 0) return;
    const uint8_t* src = in + ck.in_off;
    uint16_t* pv = prev_all + (uint64_t)c * FL_CHUNK_STRIDE;
    // A WINDOW of a stream with sync-flush points (round 6; the window's fl_chunk carries the stream's flush table and, in piece0,
    // the window's own position in the stream): the three positions before a flush point never enter the hash table
    // (deflate.zig:196-203 runs the tokenizer dry there, Lookup.zig:23-27 needs four bytes) and get no link themselves.
    const uint32_t* fp = fpts ? fpts + ck.flush_off : nullptr;
    const uint32_t wabs = ck.piece0;
    const bool has_fl = FLUSH && ck.pad_ && fp && ck.n_flush && (uint64_t)fl_next_flush(fp, ck.n_flush, wabs, 0xffffffffu) <= (uint64_t)wabs + N + 2u;
    const uint32_t sh = (uint32_t)((uintptr_t)src & 15);
    const uint4* src16 = (const uint4*)(src - sh);  // 16-byte granules; granule g covers chunk bytes 16 g - sh ..
    const uint32_t n_gran = (N + sh + 15) >> 4;     // granules holding at least one byte of the chunk
    // A chunk of ONE repeated byte (a run of zeros, say) needs no chains: k_lz_parse writes its anchors directly
    // (every position matches its predecessor over the whole lookahead).  Looked for granule by granule; the
    // scan of an ordinary chunk ends in its first step.
    if (N >= 64) {  // (pad_ != 0: a WINDOW of a long stream -- k_lz_parse<true> enters it anywhere: chains always, and cflag 3)
        const uint32_t b0 = src[0] * 0x01010101u;
        bool same = true;
        for (uint32_t g0 = 0; g0 < n_gran; g0 += 64 * FL_CHAIN_WAVES) {
            const uint32_t g = g0 + threadIdx.x;
            if (g < n_gran) {
                uint4 v = src16[g];
                // bytes of the granule outside the chunk do not count
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
            if (threadIdx.x == 0) cflag[c] = ck.pad_ ? 3u : 1u;
            if (!ck.pad_) return;
        }
    }
    {
        uint4* h4 = (uint4*)head32;
        for (uint32_t i = threadIdx.x; i < 4096; i += 64 * FL_CHAIN_WAVES) h4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // block b = chunk bytes [1024 b, 1024 b + 1024) plus what its last position needs: granules
    // 64 b .. 64 b + 65 (sh + 1023 + 3 < 1056 = 66 granules); lane l loads granule 64 b + l, lanes 0..1 two more
    auto load_block = [&](uint32_t b, uint4& g0, uint4& g1) {  // (granules beyond the chunk: zeros)
        const uint32_t ga = 64 * b + lane, gb = 64 * b + 64 + lane;
        g0 = ga < n_gran ? src16[ga] : make_uint4(0, 0, 0, 0);
        g1 = (lane < 2 && gb < n_gran) ? src16[gb] : make_uint4(0, 0, 0, 0);
    };
    const uint32_t n_blocks = (Mpos + 1023) >> 10;
    bool overtaken = false;
    // FL_CHAIN_TB consecutive blocks per wave and turn (round 5): a turn is mostly its barrier and the wait for the exchanges to
    // drain, whatever their number -- with two blocks per turn a chunk has 32 turns instead of 64.
    uint4 ga0[FL_CHAIN_TB], ga1[FL_CHAIN_TB];  // the wave's next blocks, in flight
#pragma unroll
    for (uint32_t u = 0; u < FL_CHAIN_TB; u++) load_block(FL_CHAIN_TB * wave + u, ga0[u], ga1[u]);
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += FL_CHAIN_TB * FL_CHAIN_WAVES) {  // (uniform trip count: every wave meets every barrier)
        const uint32_t bw = b0 + FL_CHAIN_TB * wave;  // this wave's first block of the turn
        // Round 6: address, mask and value of every exchange are made HERE, while other waves have their turns -- a wave alone
        // on its SIMD issues an instruction every eight cycles or so, and six ALU instructions per exchange inside the turn were
        // most of a turn's 1.6 k cycles (profiles/r06_parse_experiments.txt item 6); the turn itself is sixteen DS instructions.
        fl_lds_u32* am[16 * FL_CHAIN_TB];  // word of the table, or the lane's dummy word for a position past the end
        uint32_t mk[16 * FL_CHAIN_TB];     // which half of the word: 0xffff / 0xffff0000 (0: no position)
        uint32_t vl[16 * FL_CHAIN_TB];     // the position, in that half
#pragma unroll
        for (uint32_t u = 0; u < FL_CHAIN_TB; u++) {
            const uint32_t b = bw + u;
            ((uint4*)sb)[lane] = ga0[u];
            if (lane < 2) ((uint4*)sb)[64 + lane] = ga1[u];
            load_block(b + FL_CHAIN_TB * FL_CHAIN_WAVES, ga0[u], ga1[u]);  // (nothing beyond the chunk: load_block clamps)
            fl_lds_order();
#pragma unroll
            for (uint32_t s = 0; s < 16; s++) {
                const uint32_t p = (b << 10) + (s << 6) + lane;
                const uint32_t off = (s << 6) + lane + sh;
                const uint32_t v = __builtin_amdgcn_alignbyte(sb[(off >> 2) + 1], sb[off >> 2], off & 3);
                const uint32_t h = fl_hash_le(v);
                const bool valid = p < Mpos && (!FLUSH || !has_fl || fl_next_flush(fp, ck.n_flush, wabs + p, 0xffffffffu) - (wabs + p) >= 4u);
                const uint32_t hs = (h & 1u) << 4;
                am[16 * u + s] = (fl_lds_u32*)&head32[valid ? (h >> 1) : 16384u + lane];
                mk[16 * u + s] = valid ? (0xffffu << hs) : 0u;
                vl[16 * u + s] = valid ? (p << hs) : 0u;
            }
            fl_lds_order();  // (the staging buffer is written again for the next block)
        }
        uint32_t old[16 * FL_CHAIN_TB];
#pragma unroll 1
        for (uint32_t t = 0; t < FL_CHAIN_WAVES; t++) {
            if (t == wave) {
                // all exchanges are issued before the first result is looked at (no branch around the instruction)
#pragma unroll
                for (uint32_t i = 0; i < 16 * FL_CHAIN_TB; i++)
                    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=v"(old[i]) : "v"(am[i]), "v"(mk[i]), "v"(vl[i]) : "memory");
                // the results exist from here on (listed as operands so that no use of them is scheduled above the wait)
#pragma unroll
                for (uint32_t u = 0; u < FL_CHAIN_TB; u++) {
                    uint32_t* o = old + 16 * u;
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]),
                                   "+v"(o[6]), "+v"(o[7]), "+v"(o[8]), "+v"(o[9]), "+v"(o[10]), "+v"(o[11]),
                                   "+v"(o[12]), "+v"(o[13]), "+v"(o[14]), "+v"(o[15])
                                 :
                                 : "memory");
                }
            }
            __syncthreads();
        }
        // the links: the half of the returned word this position's hash selects
#pragma unroll
        for (uint32_t u = 0; u < FL_CHAIN_TB; u++) {
#pragma unroll
            for (uint32_t s = 0; s < 16; s++) {
                const uint32_t p = ((bw + u) << 10) + (s << 6) + lane;
                if (mk[16 * u + s]) {
                    const uint32_t o = (mk[16 * u + s] >> 16) ? (old[16 * u + s] >> 16) : (old[16 * u + s] & 0xffffu);
                    overtaken = overtaken || o > p;
                    pv[p] = (uint16_t)o;  // 0 = none: position 0 is the chain's null (deflate.zig:248)
                } else if (FLUSH && has_fl && p < Mpos) {
                    pv[p] = 0;  // (kept out of the table by a flush point: a call there finds nothing)
                }
            }
        }
    }
#ifdef FL_CHAIN_FORCE_SLOW
    overtaken = true;  // (test builds: exercise the fallback)
#endif
#ifdef FL_CHAIN_NO_FALLBACK
    overtaken = false;  // (test builds: the fast path alone, tools/ubench/chain_test.hip)
#endif
    if (__syncthreads_or(overtaken ? 1 : 0)) {
        // never seen on gfx950: one position at a time, by one lane (Lookup.zig:35-40 as written)
        uint16_t* head16 = (uint16_t*)head32;
        for (uint32_t i = threadIdx.x; i < 16384; i += 64 * FL_CHAIN_WAVES) head32[i] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t p = 0; p < Mpos; p++) {
                if (FLUSH && has_fl && fl_next_flush(fp, ck.n_flush, wabs + p, 0xffffffffu) - (wabs + p) < 4u) {
                    pv[p] = 0;
                    continue;
                }
                const uint32_t h = fl_hash_le(fl_load_u32_clamped(src, p, N));
                pv[p] = head16[h];
                head16[h] = (uint16_t)p;
            }
        }
    }
}

// ------------------------------------------------------------------ k_lz_parse
#ifndef PZ_THREADS
#define PZ_THREADS 1024
#endif
#define PZ_WAVES (PZ_THREADS / 64)
#define PZ_TA 49152u     // targets of sub-pass A: [0, PZ_TA); sub-pass B: [PZ_TA, 65536)
#define PZ_MARGIN 64u    // sub-pass B keeps this many positions more than the farthest candidate, so that relative position 0 is never one
#define PZ_LOOK 256u     // the lazy calls of a sub-pass's last anchor go at most this far past its targets (lazy <= 258)
#define PZ_PRV_N (PZ_TA + PZ_LOOK)                             // chain links per sub-pass (A: 49408, B: 49216)
#define PZ_WIN_DW ((PZ_TA + PZ_LOOK + FL_MAX_MATCH + 30u) / 4u)  // window bytes per sub-pass, zero padded (A: 49680, B: 49232)
#define PZ_NONE 0xffffu
#define PZ_NOHIT 0xffffffffu
#define PZ_DESC_LIT 0x40000000u  // descriptor of an anchor that emits one literal
#ifndef PZ_BURST
#define PZ_BURST 20              // chain steps between two visits of the slow block, at most
#endif
#ifndef PZ_NEED
#define PZ_NEED 48               // ... fewer when this many lanes wait for the slow block
#endif
#ifndef PZ_UNROLL
#define PZ_UNROLL 20              // chain steps between two looks at the other lanes
#endif
#ifndef PZ_ROT
#define PZ_ROT 3                 // the burst is PZ_ROT x 6 steps, registers rotating (0: the two-role burst of round 4, PZ_UNROLL steps)
#endif
#ifndef PZ_RUNSKIP
#define PZ_RUNSKIP 0                // 1: a call inside a run of one byte takes the run's members below p - 1 together (bit-exact; sparse zeros 3 % faster, text 1 % slower: off)
#endif
#ifndef PZ_TRANS_ITERS
#define PZ_TRANS_ITERS 1         // automaton moves per lane and slow block (runs of literals)
#endif
#define PZ_STR2(X) #X
#define PZ_STR(X) PZ_STR2(X)
#if PZ_UNROLL == 8
#define PZ_HALF_UNROLL 4
#elif PZ_UNROLL == 4
#define PZ_HALF_UNROLL 2
#elif PZ_UNROLL == 12
#define PZ_HALF_UNROLL 6
#elif PZ_UNROLL == 20
#define PZ_HALF_UNROLL 10
#elif PZ_UNROLL == 16
#define PZ_HALF_UNROLL 8
#elif PZ_UNROLL == 24
#define PZ_HALF_UNROLL 12
#elif PZ_UNROLL == 32
#define PZ_HALF_UNROLL 16
#else
#error PZ_UNROLL: 4, 8, 12, 16, 20, 24 or 32
#endif
#define PZ_SEG_A (PZ_TA / PZ_THREADS)  // 48 bytes per lane with 1024 lanes (64 with 768)
#ifndef PZ_SEG_B
#define PZ_SEG_B 24u   // bytes per segment in sub-pass B (16384 targets; 32: 8 of the 16 waves hold segments, 20.47 ms; 24: 11 waves, 20.34; 16: all of them, 21.98 with unequal segments in A)
#endif
#if PZ_SEG_B == 32
#define PZ_SEG_B_OF(D) ((D) >> 5)
#elif PZ_SEG_B == 24
#define PZ_SEG_B_OF(D) (((D) * 43691u) >> 20)
#elif PZ_SEG_B == 16
#define PZ_SEG_B_OF(D) ((D) >> 4)
#else
#error PZ_SEG_B: 16, 24 or 32
#endif
// Sub-pass A of a chunk, round 6: segments of UNEQUAL size.  The cost of a segment grows with its position -- the chains of a
// position hold what lies before it, up to 32 KiB -- and a wave is as slow as its slowest lanes: with 48 bytes everywhere wave 0
// (positions 0 .. 3071) finished its own parse in half the time of wave 15 (profiles/r06_parse_experiments.txt item 1) and waited.
// Wave w takes 64 consecutive segments of PZ_VSIZES[w] bytes each (at most 64: a lane's anchors are a 64-bit mask), the waves in
// order of position; the sizes add up to PZ_TA / 64.
#ifndef PZ_VARY
#define PZ_VARY 0
#endif
#ifndef PZ_WREV
#define PZ_WREV 1
#endif
#ifndef PZ_VSIZES
#define PZ_VSIZES 53, 46, 43, 42, 47, 46, 46, 45, 48, 48, 48, 48, 52, 52, 52, 52
#endif
struct pz_vgeom {
    uint32_t size[16], base[17];
    float rcp[16];
};
constexpr pz_vgeom pz_make_vgeom() {
    pz_vgeom g{};
    const uint32_t sz[16] = {PZ_VSIZES};
    uint32_t b = 0;
    for (int w = 0; w < 16; w++) {
        g.size[w] = sz[w];
        g.base[w] = b;
        g.rcp[w] = 1.0f / (float)sz[w];
        b += 64u * sz[w];
    }
    g.base[16] = b;
    return g;
}
constexpr pz_vgeom PZ_VG = pz_make_vgeom();
static_assert(PZ_THREADS != 1024 || PZ_VG.base[16] >= PZ_TA, "the segments of sub-pass A cover its targets");
// Segment of the target at offset D from the sub-pass's first.  The blocks are about 3072 bytes: block D / 3072 or one of its
// neighbours holds D (checked below for the sizes chosen), so the block's number comes from two bounds of a small table in LDS,
// the lane from a division that is exact in single precision (D - base < 4096, sizes <= 64: (d + 0.5) / size is at least
// 1 / 128 away from an integer).
constexpr bool pz_vgeom_ok() {
    for (int k = 0; k <= 16; k++)
        if (PZ_VG.base[k] > 3072u * (k + 1) || PZ_VG.base[k] + 3072u < 3072u * k) return false;
    return true;
}
static_assert(pz_vgeom_ok(), "block k of sub-pass A starts within 3072 bytes of 3072 k");
struct pz_vtab {
    uint16_t base[20];  // [k + 1] = first target of block k; [0] = 0, [18], [19] = beyond every target
    float rcp[17];      // [k + 1] = 1 / bytes per segment of block k
};
__device__ __forceinline__ uint32_t pz_vseg(const pz_vtab& vt, uint32_t D) {
    const uint32_t g = (D * 43691u) >> 27;  // D / 3072 (exact below 2^16)
    const uint32_t w = g - 1u + (D >= vt.base[g + 1] ? 1u : 0u) + (D >= vt.base[g + 2] ? 1u : 0u);  // in [g - 1, g + 1]; g = 0: base[1] = 0 counts
    return 64u * w + (uint32_t)(((float)(D - vt.base[w + 1]) + 0.5f) * vt.rcp[w + 1]);
}
// STREAM: sub-pass A of a window has only the targets [32506, 49152): smaller segments, so that every lane has one
#ifndef PZ_SEG_AS
#define PZ_SEG_AS 24u  // (256 x 1 MiB of text, k_lz_parse<true>: 48 / 32 / 24 / 17 bytes 9.91 / 9.29 / 9.04 / 9.57 ms)
#endif
#if PZ_SEG_AS == 17
#define PZ_SEG_AS_OF(D) (((D) * 61681u) >> 20)   // 61681 / 2^20 = 1 / 17.00002: exact below 2^16
#elif PZ_SEG_AS == 24
#define PZ_SEG_AS_OF(D) (((D) * 43691u) >> 20)
#elif PZ_SEG_AS == 32
#define PZ_SEG_AS_OF(D) ((D) >> 5)
#elif PZ_SEG_AS == 48
#define PZ_SEG_AS_OF(D) (((D) * 43691u) >> 21)
#else
#error PZ_SEG_AS: 17, 24, 32 or 48
#endif
// ... and sub-pass B of a window (16122 targets)
#ifndef PZ_SEG_BS
#define PZ_SEG_BS 24u  // (256 x 1 MiB of text, k_lz_parse<true>: 32 / 24 / 16 bytes 8.93 / 8.80 / 9.23 ms)
#endif
#if PZ_SEG_BS == 32
#define PZ_SEG_BS_OF(D) ((D) >> 5)
#elif PZ_SEG_BS == 24
#define PZ_SEG_BS_OF(D) (((D) * 43691u) >> 20)
#elif PZ_SEG_BS == 16
#define PZ_SEG_BS_OF(D) ((D) >> 4)
#else
#error PZ_SEG_BS: 16, 24 or 32
#endif

// tuning counters (compiled in with -DPZ_PROF; read with tools/parse_probe.py): per wave, summed over the grid
#ifdef PZ_PROF
#define PZ_CNT(var, v) (var) += (v)
// an event of the slow block: c_ev[K] counts the waves that pass here (its first active lane counts), c_ev[K + 1] the lanes
#define PZ_EV(K)                                                                                   \
    do {                                                                                           \
        c_ev[(K) + 1]++;                                                                           \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(1))) c_ev[(K)]++;            \
    } while (0)
#elif defined(PZ_SEC)  // markers in the assembly listing: instructions per part of the slow block (tools/parse_sections.py)
#define PZ_CNT(var, v)
#define PZ_EV(K) asm volatile(";;PZSEC " #K)
#else
#define PZ_CNT(var, v)
#define PZ_EV(K)
#endif

__device__ __forceinline__ uint32_t pz_lds4(const uint32_t* win32, uint32_t off) {
    const uint32_t* w = win32 + (off >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], off);
}

// STREAM (round 5): the whole-stream path on this tokenizer.  The reference's window after slide j IS a chunk: 65536 bytes from
// stream position 32768 j, its chain the window's own (Lookup.zig:43-51 drops what lies at or below the window start: relative
// position 0 is the chain's null there as here), its targets the positions the tokenizer visits before the next slide --
// [first position visited after slide j, first one visited after slide j + 1) (SlidingWindow.zig:56-60, deflate.zig:304-321),
// window-relative [32506, 65274) for every window but the first and the last.  One workgroup per stream walks its windows in
// order (`swins`), the anchor the path leaves a window at is where it enters the next.  Chains: k_lz_chain on the windows as
// chunks (`chunks` holds one fl_chunk per WINDOW then, `schunks` the streams).  Descriptors and anchor bits go to the stream's
// arrays at absolute positions, a literal's descriptor is 0 there (k_st_emit, kernels_stream.h).
// A workgroup's share is a GROUP of consecutive windows of one stream: the whole stream when the streams are many, a part of it
// when they are few.  A group that is not its stream's first does not know where the path enters its first window: it is parsed
// from a guess (the window's first target -- the speculation of the segments, one level up), every sub-pass leaves its exit in
// `wexit`, the group its own in `gexit`; a second launch (fix = 1) parses from the TRUE entry -- the exit of the group before --
// until a sub-pass leaves where it left before: from there on the anchors of the first launch stand (a parse from any entry
// falls in step with the true one within a few bytes, so this is the first sub-pass of the group's first window).  A group that
// had to be parsed to its end with a new exit sets `dirty`: the host launches the fix again (never seen).
struct fl_swin {
    uint32_t chunk;   // the stream (index into schunks)
    uint32_t win0;    // the group's first window (index into chunks / the chain links)
    uint32_t nwin;    // windows in the group
    uint32_t wfirst;  // number of the group's first window in its stream
    uint32_t prev;    // the group before it in the stream (index into swins), ~0u: the stream's first
    uint32_t pad_[3];
};
template <bool STREAM>
__global__ __launch_bounds__(PZ_THREADS, PZ_THREADS / 256) void k_lz_parse(const uint8_t* __restrict__ in,
                                                           const fl_chunk* __restrict__ chunks, fl_params prm,
                                                           const uint16_t* __restrict__ prev_all,
                                                           const uint32_t* __restrict__ cflag,
                                                           uint32_t* __restrict__ desc_all,
                                                           uint32_t* __restrict__ true_all,
                                                           const fl_swin* __restrict__ swins,
                                                           const fl_chunk* __restrict__ schunks,
                                                           const uint32_t* __restrict__ zones,
                                                           uint32_t* gexit, uint32_t* gentry, uint32_t* wexit,
                                                           uint32_t* dirty, uint32_t fix_cap,
                                                           const uint32_t* __restrict__ fpts) {
    const uint32_t fix = fix_cap & 1u;         // STREAM: the launch from the groups' true entries
    const uint32_t round_cap = fix_cap >> 8;   // STREAM: rounds of the stitch after which a window is given up (0: never)
    // One block of LDS in this order: the window at address 0 (a window byte's LDS address is its position: no base to add),
    // the links behind it (their base, 49680, fits the offset field of the LDS instructions).
    struct pz_lds {
        uint32_t win32[PZ_WIN_DW];
        uint16_t prv[PZ_PRV_N];
        uint16_t tX[PZ_THREADS];       // exit of a lane's own parse, as soon as it is known
        uint16_t tExg[PZ_THREADS];     // exit the path is assumed to take out of a segment
        uint16_t tNxt[2][PZ_THREADS];  // segment that exit lands in (pointer jumping, double buffered)
        uint16_t tEnt[PZ_THREADS];     // position at which the path enters a segment
        uint16_t tMark[PZ_THREADS];    // segment is on the path
        uint32_t sh_next_entry;
        uint32_t sh_exit;              // STREAM: where the path leaves the window (window-relative)
        pz_vtab vt;                    // the blocks of sub-pass A (PZ_VARY)
    };
    __shared__ pz_lds lds;
    if (PZ_VARY && !STREAM && threadIdx.x < 20) {
        const uint32_t k = threadIdx.x;  // (read behind the barrier that ends the first staging)
        lds.vt.base[k] = (uint16_t)(k == 0 ? 0u : k <= 17 ? min(PZ_VG.base[k - 1], 65535u) : 65535u);
        if (k >= 1 && k <= 16) lds.vt.rcp[k] = PZ_VG.rcp[k - 1];
    }
    if (PZ_VARY && !STREAM) __syncthreads();
    uint32_t (&win32)[PZ_WIN_DW] = lds.win32;
    uint16_t (&prv)[PZ_PRV_N] = lds.prv;
    uint16_t (&tX)[PZ_THREADS] = lds.tX;
    uint16_t (&tExg)[PZ_THREADS] = lds.tExg;
    uint16_t (&tNxt)[2][PZ_THREADS] = lds.tNxt;
    uint16_t (&tEnt)[PZ_THREADS] = lds.tEnt;
    uint16_t (&tMark)[PZ_THREADS] = lds.tMark;
    uint32_t& sh_next_entry = lds.sh_next_entry;
    uint32_t& sh_exit = lds.sh_exit;
    constexpr uint32_t LITD = STREAM ? 0u : PZ_DESC_LIT;  // descriptor of an anchor that emits one literal
    const uint32_t tid = threadIdx.x;
#ifdef PZ_PRIO
    {
        // experiment: the issue priority of a wave by its number (the hardware serves the oldest wave of a SIMD first)
        const uint32_t g = (PZ_PRIO == 1) ? (tid >> 8) : (PZ_PRIO == 2) ? 3u - (tid >> 8) : (PZ_PRIO == 3) ? ((tid >> 6) & 3u) : 0u;
        if (g == 1) __builtin_amdgcn_s_setprio(1);
        else if (g == 2) __builtin_amdgcn_s_setprio(2);
        else if (g == 3) __builtin_amdgcn_s_setprio(3);
    }
#endif
    const uint32_t chain = prm.chain, good = prm.good, lazy = prm.lazy, nice = prm.nice;
    const uint32_t prv_lds = (uint32_t)(size_t)(fl_lds_u32*)prv, win_lds = (uint32_t)(size_t)(fl_lds_u32*)win32;  // LDS byte addresses
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
        sw = swins[blockIdx.x];
        sck = schunks[sw.chunk];
        if (sck.skip) return;
        if (sw.prev != ~0u) {
            if (fix) {
                // the true entry: where the group before leaves (absolute stream position).  ONE thread reads it: the group
                // before may store a new exit in this very launch, and threads that saw different values would part ways here
                // (some return, the others parse on with two different entries)
                if (tid == 0) {
                    sh_next_entry = __hip_atomic_load(&gexit[sw.prev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh_exit = gentry[blockIdx.x];
                }
                __syncthreads();
                const uint32_t e = sh_next_entry, was = sh_exit;
                __syncthreads();  // (both slots are written again further down)
                if (e == was) return;  // parsed from there already
                if (tid == 0) gentry[blockIdx.x] = e;
                carry = e - FL_MAX_DIST * sw.wfirst;
            } else {
                guessed = true;  // (the first window's first target: set below)
            }
        } else if (fix) {
            return;  // a stream's first group starts at the stream's start
        }
    }
    for (uint32_t wi = 0; wi < sw.nwin; wi++) {
    const uint32_t ws = sw.wfirst + wi;  // STREAM: the window's number in its stream
    const uint32_t c = sw.win0 + wi;
    const fl_chunk ck = chunks[c];
    if (!STREAM && ck.skip) return;
    const uint32_t N = ck.in_len;  // (the sub-passes read the chunk's fields again: see there)
    const uint64_t pos_off = STREAM ? sck.pos_off + (uint64_t)FL_MAX_DIST * ws : ck.pos_off;
    uint32_t* descg = desc_all + pos_off;
    uint32_t* trueg = true_all + (pos_off >> 5);
    // the window's targets [t_first, t_last)
    uint32_t t_first = 0, t_last = N;
    if (STREAM) {
        const uint32_t* zone = zones + sck.zone_off;
        if (ws) t_first = zone[ws - 1] - FL_MAX_DIST * ws;
        if (ws < sck.n_slides) t_last = zone[ws] - FL_MAX_DIST * ws;
        if (wi) __syncthreads();  // the window before is done with the LDS tables
        if (guessed && wi == 0) {
            carry = t_first;
            if (tid == 0) gentry[blockIdx.x] = t_first + FL_MAX_DIST * ws;
        }
        if (t_first >= t_last) {
            if (tid == 0) wexit[2 * c] = wexit[2 * c + 1] = ~0u;
            continue;
        }
    }
    if (STREAM && round_cap && cflag[c] == 3u) {
        // a window of one repeated byte in a pass that gives up on periodic data (see the round loop): at once
        if (tid == 0) atomicOr(dirty, 0x80000000u);
        return;
    }
    if (cflag[c] == 1u) {
        // The chunk is one repeated byte (k_lz_chain saw it and built no chains).  Positions 0 and 1 are
        // literals (the only candidate of position 1 is position 0, the chain's null: deflate.zig:248); from
        // position 2 on the nearest candidate is the position before, it matches over the whole lookahead,
        // and a match that long ends the walk at every level (deflate.zig:254-258): anchors 2, 260, 518, ...
        // each emit (min(258, N - a), distance 1); what is left behind the last match (< 4 bytes: no hash
        // entry, Lookup.zig:24) goes out as literals.  A last match shorter than `lazy` waits for the next
        // position, whose match is one byte shorter, and goes out unchanged (deflate.zig:182-184).
        for (uint32_t k = tid; 2 + FL_MAX_MATCH * k < N || k < 1; k += PZ_THREADS) {
            if (k == 0) {
                uint32_t w = 0;
                for (uint32_t p = 0; p < min(N, 2u); p++) {
                    descg[p] = LITD;
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
                // (the next anchor is a + len: the next step's, or the end of the chunk)
            } else {
                for (uint32_t p = a; p < N; p++) {
                    descg[p] = LITD;
                    atomicOr(&trueg[p >> 5], 1u << (p & 31u));
                }
            }
        }
        return;
    }
#ifdef PZ_PROF
    uint64_t c_tb_le2 = 0, c_tb_le8 = 0, c_tb_le24 = 0, c_tb_more = 0, c_nb_le2 = 0;
    uint32_t c_ev[32] = {0};
    uint32_t c_fast = 0, c_walk = 0, c_slow = 0, c_meas = 0, c_measl = 0, c_trans = 0, c_transl = 0, c_rounds = 0, c_loops = 0;
    uint64_t c_t0 = __builtin_readcyclecounter(), c_tspec = 0, c_tstitch = 0, c_tfast = 0, c_tmeas = 0, c_ttrans = 0, c_tstage = 0, c_tjump = 0;
#endif

    for (uint32_t sub = 0; sub < 2; sub++) {
        // (STREAM: the window's fields anew in every sub-pass -- the kernel is short of SCALAR registers, 187 of them lived in
        // vector registers and 35 vector registers in scratch: what the staging needs must not live through the automaton)
        uint32_t c_l = c;
        if (STREAM) asm volatile("" : "+s"(c_l));
        const fl_chunk ckl = chunks[c_l];
        const uint32_t N = ckl.in_len;
        const uint32_t Mpos = N >= 4 ? N - 3 : 0u;
        const uint8_t* src = in + ckl.in_off;
        const uint16_t* pvg = prev_all + (uint64_t)c_l * FL_CHUNK_STRIDE;
        const uint32_t t0 = sub ? max(PZ_TA, t_first) : t_first;
        const uint32_t end = min(t_last, sub ? 65536u : PZ_TA);  // targets [t0, end)
        if (t0 >= end) {  // (a window's first target lies below PZ_TA: sub-pass B never runs without A's staging)
            if (STREAM && tid == 0) wexit[2 * c + sub] = ~0u;
            break;
        }
        if (STREAM && fix) {
            // the anchors the first launch left in this sub-pass's targets go: bit by bit at the ends (the words there are
            // shared with the sub-passes next to it, which other workgroups may be writing), whole words in between
            const uint64_t b0 = pos_off + t0, b1 = pos_off + end;  // absolute bits [b0, b1)
            for (uint64_t w = (b0 >> 5) + tid; w <= ((b1 - 1) >> 5); w += PZ_THREADS) {
                uint32_t keep = 0;
                if (w == (b0 >> 5) && (b0 & 31)) keep |= (1u << (b0 & 31)) - 1u;
                if (w == ((b1 - 1) >> 5) && (b1 & 31)) keep |= ~((1u << (b1 & 31)) - 1u);
                if (keep) atomicAnd(&true_all[w], keep); else true_all[w] = 0u;
            }
            __threadfence();
        }
        const uint32_t r0 = sub ? (PZ_TA - FL_MAX_DIST - PZ_MARGIN) : 0u;  // everything below is relative to r0
        // (STREAM: small segments when the sub-pass's targets leave a lane for each; a stream's first window starts at 0)
        const bool small = STREAM && !sub && min(t_last, (uint32_t)PZ_TA) - t_first <= PZ_THREADS * PZ_SEG_AS;
        const bool vary = PZ_VARY && !STREAM && !sub && PZ_THREADS == 1024;  // (a chunk's sub-pass A starts at position 0)
        const uint32_t S = sub ? (STREAM ? PZ_SEG_BS : PZ_SEG_B) : (small ? PZ_SEG_AS : PZ_SEG_A);
        // segment of a relative target position x - t0r (< 65536): a shift, or a multiplication by 1 / 48
        // (43691 / 2^21 = 1 / 47.99997: exact for arguments below 2^16)
#define PZ_SEG_OF(D) (sub ? (STREAM ? PZ_SEG_BS_OF(D) : PZ_SEG_B_OF(D)) : (small ? PZ_SEG_AS_OF(D) : (vary ? pz_vseg(lds.vt, D) : (PZ_SEG_A == 64u ? ((D) >> 6) : (((D) * 43691u) >> 21)))))
        const uint32_t nseg = vary ? pz_vseg(lds.vt, end - t0 - 1u) + 1u : PZ_SEG_OF(end - t0 + S - 1);
        const uint32_t Nr = N - r0;            // end of the input
        // (STREAM: + the bytes of the stream behind the window, at most 264 (pad_ >> 8): the lazy calls of a window's last anchor are
        // made after the next slide, with the whole lookahead -- deflate.zig:304-321; 65536 - 65279 = 257 bytes are not all such a
        // call may look at: tests/test_gpu_stream.py::test_lazy_chain_across_a_slide_sees_the_whole_lookahead)
        const uint32_t NBr = STREAM ? Nr + (ckl.pad_ >> 8) : Nr;
        // (STREAM, round 6: sync-flush points -- a flush at F ends the lookahead of everything before it: no match crosses F
        // (deflate.zig:196-203); k_lz_chain has kept F - 3 .. F - 1 out of the table)
        const uint32_t* fpl = (STREAM && fpts) ? fpts + sck.flush_off : nullptr;
        const uint32_t nfl = STREAM ? sck.n_flush : 0u;
        const uint32_t wabs_r = FL_MAX_DIST * ws + r0;  // stream position of relative position 0
        const bool has_fl = STREAM && fpl && nfl && (uint64_t)fl_next_flush(fpl, nfl, FL_MAX_DIST * ws, 0xffffffffu) <= (uint64_t)FL_MAX_DIST * ws + 65536u + 300u;
        const uint32_t endr = end - r0, t0r = t0 - r0;
        // STREAM: a position at or beyond the window's last target is visited AFTER the next slide (a lazy call of the window's
        // last anchor gets there): the reference has dropped every candidate at or below the next window's start by then
        // (Lookup.zig:43-51) -- relative to this window: at or below 32768
        const uint32_t zt = (STREAM && ws < sck.n_slides) ? t_last - r0 : ~0u;
        const uint32_t zlo = FL_MAX_DIST + 1u > r0 ? FL_MAX_DIST + 1u - r0 : 1u;
        if (sub) __syncthreads();  // the previous sub-pass is done with the LDS tables
        // ---- stage window bytes and chain links (loads in batches: one round of memory latency per batch)
#ifdef PZ_PROF
        const uint64_t c_ts0 = __builtin_readcyclecounter();
#endif
        {
            // Sub-pass B needs positions [r0, 65536); [r0, PZ_TA + ...) of them sit in LDS already, r0 positions further up
            // (r0 is a multiple of 4: whole dwords of both tables).  They are moved down -- the links re-based -- and only
            // the rest comes from memory: a third of the loads.
            const uint32_t wkeep = sub ? PZ_WIN_DW - (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u : 0u;          // window dwords that stay
            const uint32_t pkeep = sub ? (PZ_PRV_N - (PZ_TA - FL_MAX_DIST - PZ_MARGIN)) / 2u : 0u;         // link dwords that stay
            // The loads are 16 bytes wide: a granule of the input (aligned; the window is that shifted by sh16 bytes: dshift
            // dwords and ash bytes) and the first dword of the granule behind it per four window dwords, eight links per
            // load -- a quarter of the vector-memory instructions of one dword per load (staging was 9 % of the kernel and
            // bound by their issue).
            const uint32_t sh16 = (uint32_t)((uintptr_t)(src + r0) & 15);
            const uint32_t ash = sh16 & 3u, dshift = sh16 >> 2;
            const uint4* src16 = (const uint4*)(src + r0 - sh16);
            const uint32_t ngran = (NBr + sh16 + 15) >> 4;  // granules that hold at least one byte of the input
            const uint32_t nb_pos = min(Nr, (uint32_t)PZ_PRV_N);  // positions whose links are staged
            const uint4* pv4 = (const uint4*)(pvg + r0);         // (r0 * 2 bytes is a multiple of 16)
            uint32_t* prv2 = (uint32_t*)prv;
            auto put_win = [&](uint32_t i, uint32_t lo, uint32_t hi) {
                uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, ash);
                if (4 * i + 4 > NBr) v = 4 * i < NBr ? (v & ((1u << (8 * (NBr - 4 * i))) - 1u)) : 0u;  // zero padding
                if (i < PZ_WIN_DW) win32[i] = v;
            };
            auto put_prv = [&](uint32_t i, uint32_t v) {  // two links per dword; relative to r0, 0 = none (also everything below r0)
                const uint32_t pa = 2 * i + r0;  // absolute position of the low half
                if (pa >= Mpos) v &= 0xffff0000u;  // positions without a hash entry have no link (never written)
                if (pa + 1 >= Mpos) v &= 0x0000ffffu;
                uint32_t lo16 = v & 0xffffu, hi16 = v >> 16;
                lo16 = lo16 > r0 ? lo16 - r0 : 0u;
                hi16 = hi16 > r0 ? hi16 - r0 : 0u;
                if (i < PZ_PRV_N / 2) prv2[i] = lo16 | (hi16 << 16);
            };
            // granule G holds the window dwords 4 G - dshift .. 4 G - dshift + 3 (those from `first` on are written)
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
            auto put_links = [&](uint32_t i4, const uint4& v) {
                put_prv(4 * i4, v.x);
                put_prv(4 * i4 + 1, v.y);
                put_prv(4 * i4 + 2, v.z);
                put_prv(4 * i4 + 3, v.w);
            };
            if (!sub) {
                // everything a thread stages is requested before anything is written: one round of memory latency
                constexpr uint32_t WG = ((PZ_WIN_DW + 6) / 4 + PZ_THREADS - 1) / PZ_THREADS, PG = (PZ_PRV_N / 8 + PZ_THREADS - 1) / PZ_THREADS;
                uint4 wg[WG], lg[PG];
                uint32_t wx[WG];
#pragma unroll
                for (uint32_t u = 0; u < WG; u++) load_gran(u * PZ_THREADS + tid, wg[u], wx[u]);
#pragma unroll
                for (uint32_t u = 0; u < PG; u++) {
                    const uint32_t i4 = u * PZ_THREADS + tid;
                    lg[u] = 8 * i4 < nb_pos ? pv4[i4] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (uint32_t u = 0; u < WG; u++) put_gran(u * PZ_THREADS + tid, wg[u], wx[u], 0u);
#pragma unroll
                for (uint32_t u = 0; u < PG; u++) put_links(u * PZ_THREADS + tid, lg[u]);
            } else {
                // the new part: loads first ...
                constexpr uint32_t WNEW = (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u;                       // window dwords that come from memory
                constexpr uint32_t PNEW = PZ_PRV_N / 2 - (PZ_PRV_N - (PZ_TA - FL_MAX_DIST - PZ_MARGIN)) / 2u;  // link dwords
                constexpr uint32_t WNG = ((WNEW + 6) / 4 + PZ_THREADS - 1) / PZ_THREADS, PNG = (PNEW / 4 + PZ_THREADS - 1) / PZ_THREADS;
                static_assert(PNEW % 4 == 0 && ((PZ_PRV_N - (PZ_TA - FL_MAX_DIST - PZ_MARGIN)) / 2u) % 4 == 0, "whole 16-byte loads of links");
                const uint32_t Gfirst = (wkeep + dshift) >> 2;
                uint4 wg[WNG], lg[PNG];
                uint32_t wx[WNG];
#pragma unroll
                for (uint32_t u = 0; u < WNG; u++) load_gran(Gfirst + u * PZ_THREADS + tid, wg[u], wx[u]);
#pragma unroll
                for (uint32_t u = 0; u < PNG; u++) {
                    const uint32_t i4 = pkeep / 4 + u * PZ_THREADS + tid;
                    lg[u] = (8 * i4 < nb_pos && 4 * i4 < PZ_PRV_N / 2) ? pv4[i4] : make_uint4(0, 0, 0, 0);
                }
                // ... then what stays: read by everybody, a barrier, written r0 positions further down
                constexpr uint32_t WK = ((PZ_WIN_DW - (PZ_TA - FL_MAX_DIST - PZ_MARGIN) / 4u) + PZ_THREADS - 1) / PZ_THREADS;
                constexpr uint32_t PK = ((PZ_PRV_N - (PZ_TA - FL_MAX_DIST - PZ_MARGIN)) / 2u + PZ_THREADS - 1) / PZ_THREADS;
                uint32_t kw[WK], kp[PK];
#pragma unroll
                for (uint32_t u = 0; u < WK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    kw[u] = i < wkeep ? win32[i + r0 / 4u] : 0u;
                }
#pragma unroll
                for (uint32_t u = 0; u < PK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    kp[u] = i < pkeep ? prv2[i + r0 / 2u] : 0u;
                }
                __syncthreads();
#pragma unroll
                for (uint32_t u = 0; u < WK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    if (i < wkeep) win32[i] = kw[u];
                }
#pragma unroll
                for (uint32_t u = 0; u < PK; u++) {
                    const uint32_t i = u * PZ_THREADS + tid;
                    if (i < pkeep) {  // (sub-pass A's links count from position 0)
                        const uint32_t v = kp[u];
                        uint32_t lo16 = v & 0xffffu, hi16 = v >> 16;
                        lo16 = lo16 > r0 ? lo16 - r0 : 0u;
                        hi16 = hi16 > r0 ? hi16 - r0 : 0u;
                        prv2[i] = lo16 | (hi16 << 16);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < WNG; u++) put_gran(Gfirst + u * PZ_THREADS + tid, wg[u], wx[u], wkeep);
#pragma unroll
                for (uint32_t u = 0; u < PNG; u++) put_links(pkeep / 4 + u * PZ_THREADS + tid, lg[u]);
            }
        }
        __syncthreads();
#ifdef PZ_PROF
        c_tstage += __builtin_readcyclecounter() - c_ts0;
#endif
        const uint32_t y0 = sub ? sh_next_entry : carry;  // the sub-pass is entered at this anchor (relative)
        // this lane's segment.  Sub-pass A of a chunk: the waves take the blocks of 64 segments in REVERSE order -- a segment costs
        // the more the later it lies (its chains hold what lies before it) and a SIMD serves its oldest waves first (the same
        // work takes wave 15 a third longer than wave 0: profiles/r06_parse_experiments.txt item 1): the dearest segments go to
        // the waves that are served best.
        const uint32_t m = (PZ_WREV && !STREAM && !sub) ? (((PZ_WAVES - 1u - (tid >> 6)) << 6) | (tid & 63u)) : tid;
        const uint32_t Sm = vary ? PZ_VG.size[m >> 6] : S;  // this lane's segment: [seg0, seg_end)
        const uint32_t seg0 = vary ? t0r + PZ_VG.base[m >> 6] + (m & 63u) * Sm : t0r + m * S;
        const uint32_t seg_end = min(seg0 + Sm, endr);
        if (y0 >= endr) {  // the path jumps over the whole sub-pass: no anchors (the bitmap is zero already)
            if (tid == 0) {
                sh_exit = y0 + r0;
                sh_next_entry = sub ? y0 : y0 - (PZ_TA - FL_MAX_DIST - PZ_MARGIN);  // (y0 >= 49152 - 16320 here)
            }
            if (STREAM) {
                __syncthreads();
                const uint32_t was = wexit[2 * c + sub];
                __syncthreads();
                if (tid == 0) wexit[2 * c + sub] = sh_exit;
                if (fix && was == sh_exit) return;  // from here on the first launch's anchors stand
            }
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

        // Lane states.  Round 0: SPEC (the lane's own parse from the start of its segment), then WAIT until the
        // lane before has published its exit, then FIX (the parse from that exit, if it lands in this segment
        // on a position the own parse did not visit: almost always where the true path enters), then DONE.
        // Later rounds: FIX from the entry the stitch has found, for the few segments where that guess was wrong.
        enum { ST_SPEC = 0, ST_WAIT = 1, ST_FIX = 2, ST_DONE = 3 };
        for (uint32_t round = 0;; round++) {
            // STREAM: a window whose stitch does not settle -- PERIODIC data: paths from different entries never meet, every round
            // settles one more segment (1.65 ms a window where text takes 0.28) -- is given up: the host takes the sort / match
            // tiles for the pass (bit 31 of `dirty`; text settles in 4 rounds on average).  Only where the host says so: passes small
            // enough that the tiles cost little -- a run of 10 KiB of zeros inside a window takes 40 rounds and is no reason to send a
            // large pass there.
            if (STREAM && round_cap && round >= round_cap) {
                if (tid == 0) atomicOr(dirty, 0x80000000u);
                return;
            }
            PZ_CNT(c_rounds, 1);
#ifdef PZ_PROF
            const uint64_t c_tr0 = __builtin_readcyclecounter();
            const uint32_t c_l0 = c_loops;
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
                // A segment is parsed again only when the segment the path comes from is settled for the entry IT got: else
                // that one's exit -- this one's entry -- may still move (a run of one byte is entered 258 bytes further in
                // every round, and everything behind it would be parsed again in every one of them).
                tMark[m] = need ? 0 : 1;
                __syncthreads();
                const bool work = need && (m == me || tMark[tNxt[0][m]] != 0);  // (TAR-like level 6 13.4 -> 15.6 GB/s, text unchanged)
                deferred = need && !work;
#ifdef PZ_PROF
                c_tjump += __builtin_readcyclecounter() - c_tr0;
#endif
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
            uint32_t p = 0, q = 0, cnt = 0, crem = 0, lo = 1, best = 0, bdist = 0, maxlen = 0, pref = 0;
            uint32_t qh = PZ_NOHIT;
            // A walking lane holds what its candidate q needs to be judged: the link nqA = prv[q] and the two aligned
            // window dwords w0A, w1A that hold q's bytes number off .. off + 3 (xq = q + offb: the LDS byte address of
            // the first of them; offb = off + the LDS address of the window).  The burst below loads the same for the
            // link while q is judged, into nqB, w0B, w1B, and swaps the roles every step.
            uint32_t nqA = 0, w0A = 0, w1A = 0, xq = 0, offb = win_lds;
            // cnt = candidates the current call may still look at, 0 when the lane is not walking a chain
            // pref = the call's own bytes number off .. off + 3, off = best - 3: a candidate that is to beat `best`
            // agrees with all of them (SlidingWindow.zig:91-98 tests the last one)
#define PZ_SET_FILTER(OFF)                                                     \
    do {                                                                       \
        offb = (OFF) + win_lds;                                                \
        pref = pz_lds4(win32, p + (OFF));                                      \
    } while (0)
#define PZ_START_CALL(PP, LL, BUDGET)                                          \
    do {                                                                       \
        p = (PP);                                                              \
        best = (LL);                                                           \
        bdist = 0;                                                             \
        maxlen = min(NBr - p, (uint32_t)FL_MAX_MATCH);                         \
        if (STREAM && has_fl) maxlen = min(maxlen, fl_next_flush(fpl, nfl, wabs_r + p, 0xffffffffu) - (wabs_r + p)); \
        q = prv[p];                                                            \
        lo = p > FL_MAX_DIST ? p - FL_MAX_DIST : 1u;                           \
        if (STREAM && p >= zt) lo = max(lo, zlo);                              \
        cnt = (maxlen > best && q >= lo) ? (BUDGET) : 0u;                      \
        PZ_SET_FILTER(best ? best - 3u : 0u);                                  \
    } while (0)
#define PZ_LOAD_CAND()                                                         \
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
                PZ_START_CALL(a, 0u, chain);
                if (cnt != 0) PZ_LOAD_CAND();
            }
            for (;;) {
                PZ_CNT(c_loops, 1);
#ifdef PZ_PROF
                const uint64_t c_ta = __builtin_readcyclecounter();
#endif
                const uint64_t alive = __ballot(st != ST_DONE);
                if (alive == 0) break;
                const uint64_t waiting = __ballot(st == ST_WAIT);
                if (waiting == alive) __builtin_amdgcn_s_sleep(8);  // nothing to do but wait for another wave
                const uint64_t serve = alive & ~waiting;              // (a waiting lane is polled when the others are served)
                // ---- fast steps: one chain candidate per step (deflate.zig:248-263), rejected on the four bytes
                // that end at offset `best` (SlidingWindow.zig:91-98 tests one of them)
#pragma unroll 1
                for (int b = 0; b < PZ_BURST; b += PZ_UNROLL) {
                    const uint64_t mw = __ballot(cnt != 0);
                    if (mw == 0 || __popcll(serve & ~mw) >= PZ_NEED) break;
                    PZ_CNT(c_fast, 1);
                    PZ_CNT(c_walk, __popcll(mw));
#ifdef PZ_PROF
                    const uint64_t c_tb0 = __builtin_readcyclecounter();
                    const uint32_t c_nw = (uint32_t)__popcll(mw);
#endif
                    if (cnt != 0) {
                        // PZ_UNROLL steps written out by hand.  Per step and wave the serial chain is: the link of the
                        // candidate arrives -> its LDS address -> the load of ITS link; the candidate itself is judged
                        // while that load (and the load of the link's window dwords) is in flight, from registers that
                        // change roles every step (measured, profiles/r03_parse_experiments.txt: an instruction on the
                        // chain costs three times one in the shadow of the loads).  A lane leaves the burst -- its exec
                        // bit is cleared -- when its candidate passes the filter (qh, nqh, cnt keep what the measure
                        // needs) or its walk ends.
                        uint32_t nqB, w0B, w1B, a1, a2, xn, t, nqh = 0;
                        uint64_t s_save, s_t, s_hit;
#if defined(PZ_UNAL)  // (measured: the LDS of gfx950 serves unaligned dwords, bit-exact, but 25.9 ms against 23.2)
                        // experiment: ONE unaligned ds_read_b32 at the byte address (does the LDS of gfx950 serve it?)
                        asm volatile(
                            "s_mov_b64 %[ssave], exec\n\t"
                            "s_mov_b64 %[shit], 0\n\t"
                            "v_alignbyte_b32 v120, %[w1A], %[w0A], %[xq]\n\t"
                            ".rept " PZ_STR(PZ_HALF_UNROLL) "\n\t"
                            "v_lshl_add_u32 %[a1], %[nqA], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqB], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqA], %[offb]\n\t"
                            "ds_read_b32 v122, %[xn]\n\t"
                            "v_mov_b32 %[t], v120\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqA], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqA], %[lo]\n\t"
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqA]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
                            "v_lshl_add_u32 %[a1], %[nqB], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqA], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqB], %[offb]\n\t"
                            "ds_read_b32 v120, %[xn]\n\t"
                            "v_mov_b32 %[t], v122\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqB], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqB], %[lo]\n\t"
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqB]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
                            ".endr\n\t"
                            ".Lpz_done_%=:\n\t"
                            "s_mov_b64 %[st], exec\n\t"
                            "s_mov_b64 exec, %[ssave]\n\t"
                            "v_and_b32 %[a2], -4, %[xq]\n\t"
                            "ds_read2_b32 v[120:121], %[a2] offset1:1\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_mov_b32 %[w0A], v120\n\t"
                            "v_mov_b32 %[w1A], v121\n\t"
                            : [q] "+v"(q), [cnt] "+v"(cnt), [nqA] "+v"(nqA), [w0A] "+v"(w0A), [w1A] "+v"(w1A), [xq] "+v"(xq),
                              [qh] "+v"(qh), [nqh] "+v"(nqh), [nqB] "=&v"(nqB),
                              [a1] "=&v"(a1), [a2] "=&v"(a2), [xn] "=&v"(xn), [t] "=&v"(t), [ssave] "=&s"(s_save),
                              [st] "=&s"(s_t), [shit] "=&s"(s_hit)
                            : [lo] "v"(lo), [offb] "v"(offb), [pref] "v"(pref), [prvb] "s"(prv_lds)
                            : "vcc", "scc", "memory", "v120", "v121", "v122", "v123");
                        (void)w0B;
                        (void)w1B;
#elif defined(PZ_READ2)  // experiment: the two window dwords of a candidate in ONE LDS instruction: 23.19 -> 22.9-23.1 ms (noise)
                        // (ds_read2_b32 needs a register PAIR, which an asm operand cannot name half by half: v120..v123 are taken by hand)
                        asm volatile(
                            "s_mov_b64 %[ssave], exec\n\t"
                            "s_mov_b64 %[shit], 0\n\t"
                            "v_mov_b32 v120, %[w0A]\n\t"
                            "v_mov_b32 v121, %[w1A]\n\t"
                            ".rept " PZ_STR(PZ_HALF_UNROLL) "\n\t"
                            "v_lshl_add_u32 %[a1], %[nqA], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqB], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqA], %[offb]\n\t"
                            "v_and_b32 %[a2], -4, %[xn]\n\t"
                            "ds_read2_b32 v[122:123], %[a2] offset1:1\n\t"
                            "v_alignbyte_b32 %[t], v121, v120, %[xq]\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqA], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqA], %[lo]\n\t"
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqA]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
                            "v_lshl_add_u32 %[a1], %[nqB], 1, %[prvb]\n\t"
                            "ds_read_u16 %[nqA], %[a1]\n\t"
                            "v_add_u32 %[xn], %[nqB], %[offb]\n\t"
                            "v_and_b32 %[a2], -4, %[xn]\n\t"
                            "ds_read2_b32 v[120:121], %[a2] offset1:1\n\t"
                            "v_alignbyte_b32 %[t], v123, v122, %[xq]\n\t"
                            "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"
                            "v_cndmask_b32 %[qh], %[qh], %[q], vcc\n\t"
                            "v_cndmask_b32 %[nqh], %[nqh], %[nqB], vcc\n\t"
                            "s_and_b64 %[st], exec, vcc\n\t"
                            "s_or_b64 %[shit], %[shit], %[st]\n\t"
                            "s_andn2_b64 exec, exec, vcc\n\t"
                            "v_cmp_ge_u32 vcc, %[nqB], %[lo]\n\t"
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqB]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
                            ".endr\n\t"
                            ".Lpz_done_%=:\n\t"
                            "s_mov_b64 %[st], exec\n\t"
                            "s_mov_b64 exec, %[ssave]\n\t"
                            "v_mov_b32 %[w0A], v120\n\t"
                            "v_mov_b32 %[w1A], v121\n\t"
                            : [q] "+v"(q), [cnt] "+v"(cnt), [nqA] "+v"(nqA), [w0A] "+v"(w0A), [w1A] "+v"(w1A), [xq] "+v"(xq),
                              [qh] "+v"(qh), [nqh] "+v"(nqh), [nqB] "=&v"(nqB),
                              [a1] "=&v"(a1), [a2] "=&v"(a2), [xn] "=&v"(xn), [t] "=&v"(t), [ssave] "=&s"(s_save),
                              [st] "=&s"(s_t), [shit] "=&s"(s_hit)
                            : [lo] "v"(lo), [offb] "v"(offb), [pref] "v"(pref), [prvb] "s"(prv_lds)
                            : "vcc", "scc", "memory", "v120", "v121", "v122", "v123");
                        (void)w0B;
                        (void)w1B;
#elif PZ_ROT
                        // Registers change roles instead of being copied: the candidate, its link and the link's link rotate
                        // through q0, q1, q2 (period 3), the window dwords and byte addresses through the pairs a / b (period 2):
                        // six steps written out.  A lane that leaves (exec bit cleared) keeps its registers as they were in
                        // the step it left in: h0 / h1 / h2 remember in which role a hit found them; the budget counts down to
                        // a borrow (cb = cnt - 1), one instruction for decrement and test.  7 VALU instructions a step
                        // instead of 12.
#ifdef PZ_PIPE  // experiment: the next link is asked for as soon as this one is there, the window dwords may still be on their way
#define PZ_W1 "s_waitcnt lgkmcnt(2)\n\t"
#define PZ_W2 "s_waitcnt lgkmcnt(3)\n\t"
#define PZ_W3
#else
#define PZ_W1
#define PZ_W2
#define PZ_W3 "s_waitcnt lgkmcnt(0)\n\t"
#endif
#ifdef PZ_EARLY  // experiment: a burst ends after six or twelve steps when fewer than PZ_EARLY lanes still walk
#define PZ_EARLY_EXIT "s_bcnt1_i32_b64 vcc_lo, exec\n\ts_cmp_lt_u32 vcc_lo, " PZ_STR(PZ_EARLY) "\n\ts_cbranch_scc1 .Lpz_done_%=\n\t"
#else
#define PZ_EARLY_EXIT
#endif
#define PZ_RSTEP(QC, QN, QNN, WC0, WC1, WN0, WN1, XC, XN, H)                  \
    PZ_W1                                                                      \
    "v_lshl_add_u32 %[a1], %[" QN "], 1, %[prvb]\n\t"                          \
    "ds_read_u16 %[" QNN "], %[a1]\n\t"                                        \
    "v_add_u32 %[" XN "], %[" QN "], %[offb]\n\t"                              \
    "v_and_b32 %[a2], -4, %[" XN "]\n\t"                                       \
    "ds_read_b32 %[" WN0 "], %[a2]\n\t"                                        \
    "ds_read_b32 %[" WN1 "], %[a2] offset:4\n\t"                               \
    PZ_W2                                                                      \
    "v_alignbyte_b32 %[t], %[" WC1 "], %[" WC0 "], %[" XC "]\n\t"              \
    "v_cmp_eq_u32 vcc, %[t], %[pref]\n\t"                                      \
    "s_and_b64 %[st], exec, vcc\n\t"                                           \
    "s_or_b64 %[" H "], %[" H "], %[st]\n\t"                                   \
    "s_andn2_b64 exec, exec, vcc\n\t"                                          \
    "v_cmp_ge_u32 vcc, %[" QN "], %[lo]\n\t"                                   \
    "v_sub_co_u32_e64 %[cb], %[st], %[cb], 1\n\t"                              \
    "s_andn2_b64 vcc, vcc, %[st]\n\t"                                          \
    "s_and_b64 exec, exec, vcc\n\t"                                            \
    PZ_W3                                                                      \
    "s_cbranch_execz .Lpz_done_%=\n\t"
                        uint32_t q2r, xb, cb;
                        uint64_t s_h0, s_h1, s_h2;
                        asm volatile(
                            "s_mov_b64 %[ssave], exec\n\t"
                            "s_mov_b64 %[h0], 0\n\t"
                            "s_mov_b64 %[h1], 0\n\t"
                            "s_mov_b64 %[h2], 0\n\t"
                            "v_add_u32 %[cb], -1, %[cnt]\n\t"
                            ".rept " PZ_STR(PZ_ROT) "\n\t"
                            PZ_RSTEP("q0", "q1", "q2", "wa0", "wa1", "wb0", "wb1", "xa", "xb", "h0")
                            PZ_RSTEP("q1", "q2", "q0", "wb0", "wb1", "wa0", "wa1", "xb", "xa", "h1")
                            PZ_RSTEP("q2", "q0", "q1", "wa0", "wa1", "wb0", "wb1", "xa", "xb", "h2")
                            PZ_RSTEP("q0", "q1", "q2", "wb0", "wb1", "wa0", "wa1", "xb", "xa", "h0")
                            PZ_RSTEP("q1", "q2", "q0", "wa0", "wa1", "wb0", "wb1", "xa", "xb", "h1")
                            PZ_RSTEP("q2", "q0", "q1", "wb0", "wb1", "wa0", "wa1", "xb", "xa", "h2")
                            PZ_EARLY_EXIT  // (behind the six steps: a burst always makes some)
                            ".endr\n\t"
                            ".Lpz_done_%=:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_mov_b64 %[st], exec\n\t"
                            "s_mov_b64 exec, %[ssave]\n\t"
                            "v_add_u32 %[cnt], 1, %[cb]\n\t"
                            "v_cndmask_b32_e64 %[qh], %[qh], %[q0], %[h0]\n\t"
                            "v_cndmask_b32_e64 %[nqh], %[nqh], %[q1], %[h0]\n\t"
                            "v_cndmask_b32_e64 %[qh], %[qh], %[q1], %[h1]\n\t"
                            "v_cndmask_b32_e64 %[nqh], %[nqh], %[q2], %[h1]\n\t"
                            "v_cndmask_b32_e64 %[qh], %[qh], %[q2], %[h2]\n\t"
                            "v_cndmask_b32_e64 %[nqh], %[nqh], %[q0], %[h2]\n\t"
                            "s_or_b64 %[shit], %[h0], %[h1]\n\t"
                            "s_or_b64 %[shit], %[shit], %[h2]\n\t"
                            : [q0] "+v"(q), [cnt] "+v"(cnt), [q1] "+v"(nqA), [wa0] "+v"(w0A), [wa1] "+v"(w1A), [xa] "+v"(xq),
                              [qh] "+v"(qh), [nqh] "+v"(nqh), [q2] "=&v"(q2r), [wb0] "=&v"(w0B), [wb1] "=&v"(w1B), [xb] "=&v"(xb),
                              [cb] "=&v"(cb), [a1] "=&v"(a1), [a2] "=&v"(a2), [t] "=&v"(t), [ssave] "=&s"(s_save),
                              [st] "=&s"(s_t), [shit] "=&s"(s_hit), [h0] "=&s"(s_h0), [h1] "=&s"(s_h1), [h2] "=&s"(s_h2)
                            : [lo] "v"(lo), [offb] "v"(offb), [pref] "v"(pref), [prvb] "s"(prv_lds)
                            : "vcc", "scc", "memory");
                        (void)nqB;
                        (void)xn;
#undef PZ_RSTEP
#undef PZ_W1
#undef PZ_W2
#undef PZ_W3
#else
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
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqA]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
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
                            "v_add_u32 %[cnt], -1, %[cnt]\n\t"
                            "v_cmp_lt_i32_e64 %[st], 0, %[cnt]\n\t"
                            "s_and_b64 vcc, vcc, %[st]\n\t"
                            "s_and_b64 exec, exec, vcc\n\t"
                            "v_mov_b32 %[q], %[nqB]\n\t"
                            "v_mov_b32 %[xq], %[xn]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cbranch_execz .Lpz_done_%=\n\t"
                            ".endr\n\t"
                            ".Lpz_done_%=:\n\t"
                            "s_mov_b64 %[st], exec\n\t"
                            "s_mov_b64 exec, %[ssave]\n\t"
                            : [q] "+v"(q), [cnt] "+v"(cnt), [nqA] "+v"(nqA), [w0A] "+v"(w0A), [w1A] "+v"(w1A), [xq] "+v"(xq),
                              [qh] "+v"(qh), [nqh] "+v"(nqh), [nqB] "=&v"(nqB), [w0B] "=&v"(w0B), [w1B] "=&v"(w1B),
                              [a1] "=&v"(a1), [a2] "=&v"(a2), [xn] "=&v"(xn), [t] "=&v"(t), [ssave] "=&s"(s_save),
                              [st] "=&s"(s_t), [shit] "=&s"(s_hit)
                            : [lo] "v"(lo), [offb] "v"(offb), [pref] "v"(pref), [prvb] "s"(prv_lds)
                            : "vcc", "scc", "memory");
#endif
                        // s_t: the lanes that are still walking (their candidate's data is in the A registers again:
                        // an even number of steps); s_hit: those that stopped on a candidate that passes the filter
                        const uint32_t ln = threadIdx.x & 63u;
                        if (!((s_t >> ln) & 1ull)) {
                            if ((s_hit >> ln) & 1ull) {
                                crem = cnt - 1u;
                                q = nqh;  // (the walk goes on behind the candidate, at its link)
                            }
                            cnt = 0;
                        }
                    }
#ifdef PZ_PROF
                    {
                        const uint64_t dtb = __builtin_readcyclecounter() - c_tb0;
                        if (c_nw <= 2) c_tb_le2 += dtb; else if (c_nw <= 8) c_tb_le8 += dtb; else if (c_nw <= 24) c_tb_le24 += dtb; else c_tb_more += dtb;
                        if (c_nw <= 2) c_nb_le2++;
                    }
#endif
                }
                // ---- slow block: lanes that are not walking
#ifdef PZ_PROF
                const uint64_t c_tb = __builtin_readcyclecounter();
                c_tfast += c_tb - c_ta;
#endif
                PZ_CNT(c_slow, 1);
                PZ_CNT(c_measl, __popcll(__ballot(st != ST_DONE && qh != PZ_NOHIT)));
#ifdef PZ_TNEED
                const uint32_t n_walk_before = (uint32_t)__popcll(__ballot(cnt != 0));
#endif
                if (st != ST_DONE && cnt == 0) {
                    PZ_EV(0);
                    if (qh != PZ_NOHIT) {
                        PZ_EV(2);
                        // the candidate agrees where it must: its exact common prefix with p
                        uint32_t l = 0;
                        uint64_t x;
                        for (;;) {
                            PZ_EV(4);
                            PZ_CNT(c_meas, 1);
                            uint32_t a0, a1, b0, b1, a2, a3, b2, b3;
                            fl_lds_load8(win32, p + l, a0, a1);
                            fl_lds_load8(win32, qh + l, b0, b1);
                            fl_lds_load8(win32, p + l + 8u, a2, a3);
                            fl_lds_load8(win32, qh + l + 8u, b2, b3);
                            // (64-bit tests: all the dwords of 16 bytes are loaded together, one LDS round trip per 16 bytes -- round 6:
                            // 1.85 trips a compare at 8)
                            x = (uint64_t)(a0 ^ b0) | ((uint64_t)(a1 ^ b1) << 32);
                            const uint64_t y = (uint64_t)(a2 ^ b2) | ((uint64_t)(a3 ^ b3) << 32);
                            if (x || l + 8 >= maxlen) break;
                            l += 8;
                            x = y;
                            if (x || l + 8 >= maxlen) break;
                            l += 8;
                        }
                        // (the byte count of the last eight once, behind the loop, for all lanes together; eight equal bytes: ctz of 0
                        // would be undefined, bit 63 of ~0 stands for "8")
                        l = min(l + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u), maxlen);
                        cnt = q >= lo ? crem : 0u;            // the walk goes on behind the candidate ...
                        bool runhit = false;  // the candidate is the position before p and it is the best so far
                        if (l >= FL_MIN_MATCH && l > best) {  // deflate.zig:254-261
                            PZ_EV(6);
                            best = l;
                            bdist = p - qh;
                            runhit = bdist == 1u;
                            if (l >= nice || l >= maxlen) {
                                cnt = 0;  // ... unless the match is good enough / nothing longer is possible
                            } else {
                                PZ_SET_FILTER(l - 3u);
                            }
                        }
#if PZ_RUNSKIP
                        // The candidate right before p matched l bytes and the walk goes on (l < nice, l < maxlen): p starts a
                        // run of l bytes b that ends inside the lookahead, and so does every position below p - 1 as far down
                        // as the bytes are b -- each of them is the chain's next member (same four bytes, consecutive positions),
                        // matches exactly l bytes (its byte number l is still b, p's is not) and so changes nothing but the
                        // budget (deflate.zig:248-263).  They are taken together: k of them, as far as the run, the budget and
                        // the distance go; the walk goes on behind the lowest.  (Zero padding, sparse data, records: at level 6
                        // such a call walked its 128 candidates one LDS round trip at a time.)
                        if (cnt != 0 && runhit) {
                            const uint32_t c0 = p - 1u;  // (= the candidate just measured)
                            const uint32_t bp = (pz_lds4(win32, p) & 0xffu) * 0x01010101u;
                            const uint32_t fl_ = max(lo, c0 > cnt ? c0 - cnt : 0u);
                            uint32_t t = c0;
                            while (t > fl_) {
                                const uint32_t stp = min(8u, t - fl_);
                                uint32_t w0, w1;
                                fl_lds_load8(win32, t - stp, w0, w1);  // bytes t - stp .. t - stp + 7: the first stp of them count
                                uint64_t xx = (uint64_t)(w0 ^ bp) | ((uint64_t)(w1 ^ bp) << 32);
                                if (stp < 8u) xx &= (1ull << (8u * stp)) - 1ull;
                                if (xx) {
                                    t -= (uint32_t)(__builtin_clzll(xx) - (64 - 8 * (int)stp)) >> 3;  // the equal bytes right below t
                                    break;
                                }
                                t -= stp;
                            }
                            const uint32_t k = c0 - t;
                            if (k) {
                                q = prv[t];
                                cnt = q >= lo ? cnt - k : 0u;
                            }
                        }
#endif
                        qh = PZ_NOHIT;
                    }
#ifdef PZ_PROF
                    c_tmeas += __builtin_readcyclecounter() - c_tb;
#endif
#ifdef PZ_TNEED
                    // experiment: the automaton's moves wait until PZ_TNEED lanes ask for one (or fewer than PZ_WMIN lanes walk)
                    const bool doT = __popcll(__ballot(cnt == 0)) >= PZ_TNEED || n_walk_before + __popcll(__ballot(cnt != 0)) < PZ_WMIN;
#else
                    const bool doT = true;
#endif
                    if (cnt == 0 && doT) {
                        PZ_CNT(c_trans, 1);
                        PZ_EV(8);
                        // one move per lane and visit; every path through it ends in at most one new call
                        bool start = false;
                        uint32_t sp = 0, sl = 0, sb = chain;
                        if (st == ST_WAIT) {
                            PZ_EV(10);
                            // the lane before has finished its own parse: where does that leave this segment?
                            const uint32_t v = __hip_atomic_load(&tX[m - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (v != PZ_NONE) {
                                PZ_EV(12);
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
                            // the call has ended: the automaton's next move
                            bool emit = true;  // the pending match goes out (deflate.zig:182-184), or a literal
                            PZ_EV(14);
                            if (bdist) {       // a match, longer than the pending one if there is one
                                PZ_EV(16);
                                if (p != a) j++;  // the pending match's position becomes a literal (deflate.zig:166-168)
                                plen = best;
                                pdist = bdist;
                                emit = plen >= lazy;  // deflate.zig:171-173
                            }
                            if (emit) {
                                PZ_EV(18);
                                uint32_t desc = LITD, next = a + 1;
                                if (plen) {
                                    PZ_EV(20);
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
                                    PZ_EV(22);
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
                                PZ_EV(24);
                                // keep the match, look one position further (deflate.zig:174-178), in a quarter
                                // of the chain if the match is good enough (deflate.zig:241-245)
                                start = true;
                                sp = a + j + 1u;
                                sl = plen;
                                sb = plen >= good ? (chain >> 2) : chain;
                            }
                        }
                        if (start) {
                            PZ_EV(26);
                            PZ_START_CALL(sp, sl, sb);
                        }
                    }
                    if (cnt != 0) {
                        PZ_EV(28);
                        PZ_LOAD_CAND();
                    }  // (every lane that comes out of this block walking has a new q)
                }
#ifdef PZ_PROF
                c_ttrans += __builtin_readcyclecounter() - c_tb;
#endif
            }
#undef PZ_START_CALL
#undef PZ_SET_FILTER
#undef PZ_LOAD_CAND
#ifdef PZ_PROF
            if (round == 0) c_tspec += __builtin_readcyclecounter() - c_tr0; else c_tstitch += __builtin_readcyclecounter() - c_tr0;
            if (round == 0 && (tid & 63) == 0) atomicAdd((unsigned long long*)&g_fl_prof[64 + 16 * sub + (tid >> 6)], (unsigned long long)(__builtin_readcyclecounter() - c_tr0));
            if (round == 0 && (tid & 63) == 0) atomicAdd((unsigned long long*)&g_fl_prof[96 + 16 * sub + (tid >> 6)], (unsigned long long)(c_loops - c_l0));
#endif
            // Every segment parsed again in this round leaves where the round's path assumed: the path stands, and with
            // it the marks and entries found above -- no round to confirm it.
            if (round >= 1 && !__syncthreads_or(((fixing && res_exit != ex_used) || deferred) ? 1 : 0)) break;
        }
        // ---- the true anchors of this sub-pass
        if (m < nseg) {
            uint64_t T = 0;
            if (marked) {
                T = F;
                if (Z != PZ_NONE) T |= A & (~0ull << (Z - seg0));
            }
            // (OR into the bitmap the host has cleared: a segment need not start on a word boundary)
            const uint32_t pa = seg0 + r0, sh = pa & 31u;
            const uint64_t lo64 = T << sh;
            const uint32_t w0 = (uint32_t)lo64, w1 = (uint32_t)(lo64 >> 32), w2 = sh ? (uint32_t)(T >> (64u - sh)) : 0u;
            if (w0) atomicOr(&trueg[pa >> 5], w0);
            if (w1) atomicOr(&trueg[(pa >> 5) + 1], w1);
            if (w2) atomicOr(&trueg[(pa >> 5) + 2], w2);
        }
        __syncthreads();
        // the next sub-pass counts from its own r0
        if (tid == 0) {
            sh_exit = sh_next_entry + r0;
            if (sub == 0) sh_next_entry = sh_next_entry - (PZ_TA - FL_MAX_DIST - PZ_MARGIN);
        }
        if (STREAM) {
            __syncthreads();
            const uint32_t was = wexit[2 * c + sub];
            __syncthreads();
            if (tid == 0) wexit[2 * c + sub] = sh_exit;
            if (fix && was == sh_exit) return;  // from here on the first launch's anchors stand
        }
    }
    if (STREAM) {
        __syncthreads();
        carry = sh_exit - FL_MAX_DIST;  // (a window that is not the stream's last is left at or beyond 65274)
        if (wi + 1 == sw.nwin && tid == 0) {
            // the group's exit; in a fix launch: a NEW one (no sub-pass left where it left before): the group behind has to be
            // parsed again from it
            __hip_atomic_store(&gexit[blockIdx.x], sh_exit + FL_MAX_DIST * ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fix) atomicAdd(dirty, 1u);
        }
    }
#ifdef PZ_PROF
#pragma unroll
    for (int k = 0; k < 30; k++) {
        const uint32_t v = fl_wave_sum(c_ev[k]);
        if ((tid & 63) == 0) atomicAdd((unsigned long long*)&g_fl_prof[k], (unsigned long long)v);
    }
    if ((tid & 63) == 0) {
        // (c_meas / c_trans are per-lane counters of lane 0's view: only the wave-uniform ones are exact)
        atomicAdd((unsigned long long*)&g_fl_prof[40], (unsigned long long)c_fast);
        atomicAdd((unsigned long long*)&g_fl_prof[41], (unsigned long long)c_walk);
        atomicAdd((unsigned long long*)&g_fl_prof[42], (unsigned long long)c_slow);
        atomicAdd((unsigned long long*)&g_fl_prof[43], (unsigned long long)c_measl);
        atomicAdd((unsigned long long*)&g_fl_prof[44], (unsigned long long)c_loops);
        atomicAdd((unsigned long long*)&g_fl_prof[45], (unsigned long long)c_rounds);
        atomicAdd((unsigned long long*)&g_fl_prof[46], (unsigned long long)c_tspec);
        atomicAdd((unsigned long long*)&g_fl_prof[47], (unsigned long long)c_tstitch);
        atomicAdd((unsigned long long*)&g_fl_prof[48], (unsigned long long)(__builtin_readcyclecounter() - c_t0));
        atomicAdd((unsigned long long*)&g_fl_prof[49], 1ull);
        atomicAdd((unsigned long long*)&g_fl_prof[50], (unsigned long long)c_tfast);
        atomicAdd((unsigned long long*)&g_fl_prof[51], (unsigned long long)c_tmeas);
        atomicAdd((unsigned long long*)&g_fl_prof[52], (unsigned long long)c_ttrans);
        atomicAdd((unsigned long long*)&g_fl_prof[53], (unsigned long long)c_meas);
        atomicAdd((unsigned long long*)&g_fl_prof[54], (unsigned long long)c_trans);
        atomicAdd((unsigned long long*)&g_fl_prof[55], (unsigned long long)c_tstage);
        atomicAdd((unsigned long long*)&g_fl_prof[56], (unsigned long long)c_tjump);
        atomicAdd((unsigned long long*)&g_fl_prof[58], (unsigned long long)c_tb_le2);
        atomicAdd((unsigned long long*)&g_fl_prof[59], (unsigned long long)c_tb_le8);
        atomicAdd((unsigned long long*)&g_fl_prof[60], (unsigned long long)c_tb_le24);
        atomicAdd((unsigned long long*)&g_fl_prof[61], (unsigned long long)c_tb_more);
        atomicAdd((unsigned long long*)&g_fl_prof[62], (unsigned long long)c_nb_le2);
    }
    (void)c_meas; (void)c_trans; (void)c_transl;
#endif
    }  // windows
}

// ------------------------------------------------------------------ k_lz_emit
// Tokens of a chunk from its true anchors.  The chunk is handled in parts of 8192 positions; wave w
// owns positions [h0 + 512 w, h0 + 512 (w + 1)) of a part, 8 per lane; token offsets by DPP prefix
// sums, literals from the staged part, per-block histograms (LDS atomics), block cut at 32768
// tokens incl. the input slice a stored block would copy (deflate.zig:268-288: `rp` at the moment
// the 32768th token is added).
#define FL_EMITZ_THREADS 1024

__global__ __launch_bounds__(FL_EMITZ_THREADS, 8) void k_lz_emit(const uint8_t* __restrict__ in,
                                                                 const fl_chunk* __restrict__ chunks, fl_params prm,
                                                                 const uint32_t* __restrict__ desc_all,
                                                                 const uint32_t* __restrict__ true_all,
                                                                 uint32_t* __restrict__ tokens_all,
                                                                 uint32_t* __restrict__ hist_all,
                                                                 fl_block_plan* __restrict__ plans,
                                                                 uint32_t* __restrict__ ntok_all) {
    __shared__ uint32_t winp[FL_TOK_WIN_DW];
    __shared__ uint32_t hist[2][320];
    __shared__ uint32_t wtot[16];
    __shared__ uint16_t alist[16][FL_TOK_SPAN];  // per wave: the positions (in its span) of the span's anchors, ascending
    __shared__ uint32_t v1_sh, e1_sh;  // where the reference's window stands when token 32768 is added; where that token's bytes end
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fl_block_plan* plan0 = &plans[ck.first_block];
    fl_block_plan* plan1 = &plans[ck.first_block + 1];
    if (ck.skip) {
        if (tid == 0) {
            plan0->valid = 0;
            plan1->valid = 0;
            ntok_all[c] = 0;
        }
        return;
    }
    const uint32_t N = ck.in_len;
    const uint8_t* src = in + ck.in_off;
    const uint32_t* descg = desc_all + ck.pos_off;
    const uint32_t* trueg = true_all + (ck.pos_off >> 5);
    uint32_t* tokens = tokens_all + ck.pos_off;

    for (uint32_t i = tid; i < 640; i += FL_EMITZ_THREADS) (&hist[0][0])[i] = 0;
    if (tid == 0) v1_sh = e1_sh = N;
    uint32_t run0 = 0;  // tokens of the parts before this one (same value in every thread)
    uint32_t tw_next = 0;  // the wave's 16 words of anchor bits of the next part
    if (lane < FL_TOK_SPAN / 32u && wave * FL_TOK_SPAN + 32 * lane < min((uint32_t)FL_TOK_PART, N)) tw_next = trueg[((wave * FL_TOK_SPAN) >> 5) + lane];
    for (uint32_t h0 = 0; h0 < N; h0 += FL_TOK_PART) {
        const uint32_t span0 = h0 + wave * FL_TOK_SPAN;
        // the wave's 512 anchor bits (16 words); the positions of its anchors go to a list (a third of the positions
        // of text are anchors: the rounds below are over anchors, 64 at a time, not over positions)
        const uint32_t tw = tw_next;  // (requested one part ahead)
        {
            const uint32_t h0n = h0 + FL_TOK_PART, h1n = min(h0n + FL_TOK_PART, N), s0n = h0n + wave * FL_TOK_SPAN;
            tw_next = 0;
            if (h0n < N && lane < FL_TOK_SPAN / 32u && s0n + 32 * lane < h1n) tw_next = trueg[(s0n >> 5) + lane];
        }
        uint32_t na = 0;  // anchors of the span (wave-uniform)
#pragma unroll
        for (int r = 0; r < (int)FL_TOK_R; r++) {
            const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)tw, 2 * r);
            const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)tw, 2 * r + 1);
            const uint64_t m64 = (uint64_t)wlo | ((uint64_t)whi << 32);  // (no bit is set beyond the end of the input)
            // (the anchors below this lane: two v_mbcnt; the lane's own bit: the mask itself as the predicate -- the kernel is bound
            // by instruction issue, the shifts and masks of a 64-bit `below` were a third of this loop)
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
            if (__builtin_amdgcn_inverse_ballot_w64(m64)) alist[wave][na + below] = (uint16_t)(r * 64 + lane);
            na += (uint32_t)__popcll(m64);
        }
        fl_lds_order();
        // d[k]: descriptor of anchor number 64 k + lane (0 = none): PZ_DESC_LIT = one literal, else j literals and a match
        uint32_t d[FL_TOK_R];
#pragma unroll
        for (int k = 0; k < (int)FL_TOK_R; k++) {
            const uint32_t j = 64 * k + lane;
            d[k] = j < na ? descg[span0 + alist[wave][j]] : 0u;
        }
        // the part's bytes (+ lookahead for the literals of its last anchors), zero padded
        {
            const uint32_t nb = min(N - h0, FL_TOK_PART + FL_TOK_LOOK);
            // (all of a thread's loads first, then the stores: one round of memory latency, not three)
            constexpr uint32_t WR = (FL_TOK_WIN_DW + FL_EMITZ_THREADS - 1) / FL_EMITZ_THREADS;
            uint32_t wv[WR];
#pragma unroll
            for (uint32_t u = 0; u < WR; u++) {
                const uint32_t i = u * FL_EMITZ_THREADS + tid;
                wv[u] = 4 * i < nb ? fl_load_u32_clamped(src + h0, 4 * i, nb) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < WR; u++) {
                const uint32_t i = u * FL_EMITZ_THREADS + tid;
                if (i < FL_TOK_WIN_DW) winp[i] = wv[u];
            }
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < (int)FL_TOK_R; r++) cnt += d[r] ? ((d[r] & PZ_DESC_LIT) ? 1u : ((d[r] >> 23) & 0xff) + 1u) : 0u;
        cnt = fl_wave_sum(cnt);
        if (lane == 0) wtot[wave] = cnt;
        __syncthreads();
        uint32_t run = run0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) run += wtot[w];
            run0 += wtot[w];
        }
#pragma unroll 1  // (rolled: the descriptors rotate through d[0])
        for (uint32_t k0 = 0; k0 < na; k0 += 64) {
            const uint32_t j = k0 + lane;
            const uint32_t d0r = d[0];
#pragma unroll
            for (int k = 0; k + 1 < (int)FL_TOK_R; k++) d[k] = d[k + 1];
            d[FL_TOK_R - 1] = d0r;
            const bool mk = d0r != 0;  // (= j < na: every anchor has a descriptor)
            const uint32_t p = span0 + (mk ? (uint32_t)alist[wave][j] : 0u);
            const uint32_t q = p - h0;
            const uint32_t dd = (d0r & PZ_DESC_LIT) ? 0u : d0r;
            const uint32_t nl = mk ? (dd ? ((dd >> 23) & 0xff) : 1u) : 0u;  // literals of this anchor
            const uint32_t nt = mk ? (dd ? nl + 1 : 1u) : 0u;
            const uint32_t incl = fl_wave_incl_scan_dpp(nt);
            uint32_t idx = run + incl - nt;
            run += __builtin_amdgcn_readlane(incl, 63);
            for (uint32_t x = 0; x < nl; x++) {
                const uint32_t byte = fl_win_byte(winp, q + x);
                tokens[idx] = FL_TOK_LIT(byte);
                atomicAdd(&hist[idx >> 15][byte], 1u);
                if (idx == FL_MAX_TOKENS - 1) v1_sh = e1_sh = p + x + 1;  // emitted at the visit of the next position
                idx++;
            }
            if (mk && dd) {
                const uint32_t ll = (dd >> 15) & 0xff, d0 = dd & 0x7fff;
                tokens[idx] = (1u << 23) | (ll << 15) | d0;
                atomicAdd(&hist[idx >> 15][257 + fl_len_index(ll)], 1u);
                atomicAdd(&hist[idx >> 15][286 + fl_dist_code(d0)], 1u);
                // a match of at least `lazy` goes out at its own visit, a shorter one at the next
                // (deflate.zig:171-173 vs 182-184); rp at that moment decides the Q1 input slice
                if (idx == FL_MAX_TOKENS - 1) {
                    v1_sh = p + nl + ((ll + 3 >= prm.lazy) ? 0 : 1);
                    e1_sh = p + nl + ll + 3;  // (Q1: the window has not advanced over the match yet, deflate.zig:193)
                }
            }
        }
        __syncthreads();  // wtot and winp are reused by the next part
    }
    // block boundaries (deflate.zig:227-230, 268-288) and histograms
    const uint32_t total = run0;
    const uint32_t nblk = total >= FL_MAX_TOKENS ? 2 : 1;
    uint32_t* hg = hist_all + (uint64_t)ck.first_block * 320;
    for (uint32_t i = tid; i < 640; i += FL_EMITZ_THREADS)
        if (i < 320 * nblk) hg[i] = (&hist[0][0])[i];
    if (tid == 0) {
        ntok_all[c] = total;
        // the slice the block writer is handed with the first 32768 tokens ends at v1 (the reference: Q1) or where the tokens end
        const bool repair = (prm.flags & FL_PRM_REPAIR_Q1) != 0;
        const uint32_t v1 = repair ? e1_sh : v1_sh;
        plan0->no_input = 0;
        plan1->no_input = 0;
        plan0->q1_gap = (nblk == 2 && !repair) ? e1_sh - v1_sh : 0u;
        plan1->q1_gap = 0;
        if (nblk == 1) {
            plan0->valid = 1;
            plan0->tok_start = 0;
            plan0->tok_count = total;
            plan0->in_start = 0;
            plan0->in_len = N;
            plan0->final_block = 1;
            plan1->valid = 0;
        } else {
            plan0->valid = 1;
            plan0->tok_start = 0;
            plan0->tok_count = FL_MAX_TOKENS;
            plan0->in_start = 0;
            plan0->in_len = v1;
            plan0->final_block = 0;
            plan1->valid = 1;
            plan1->tok_start = FL_MAX_TOKENS;
            plan1->tok_count = total - FL_MAX_TOKENS;
            plan1->in_start = v1;
            plan1->in_len = N - v1;
            plan1->final_block = 1;
        }
    }
}
