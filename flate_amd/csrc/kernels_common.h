// kernels_common.h -- device-side descriptors and wave-level helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flate_common.h"

#define FL_WAVE 64
#define FL_CHUNK_STRIDE 65536u  // per-chunk stride of the LZ scratch arrays
#define FL_BLOCK_BYTES 65535u   // SimpleCompressor buffer, deflate.zig:456

// One independent input chunk (= one output stream).
struct fl_chunk {
    uint64_t in_off;   // byte offset of the chunk in `in`
    uint64_t out_off;  // byte offset of the chunk's output slot in `out`
    uint64_t out_cap;  // slot size in bytes
    uint64_t pos_off;  // levels 4..9: base index of the chunk in the per-position scratch arrays
    uint32_t in_len;
    uint32_t first_block;  // index of the chunk's first fl_block_plan
    uint32_t n_blocks;     // huffman/store: in_len / 65535 + 1; levels 4..9: 2 slots (chunk), in_len / 32768 + 2 (stream)
    uint32_t skip;         // non-zero: chunk is not processed (status already set by the host)
    uint32_t piece0;       // whole-stream passes: index of the chunk's first piece in the pass
    uint32_t n_piece;      // whole-stream passes: number of pieces (runs between sync-flush points)
    uint32_t flush_off;    // whole-stream passes: the chunk's flush points in the pass's table ...
    uint32_t n_flush;      // ... and how many (ascending stream positions, each <= in_len)
    uint32_t zone_off;     // whole-stream passes: the chunk's entries in the slide table ...
    uint32_t n_slides;     // ... = how often the reference slides its window over this stream
    uint32_t unfinished;   // the stream ends with a sync-flush marker: no final block, no container footer
    uint32_t pad_;
};

// Whole-stream passes (levels 4..9; inputs longer than 65535 bytes, or any input with sync-flush
// points): one match-finder tile is a 64 KiB window of the stream whose positions >= tgt0 are
// searched ("targets"); the others are only history (SlidingWindow.zig:36-44 keeps 32 KiB of
// history across a slide).
struct fl_tile {
    uint32_t chunk;  // index into the pass's chunk table
    uint32_t w0;     // stream-relative position of the window start (multiple of 32768)
    uint32_t tgt0;   // window-relative position of the first target (0 or 32768)
    uint32_t zone;   // window-relative: targets at or beyond it are visited after the next slide (65536 = none)
};
// A piece = the stream positions between two sync-flush points (the whole stream when there are
// none): the lazy-matching automaton restarts at a flush (deflate.zig:196-203), and so do token
// numbering and blocks (deflate.zig:268-288).
struct fl_piece {
    uint32_t chunk;
    uint32_t start, end;   // stream positions [start, end)
    uint32_t first_block;  // plan slot of the piece's first block
    uint32_t n_blocks;     // slots: (end - start) / 32768 + 1 token blocks (+ 1 for a flush marker)
    uint32_t seg0, n_seg;  // its segments in the pass's segment table
    uint32_t flags;        // bit0: ends the stream (its last block is the final block); bit1: a sync-flush marker follows
};
// at most 32768 positions of one piece for the parse / emit kernels:
// [max(h0, piece.start), min(h0 + 32768, piece.end))
struct fl_seg {
    uint32_t piece;
    uint32_t h0;  // multiple of 32768
};

// call-wide constants
struct fl_params {
    uint32_t n_chunks;
    uint32_t n_blocks;
    int32_t container;  // 0 raw, 1 gzip, 2 zlib
    int32_t mode;       // 0 store, 1 huffman, 4..9
    // level args (deflate.zig:41-52)
    uint32_t good, lazy, nice, chain;
    uint32_t dbg;     // tuning experiments only (FLATE_HIP_DBG), 0 in production
    uint32_t stream;  // non-zero: whole-stream pass (kernels_stream.h)
};

// CRC-32 helper constants computed on the host once (reflected representation,
// x^0 = 0x80000000): xpow8[j] = x^(8 * 2^j) mod P, pow1024[m] = x^(8*1024*m) mod P.
struct fl_crc_consts {
    uint32_t xpow8[32];
    uint32_t pow1024[64];
    uint32_t pow65535;  // x^(8*65535)
};

// Phase timestamps (shader clock) of workgroup 0 of the tokenizer kernels: a debugging /
// tuning aid read back through flate_hip_debug_phase_cycles.  One s_memtime + one store by
// one thread per phase.
#define FL_PROF_SLOTS 64
__device__ uint64_t g_fl_prof[FL_PROF_SLOTS];
__device__ __forceinline__ void fl_prof_mark(uint32_t slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] = __builtin_readcyclecounter();
}

__device__ __forceinline__ uint32_t fl_lane() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t fl_wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t fl_wave_xor(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t fl_wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ uint32_t fl_wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}
// make this wave's LDS writes visible to its own later reads (cross-lane through LDS)
__device__ __forceinline__ void fl_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Order this wave's LDS accesses (cross-lane hand-off through LDS inside one wave).  The
// LDS unit executes one wave's DS instructions in issue order, so no wait is needed --
// only the compiler must not reorder across the hand-off.  Unlike fl_wave_lds_sync this
// does not drain outstanding global stores.
__device__ __forceinline__ void fl_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// little-endian 32-bit load from an arbitrarily aligned global address
// (the aligned address is derived with pointer arithmetic, not an integer round trip, so the
// compiler keeps the global address space and emits global_load rather than flat_load)
__device__ __forceinline__ uint32_t fl_load_u32_unaligned(const uint8_t* p) {
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
    const uint32_t* w = (const uint32_t*)(p - sh);
    const uint32_t lo = w[0];
    if (sh == 0) return lo;
    const uint32_t hi = w[1];
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// ---- CRC-32 (IEEE, reflected) polynomial arithmetic, as zlib's multmodp ----
#define FL_CRC_POLY 0xEDB88320u
__device__ __host__ inline uint32_t fl_crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1) ? ((b >> 1) ^ FL_CRC_POLY) : (b >> 1);
    }
    return p;
}
// x^(8*n) mod P using the table of x^(8*2^j)
__device__ __host__ inline uint32_t fl_crc_xpow8n(const uint32_t* xpow8, uint64_t n) {
    uint32_t p = 0x80000000u;
    for (int j = 0; n; j++, n >>= 1)
        if (n & 1) p = fl_crc_mulmod(xpow8[j], p);
    return p;
}

// ---- OR `nbits` (<= 64-ish, value already masked) bits into the output bit stream ----
__device__ __forceinline__ void fl_atomic_or_bits(uint32_t* out32, uint64_t bitpos, uint64_t v, uint32_t nbits) {
    if (nbits == 0) return;
    const uint64_t dw = bitpos >> 5;
    const uint32_t sh = (uint32_t)(bitpos & 31);
    const uint64_t a = v << sh;
    const uint32_t b = sh ? (uint32_t)(v >> (64 - sh)) : 0u;
    if ((uint32_t)a) atomicOr(&out32[dw], (uint32_t)a);
    if ((uint32_t)(a >> 32)) atomicOr(&out32[dw + 1], (uint32_t)(a >> 32));
    if (b) atomicOr(&out32[dw + 2], b);
}

// Copy n bytes from src (any alignment) to byte offset dst_byte of the output,
// cooperatively by `nthreads` threads (thread index tid).  Whole destination
// dwords are plain stores; the first/last partial dwords are atomic ORs into the
// pre-zeroed output (they may be shared with neighbouring blocks).
__device__ inline void fl_copy_bytes(uint32_t* out32, uint64_t dst_byte, const uint8_t* src, uint32_t n,
                                     uint32_t tid, uint32_t nthreads) {
    if (n == 0) return;
    const uint64_t dw0 = dst_byte >> 2;
    const uint64_t dw1 = (dst_byte + n + 3) >> 2;  // exclusive
    for (uint64_t dw = dw0 + tid; dw < dw1; dw += nthreads) {
        const int64_t first = (int64_t)(dw << 2) - (int64_t)dst_byte;  // src index of byte 0 of this dword
        uint32_t v = 0;
        bool full = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t si = first + k;
            if (si >= 0 && si < (int64_t)n)
                v |= (uint32_t)src[si] << (8 * k);
            else
                full = false;
        }
        if (full)
            out32[dw] = v;
        else if (v)
            atomicOr(&out32[dw], v);
    }
}
