// kernels_common.h -- device-side descriptors and wave-level helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>


#include "flate_layout.h"

#define FL_WAVE 64
// Phase timestamps (shader clock) of workgroup 0 of the tokenizer kernels: a debugging /
// tuning aid read back through flate_hip_debug_phase_cycles.  One s_memtime + one store by
// one thread per phase.
#define FL_PROF_SLOTS 160
__device__ uint64_t g_fl_prof[FL_PROF_SLOTS];
__device__ __forceinline__ void fl_prof_mark(uint32_t slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] = __builtin_readcyclecounter();
}

__device__ __forceinline__ uint32_t fl_lane() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
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
// inclusive prefix sum across the 64 lanes, on the cross-lane data path (DPP row shifts and row
// broadcasts: no LDS round trips)
__device__ __forceinline__ uint32_t fl_wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t fl_wave_incl_scan(uint32_t v, uint32_t lane) {
    (void)lane;
    return fl_wave_incl_scan_dpp(v);
}
__device__ __forceinline__ uint32_t fl_wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)fl_wave_incl_scan_dpp(v), 63);
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
