// kernels_rank.h -- round 5: the reference's hash chain AND its candidate budget as a bound on positions.
//
// findMatch (deflate.zig:233-266) looks at the first `chain` members of the position's hash chain (a quarter of them
// when the match in hand has at least `good` bytes: deflate.zig:241-245) and at nothing beyond 32768 bytes.  The
// members of a chain are the earlier positions of the same 15-bit hash bucket (Lookup.zig:23-51), nearest first, so
// "among the first ch candidates" is the same as "at or above B_ch[p], the ch-th previous member of p's bucket":
//
//     looked at by the reference  <=>  q >= max(1, p - 32768, B_ch[p])
//
// -- ONE value per call instead of a count per step, which is what lets k_lz_parse6 (kernels_parse6.h) follow a
// sparser chain than the reference's and still stop where the reference stops.
//
// k_lz_rank, one workgroup per chunk:
//   R1  rank of every position in its bucket: RK[p] = number of earlier positions with the same hash -- DS_ADD_RTN on a
//       table of 16-bit counters in LDS, the waves taking turns block by block so that the table sees the positions
//       in ascending order (the scheme of k_lz_chain / k_lz_links<1>: lanes of one instruction are served in lane
//       order, one wave's instructions in program order);
//   R2  exclusive scan of the 32768 counters: base[h] = slot of the bucket's first member;
//   R3  slot[p] = base[h(p)] + RK[p] (all slots are computed before the table is overwritten);
//   R4  sorted[slot[p]] = p: the positions ordered by (bucket, position) -- a stable counting sort in LDS (128 KiB);
//   R5  L4[p] = sorted[slot - 1] (the reference's chain link, 0 = none: Lookup.zig:35-40), B_chain[p] = sorted[slot - chain],
//       B_quarter[p] = sorted[slot - chain / 4] (0 while the bucket has fewer earlier members).
// A rank served out of order shows up as a link above its position (sorted[slot - 1] > p); the chunk is then ranked
// again by one lane, one position at a time (never seen on gfx950; libflate_hip_slowchain.so always takes that path).
// It also recognises the chunk of ONE repeated byte (cflag 1: no chains; k_lz_parse6 writes its anchors directly).
//
// Bound: the turns of R1 (LDS atomics behind barriers), as k_lz_chain; R2-R5 are 5 unordered LDS accesses per position.
// No MFMA (integer ranks).
#pragma once
#include "kernels_walk.h"

#define RK_THREADS 1024
#define RK_WAVES (RK_THREADS / 64)

__global__ __launch_bounds__(RK_THREADS) void k_lz_rank(const uint8_t* __restrict__ in, const fl_chunk* __restrict__ chunks,
                                                        fl_params prm, uint16_t* __restrict__ prev_all,
                                                        uint32_t* __restrict__ bnd_all, uint32_t* __restrict__ cflag) {
    __shared__ uint32_t tab[32768];  // 128 KiB: R1-R3 the counters (two per word; words 16384.. are the dummies of positions past the end), R4-R5 sorted[]
    __shared__ uint32_t stg_all[RK_WAVES][FL_CHAIN_STG_DW];
    __shared__ uint32_t wsum[RK_WAVES];
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    if (ck.skip) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    uint32_t lane = tid & 63u;  // (not const: laundered at the top of the retry loop below, see there)
    uint32_t* sb = stg_all[wave];
    const uint32_t N = ck.in_len;
    const uint32_t Mpos = N >= 4 ? N - 3 : 0u;  // positions with 4 bytes left (Lookup.zig:24)
    if (tid == 0) cflag[c] = 0u;
    uint16_t* pv = prev_all + (uint64_t)c * FL_CHUNK_STRIDE;
    uint32_t* bnd = bnd_all + (uint64_t)c * FL_CHUNK_STRIDE;
    if (Mpos == 0) {
        for (uint32_t p = tid; p < N; p += RK_THREADS) bnd[p] = 0u;
        return;
    }
    const uint8_t* src = in + ck.in_off;
    const uint32_t sh = (uint32_t)((uintptr_t)src & 15);
    const uint4* src16 = (const uint4*)(src - sh);  // 16-byte granules; granule g covers chunk bytes 16 g - sh ..
    const uint32_t n_gran = (N + sh + 15) >> 4;
    // A chunk of ONE repeated byte needs no chains (k_lz_chain, kernels_parse.h).
    if (N >= 64) {
        const uint32_t b0 = src[0] * 0x01010101u;
        bool same = true;
        for (uint32_t g0 = 0; g0 < n_gran; g0 += RK_THREADS) {
            const uint32_t g = g0 + tid;
            if (g < n_gran) {
                uint4 v = src16[g];
                const int32_t first = (int32_t)(16 * g) - (int32_t)sh;
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
        if (same) {
            if (tid == 0) cflag[c] = 1u;
            return;
        }
    }
    {
        uint4* h4 = (uint4*)tab;
        for (uint32_t i = tid; i < 4096 + 16; i += RK_THREADS) h4[i] = make_uint4(0, 0, 0, 0);  // counters + dummies
    }
    __syncthreads();
    const uint32_t n_blocks = (Mpos + 1023) >> 10;  // <= 64: at most four per wave
    auto load_block = [&](uint32_t b, uint4& g0, uint4& g1) {
        const uint32_t ga = 64 * b + lane, gb = 64 * b + 64 + lane;
        g0 = (b < n_blocks && ga < n_gran) ? src16[ga] : make_uint4(0, 0, 0, 0);
        g1 = (b < n_blocks && lane < 2 && gb < n_gran) ? src16[gb] : make_uint4(0, 0, 0, 0);
    };
    // the hash of position (b << 10) + (s << 6) + lane from the wave's staged block
    auto hash_at = [&](uint32_t s) {
        const uint32_t off = (s << 6) + lane + sh;
        return fl_hash_le(__builtin_amdgcn_alignbyte(sb[(off >> 2) + 1], sb[off >> 2], off & 3));
    };
    // ---- R1: rk2[(16 k + s) / 2], half s & 1 = rank of position ((k * 16 + wave) << 10) + (s << 6) + lane; 0xffff: no such position
    uint32_t rk2[32];
    {
        uint4 ga0, ga1;
        load_block(wave, ga0, ga1);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (k * RK_WAVES < n_blocks) {  // (uniform: every wave meets every barrier)
                const uint32_t b = k * RK_WAVES + wave;
                ((uint4*)sb)[lane] = ga0;
                if (lane < 2) ((uint4*)sb)[64 + lane] = ga1;
                load_block(b + RK_WAVES, ga0, ga1);
                fl_lds_order();
                uint32_t hh[16];
                uint32_t val = 0;
#pragma unroll
                for (uint32_t s = 0; s < 16; s++) {
                    const uint32_t p = (b << 10) + (s << 6) + lane;
                    hh[s] = hash_at(s);
                    val |= (p < Mpos ? 1u : 0u) << s;
                }
                uint32_t old[16];
#pragma unroll 1
                for (uint32_t t = 0; t < RK_WAVES; t++) {
                    if (t == wave) {
#pragma unroll
                        for (uint32_t s = 0; s < 16; s++) {
                            const bool valid = (val >> s) & 1u;
                            old[s] = fl_lds_add_rtn(&tab[valid ? (hh[s] >> 1) : 16384u + lane], valid ? (1u << ((hh[s] & 1u) << 4)) : 0u);
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
                for (uint32_t s = 0; s < 16; s += 2) {
                    const uint32_t o0 = ((val >> s) & 1u) ? ((hh[s] & 1u) ? (old[s] >> 16) : (old[s] & 0xffffu)) : 0xffffu;
                    const uint32_t o1 = ((val >> (s + 1)) & 1u) ? ((hh[s + 1] & 1u) ? (old[s + 1] >> 16) : (old[s + 1] & 0xffffu)) : 0xffffu;
                    rk2[8 * k + (s >> 1)] = o0 | (o1 << 16);
                }
            } else {
#pragma unroll
                for (uint32_t s = 0; s < 8; s++) rk2[8 * k + s] = ~0u;
            }
        }
    }
    const uint32_t ch = prm.chain, chq = prm.chain >> 2;
    uint16_t* tab16 = (uint16_t*)tab;
    for (uint32_t attempt = 0;; attempt++) {
        // (The loop runs once; but as a loop it invites the compiler to compute every address of its body ahead of it -- and to
        // spill them all.  An opaque copy of the lane number keeps the address arithmetic where it is used.)
        asm volatile("" : "+v"(lane));
        // ---- R2: exclusive scan of the counters, bucket order = hash order (word h >> 1, half h & 1)
        {
            uint4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w4[u] = ((const uint4*)tab)[4 * tid + u];
            uint32_t w[16] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w,
                              w4[2].x, w4[2].y, w4[2].z, w4[2].w, w4[3].x, w4[3].y, w4[3].z, w4[3].w};
            uint32_t tot = 0;
#pragma unroll
            for (int u = 0; u < 16; u++) tot += (w[u] & 0xffffu) + (w[u] >> 16);
            const uint32_t incl = fl_wave_incl_scan_dpp(tot);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();  // (also: every thread has read its counters)
            uint32_t run = incl - tot;
            for (uint32_t x = 0; x < wave; x++) run += wsum[x];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const uint32_t lo16 = w[u] & 0xffffu, hi16 = w[u] >> 16;
                w[u] = run | ((run + lo16) << 16);  // (a prefix is at most 65532: it fits)
                run += lo16 + hi16;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) ((uint4*)tab)[4 * tid + u] = make_uint4(w[4 * u], w[4 * u + 1], w[4 * u + 2], w[4 * u + 3]);
        }
        __syncthreads();
        // ---- R3: the slot takes the rank's place (0xffff: no position); what the rank allows goes to three masks, bit 16 k + s:
        // a predecessor / chain / 4 of them / chain of them.  The hashes are made again from the input (L2): keeping them would
        // cost a register per position.
        uint64_t m1 = 0, mq = 0, mc = 0;
        {
            uint4 ga0, ga1;
            load_block(wave, ga0, ga1);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (k * RK_WAVES < n_blocks) {
                    const uint32_t b = k * RK_WAVES + wave;
                    ((uint4*)sb)[lane] = ga0;
                    if (lane < 2) ((uint4*)sb)[64 + lane] = ga1;
                    load_block(b + RK_WAVES, ga0, ga1);
                    fl_lds_order();
#pragma unroll
                    for (uint32_t s = 0; s < 16; s++) {
                        const uint32_t rk = (rk2[8 * k + (s >> 1)] >> ((s & 1u) << 4)) & 0xffffu;
                        if ((s & 3u) == 0u) asm volatile("" ::: "memory");
                        if (rk != 0xffffu) {
                            const uint32_t slot = (uint32_t)tab16[hash_at(s)] + rk;
                            const uint64_t bit = 1ull << (16 * k + s);
                            if (rk >= 1u) m1 |= bit;
                            if (rk >= chq) mq |= bit;
                            if (rk >= ch) mc |= bit;
                            rk2[8 * k + (s >> 1)] = (s & 1u) ? ((rk2[8 * k + (s >> 1)] & 0x0000ffffu) | (slot << 16))
                                                             : ((rk2[8 * k + (s >> 1)] & 0xffff0000u) | slot);
                        }
                    }
                    fl_lds_order();  // (the staging buffer is written again in the next step)
                }
            }
        }
        __syncthreads();  // the table has been read: it becomes sorted[]
        // ---- R4
#pragma unroll
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t p = (((j >> 4) * RK_WAVES + wave) << 10) + ((j & 15u) << 6) + lane;
            const uint32_t slot = (rk2[j >> 1] >> ((j & 1u) << 4)) & 0xffffu;
            if (slot != 0xffffu) tab16[slot] = (uint16_t)p;
        }
        __syncthreads();
        // ---- R5
        bool overtaken = false;
#pragma unroll
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t p = (((j >> 4) * RK_WAVES + wave) << 10) + ((j & 15u) << 6) + lane;
            const uint32_t slot = (rk2[j >> 1] >> ((j & 1u) << 4)) & 0xffffu;
            if ((j & 3u) == 0u) asm volatile("" ::: "memory");  // (a few gathers in flight at a time: more cost registers)
            if (slot != 0xffffu) {
                const uint32_t l4 = (m1 >> j) & 1ull ? tab16[slot - 1] : 0u;
                const uint32_t b2 = (mq >> j) & 1ull ? tab16[slot - chq] : 0u;
                const uint32_t b1 = (mc >> j) & 1ull ? tab16[slot - ch] : 0u;
                overtaken = overtaken || (((m1 >> j) & 1ull) && l4 >= p);  // (a predecessor lies below its position)
                pv[p] = (uint16_t)l4;
                bnd[p] = b1 | (b2 << 16);
            } else if (p < N) {
                bnd[p] = 0u;
            }
        }
#ifdef FL_CHAIN_FORCE_SLOW
        if (attempt == 0) overtaken = true;  // (test builds: exercise the fallback)
#endif
        if (!__syncthreads_or((overtaken && attempt == 0) ? 1 : 0)) break;
        // ---- never seen on gfx950: ranks by one lane, one position at a time (Lookup.zig:35-40 as written), then R2-R5 again
        for (uint32_t i = tid; i < 16384; i += RK_THREADS) tab[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            for (uint32_t p = 0; p < Mpos; p++) {
                const uint32_t h = fl_hash_le(fl_load_u32_clamped(src, p, N));
                bnd[p] = tab16[h];  // (scratch: overwritten in R5)
                tab16[h] = (uint16_t)(tab16[h] + 1u);
            }
            __threadfence_block();
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 64; j += 2) {
            uint32_t o[2];
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
                const uint32_t p = ((((j + u) >> 4) * RK_WAVES + wave) << 10) + (((j + u) & 15u) << 6) + lane;
                o[u] = p < Mpos ? (__hip_atomic_load(&bnd[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffu) : 0xffffu;
            }
            rk2[j >> 1] = o[0] | (o[1] << 16);
        }
        __syncthreads();
    }
}
