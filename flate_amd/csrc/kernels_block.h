// kernels_block.h -- the Huffman block writer on the GPU: symbol histograms,
// block planning (one lane per block runs the serial planner of flate_common.h),
// output bit-offset scan, and the wave-parallel bit packer.
//
// Reference path: BlockWriter.write / huffmanBlock / storedBlock
// (block_writer.zig:307-388, 524-585), SimpleCompressor (deflate.zig:449-529),
// container header/footer + hasher (container.zig:53-109, 168-206).
//
// Bound: HBM (read every input byte / token once, write every output byte once);
// the bit packer stages bits in LDS and emits whole dwords, coalesced.
#pragma once
#include "kernels_common.h"

// block b of a huffman-only / store-only pass: from the host table when there is one, else
// block j = b - first_block of its chunk covers bytes [65535 j, 65535 (j+1)) (deflate.zig:498-511)
struct fl_sb {
    uint32_t start, len;
    bool final_block, marker;
};
__device__ __forceinline__ fl_sb fl_simple_block(const fl_chunk& ck, const fl_sblock* __restrict__ sblocks, uint32_t b) {
    fl_sb r;
    if (sblocks) {
        const fl_sblock e = sblocks[b];
        r.start = e.start;
        r.len = e.len;
        r.final_block = (e.flags & 1) != 0;
        r.marker = (e.flags & 2) != 0;
    } else {
        const uint32_t j = b - ck.first_block;
        const uint64_t s = (uint64_t)j * FL_BLOCK_BYTES;
        r.start = (uint32_t)min(s, (uint64_t)ck.in_len);
        r.len = min(FL_BLOCK_BYTES, ck.in_len - r.start);
        r.final_block = j + 1 == ck.n_blocks;
        r.marker = false;
    }
    return r;
}
// the empty stored block a sync flush appends (deflate.zig:474-478, 276-278)
__device__ __forceinline__ void fl_plan_marker(fl_block_plan* plan) {
    plan->valid = 2;
    plan->type = FL_BLOCK_STORED;
    plan->size_bits = 0;
    plan->hdr_nbits = 0;
    plan->final_block = 0;
    plan->in_start = 0;
    plan->in_len = 0;
    plan->tok_start = 0;
    plan->tok_count = 0;
    plan->no_input = 0;
    plan->q1_gap = 0;
}

// ------------------------------------------------------------------ histograms
// huffman-only mode: 256-bin byte histogram of each 65535-byte block
// (block_writer.zig:575-585).  One workgroup (256 threads) per block, 16-byte loads.  Every LANE has a histogram of its own
// (64 copies, shared by the four waves' lanes of one number): 16-bit counters, two to a word (a copy sees 4 x 256 bytes at most),
// the words of a copy swizzled by the copy's number -- lanes that count the same byte (text: a space, an e; padding: all of
// them) hit 32 different banks, not one counter.  (Round 4: 8 copies per wave of 32-bit counters: an LDS atomic took 47 cycles
// on text, the kernel 0.16 ms of config #4's 0.78.)
#define FL_HIST_WORD(cp, bin) ((cp) * 128u + ((((bin) >> 1) ^ (cp)) & 127u))
__global__ __launch_bounds__(256) void k_byte_hist(const uint8_t* __restrict__ in,
                                                   const fl_chunk* __restrict__ chunks,
                                                   const uint32_t* __restrict__ blk_chunk,
                                                   const fl_sblock* __restrict__ sblocks,
                                                   uint32_t* __restrict__ hist /* [n_blocks][320] */) {
    __shared__ uint32_t sh[64 * 128];
    const uint32_t b = blockIdx.x;
    const fl_chunk ck = chunks[blk_chunk[b]];
    const uint32_t tid = threadIdx.x;
    const uint32_t cp = tid & 63u;
    for (uint32_t i = tid; i < 64 * 128; i += 256) sh[i] = 0;
    __syncthreads();
    bool wide = false;  // a block of more than 65535 bytes (never made: the counters would not hold it)
    if (!ck.skip) {
        const fl_sb sb = fl_simple_block(ck, sblocks, b);
        const uint32_t len = sb.len;
        const uint8_t* src = in + ck.in_off + sb.start;
        wide = len > 65535u;
        auto count = [&](uint32_t byte) { atomicAdd(&sh[FL_HIST_WORD(cp, byte)], 1u << (16u * (byte & 1u))); };
        // head bytes up to 16-byte alignment, then 16-byte loads
        const uint32_t mis = (uint32_t)((16 - ((uintptr_t)src & 15)) & 15);
        const uint32_t head = mis < len ? mis : len;
        if (!wide) {
            if (tid < head) count(src[tid]);
            const uint32_t body = (len - head) >> 4;
            const uint4* src16 = (const uint4*)(src + head);
            for (uint32_t i = tid; i < body; i += 256) {
                const uint4 v = src16[i];
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = w4[k];
                    count(w & 0xff);
                    count((w >> 8) & 0xff);
                    count((w >> 16) & 0xff);
                    count(w >> 24);
                }
            }
            const uint32_t tail0 = head + (body << 4);
            if (tail0 + tid < len) count(src[tail0 + tid]);
        }
    }
    __syncthreads();
    uint32_t v = 0;
#pragma unroll 8
    for (uint32_t k = 0; k < 64; k++) v += (sh[FL_HIST_WORD(k, tid)] >> (16u * (tid & 1u))) & 0xffffu;
    if (wide) {  // (one byte at a time, 32-bit counters: what the kernel was before it had any copies)
        __syncthreads();
        sh[tid] = 0;
        __syncthreads();
        const fl_sb sb = fl_simple_block(ck, sblocks, b);
        const uint8_t* src = in + ck.in_off + sb.start;
        for (uint32_t i = tid; i < sb.len; i += 256) atomicAdd(&sh[src[i]], 1u);
        __syncthreads();
        v = sh[tid];
    }
    hist[(uint64_t)b * 320 + tid] = v;
}

// debug seam (flate_hip_debug_write_block): histogram of a caller-supplied token list, as the
// tokenizer's emit kernels build it (block_writer.zig:444-462)
__global__ __launch_bounds__(256) void k_dbg_token_hist(const uint32_t* __restrict__ tokens, uint32_t n,
                                                        uint32_t* __restrict__ hist /* [320] */) {
    __shared__ uint32_t sh[320];
    for (uint32_t i = threadIdx.x; i < 320; i += 256) sh[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t t = tokens[i];
        if (FL_TOK_IS_MATCH(t)) {
            atomicAdd(&sh[257 + fl_len_index(FL_TOK_LENLIT(t))], 1u);
            atomicAdd(&sh[286 + fl_dist_code(FL_TOK_DIST0(t))], 1u);
        } else {
            atomicAdd(&sh[FL_TOK_LENLIT(t)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 320; i += 256) hist[i] = sh[i];
}

// ------------------------------------------------------------------ checksums
// Per-block CRC-32 (gzip) or Adler-32 partial sums (zlib) of the raw input
// (deflate.zig:314,507 -> container.zig:168-206).  One wave per 65535-byte block,
// lane i owns bytes [1024 i, 1024 (i+1)); partial CRCs are folded with
// crc(A||B) = crc(A) * x^(8|B|) + crc(B).
__global__ __launch_bounds__(64) void k_checksum(const uint8_t* __restrict__ in,
                                                 const fl_chunk* __restrict__ chunks,
                                                 const uint32_t* __restrict__ blk_chunk,
                                                 const fl_sblock* __restrict__ sblocks, fl_params prm,
                                                 fl_crc_consts cc, uint32_t* __restrict__ part /* [n_blocks][2] */) {
    __shared__ uint32_t tab[4][256];
    const uint32_t b = blockIdx.x;
    const fl_chunk ck = chunks[blk_chunk[b]];
    const uint32_t lane = threadIdx.x;
    if (ck.skip) return;
    uint32_t start, len;
    if (prm.mode >= 4 && !prm.stream) {
        if (b != ck.first_block) return;  // level 4..9: the chunk (<= 65535 bytes) is one checksum unit
        start = 0;
        len = ck.in_len;
    } else {
        const fl_sb sb = fl_simple_block(ck, prm.mode < 4 ? sblocks : nullptr, b);
        start = sb.start;
        len = sb.len;
    }
    const uint8_t* src = in + ck.in_off + start;
    const uint32_t lo = min(len, lane * 1024u), hi = min(len, lane * 1024u + 1024u);
    if (prm.container == 1) {
        for (uint32_t t = lane; t < 256; t += 64) {
            uint32_t c = t;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (FL_CRC_POLY ^ (c >> 1)) : (c >> 1);
            tab[0][t] = c;
        }
        fl_wave_lds_sync();
        for (int k = 1; k < 4; k++) {
            for (uint32_t t = lane; t < 256; t += 64) {
                const uint32_t c = tab[k - 1][t];
                tab[k][t] = tab[0][c & 0xff] ^ (c >> 8);
            }
            fl_wave_lds_sync();
        }
        uint32_t c = 0xffffffffu;
        uint32_t i = lo;
        // (16 bytes per load, the next ones requested before these are folded: the lanes' slices are 1 KiB apart, every load
        // instruction touches 64 cache lines -- with one dword per load the kernel took 1.07 ms for 2705 chunks)
        while (i < hi && (((uintptr_t)(src + i)) & 15)) c = tab[0][(c ^ src[i++]) & 0xff] ^ (c >> 8);
        if (i + 16 <= hi) {
            uint4 v = *(const uint4*)(src + i);
            for (; i + 16 <= hi; i += 16) {
                const uint4 cur = v;
                if (i + 32 <= hi) v = *(const uint4*)(src + i + 16);
                c ^= cur.x;
                c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
                c ^= cur.y;
                c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
                c ^= cur.z;
                c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
                c ^= cur.w;
                c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
            }
        }
        for (; i < hi; i++) c = tab[0][(c ^ src[i]) & 0xff] ^ (c >> 8);
        c = (hi > lo) ? ~c : 0u;  // crc of an empty slice is 0
        // bytes of the block after this lane's slice
        const uint32_t after = len - hi;
        const uint32_t full = after >> 10, tail = after & 1023u;
        uint32_t tpow = 0x80000000u;
        for (int j = 0; j < 10; j++)
            if (tail & (1u << j)) tpow = fl_crc_mulmod(cc.xpow8[j], tpow);
        c = fl_crc_mulmod(fl_crc_mulmod(c, cc.pow1024[full]), tpow);
        c = fl_wave_xor(c);
        if (lane == 0) {
            part[2 * (uint64_t)b] = c;
            part[2 * (uint64_t)b + 1] = len;
        }
    } else if (prm.container == 2) {
        // Adler-32 pieces with a = b = 0 start: A = sum d_k, B = sum (n - k) d_k
        uint32_t A = 0, B = 0;
        uint32_t i = lo;  // <= 1024 bytes: B < 2^28, no overflow
        for (; i < hi && (((uintptr_t)(src + i)) & 15); i++) {
            A += src[i];
            B += A;
        }
        for (; i + 16 <= hi; i += 16) {  // (16 bytes per load: see the CRC above)
            const uint4 v = *(const uint4*)(src + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int bb = 0; bb < 4; bb++) {
                    A += (w[k] >> (8 * bb)) & 0xffu;
                    B += A;
                }
            }
        }
        for (; i < hi; i++) {
            A += src[i];
            B += A;
        }
        // combine lanes in order: B_tot = sum_i (B_i + A_i * bytes_after_i), all mod 65521
        const uint32_t after = len - hi;
        uint64_t bb = (uint64_t)B + (uint64_t)A * after;
        uint32_t Bm = (uint32_t)(bb % 65521u);
        uint32_t Am = fl_wave_sum(A) % 65521u;  // <= 65535*255 fits
        // sum of 64 values < 65521 fits in 32 bits
        Bm = fl_wave_sum(Bm) % 65521u;
        if (lane == 0) {
            part[2 * (uint64_t)b] = Am | (Bm << 16);
            part[2 * (uint64_t)b + 1] = len;
        }
    }
}

// ------------------------------------------------------------------ planning
// One wave per block, four blocks per workgroup (more resident waves per CU than one-wave
// workgroups get); lane 0 of each wave runs the serial planner (flate_common.h) with its
// scratch in LDS.  mode 1: huffmanBlock; mode >= 4: BlockWriter.write.
#define FL_PLAN_WAVES 4
__global__ __launch_bounds__(64 * FL_PLAN_WAVES) void k_plan(const fl_chunk* __restrict__ chunks,
                                                             const uint32_t* __restrict__ blk_chunk,
                                                             const fl_sblock* __restrict__ sblocks, fl_params prm,
                                                             const uint32_t* __restrict__ hist,
                                                             fl_block_plan* __restrict__ plans) {
    __shared__ fl_plan_ws wss[FL_PLAN_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t b = blockIdx.x * FL_PLAN_WAVES + wave;
    if (b >= prm.n_blocks) return;
    if (prm.mode >= 4 && !prm.stream) {
        // two plan slots per chunk and the second one is rarely used: visit all first slots
        // before the second ones, so that resident waves are waves with work
        const uint32_t half = prm.n_blocks >> 1;
        b = b < half ? 2 * b : 2 * (b - half) + 1;
    }
    fl_plan_ws& ws = wss[wave];
    fl_block_plan* plan = &plans[b];
    if (prm.mode == 1) {
        const fl_chunk ck = chunks[blk_chunk[b]];
        if (ck.skip) {
            if (lane == 0) plan->valid = 0;
            return;
        }
        const fl_sb sb = fl_simple_block(ck, sblocks, b);
        if (sb.marker) {
            if (lane == 0) fl_plan_marker(plan);
            return;
        }
        for (uint32_t i = lane; i < 256; i += 64) ws.lit_freq[i] = (uint16_t)hist[(uint64_t)b * 320 + i];
        fl_wave_lds_sync();
        // all lanes run the planner in lock step (flate_common.h: only its sorts are lane-parallel)
        plan->valid = 1;
        plan->in_start = sb.start;
        plan->tok_start = sb.start;
        plan->tok_count = sb.len;
        fl_plan_huffman_block(&ws, plan, sb.len, sb.final_block);
    } else {
        // token block: metadata (valid, tok_*, in_*, final_block) was written by the emit kernel,
        // valid = 0 for the slots of skipped chunks
        if (plan->valid != 1) return;  // 0: unused slot; 2: a sync-flush marker, already a stored block
        for (uint32_t i = lane; i < FL_NUM_LIT; i += 64) ws.lit_freq[i] = (uint16_t)hist[(uint64_t)b * 320 + i];
        if (lane < FL_NUM_DIST) ws.dist_freq[lane] = (uint16_t)hist[(uint64_t)b * 320 + 286 + lane];
        fl_wave_lds_sync();
        // in_len == FL_NO_INPUT: a window slide since the previous flush took the raw bytes away
        // (SlidingWindow.zig:119-123); only whole-stream passes ever set it
        fl_plan_token_block(&ws, plan, plan->in_len, plan->final_block, prm.plan_dynamic_only != 0);  // all lanes, in lock step
    }
}

// store-only mode: every block is a stored block (deflate.zig:486-493)
__global__ __launch_bounds__(256) void k_plan_store(const fl_chunk* __restrict__ chunks,
                                                    const uint32_t* __restrict__ blk_chunk,
                                                    const fl_sblock* __restrict__ sblocks, uint32_t n_blocks,
                                                    fl_block_plan* __restrict__ plans) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const fl_chunk ck = chunks[blk_chunk[b]];
    fl_block_plan* plan = &plans[b];
    const fl_sb sb = fl_simple_block(ck, sblocks, b);
    plan->valid = ck.skip ? 0 : 1;
    plan->type = FL_BLOCK_STORED;
    plan->size_bits = 0;
    plan->hdr_nbits = 0;
    plan->final_block = sb.final_block ? 1 : 0;
    plan->in_start = sb.start;
    plan->in_len = sb.len;
    plan->tok_start = sb.start;
    plan->tok_count = sb.len;
    plan->no_input = 0;
    plan->q1_gap = 0;
}

// Fold the checksum parts of n_blocks consecutive blocks (k_checksum: CRC-32 of each block / Adler-32
// sums with a = b = 0 start, and the block lengths) into the checksum of their concatenation; all 64
// lanes call it, the result is valid on every lane.  Every lane folds a contiguous run of blocks, then
// the lanes' results are combined in order with the same associative rule (crc(A||B) = crc(A) x^(8|B|)
// + crc(B); Adler-32: A = A1 + A2, B = B1 + |B| A1 + B2, the leading 1 added at the end).
__device__ __forceinline__ uint32_t fl_fold_checksums(const uint32_t* __restrict__ cks_part, uint32_t first_block,
                                                      uint32_t n_blocks, int container, const fl_crc_consts& cc,
                                                      uint32_t lane) {
    const uint32_t per = (n_blocks + 63) / 64;
    const uint32_t j0 = min(lane * per, n_blocks), j1 = min(j0 + per, n_blocks);
    uint32_t x = 0, y = 0;  // crc | (A, B) of this lane's run
    uint64_t len = 0;       // bytes of this lane's run
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t pc = cks_part[2 * (uint64_t)(first_block + j)];
        const uint32_t pl = cks_part[2 * (uint64_t)(first_block + j) + 1];
        if (container == 1) {
            const uint32_t sh = pl == FL_BLOCK_BYTES ? cc.pow65535 : fl_crc_xpow8n(cc.xpow8, pl);
            x = fl_crc_mulmod(x, sh) ^ pc;
        } else {
            y = (uint32_t)(((uint64_t)y + (uint64_t)x * pl + (pc >> 16)) % 65521u);
            x = (x + (pc & 0xffff)) % 65521u;
        }
        len += pl;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        // this lane = (lanes below, already folded) followed by its own run
        const uint32_t ox = __shfl_up(x, d, 64), oy = __shfl_up(y, d, 64);
        const uint64_t ol = __shfl_up(len, d, 64);
        if (lane >= (uint32_t)d) {
            if (container == 1) {
                x = fl_crc_mulmod(ox, fl_crc_xpow8n(cc.xpow8, len)) ^ x;
            } else {
                y = (uint32_t)(((uint64_t)oy + (uint64_t)ox * (len % 65521u) + y) % 65521u);
                x = (ox + x) % 65521u;
            }
            len += ol;
        }
    }
    // lane 63 holds the fold over all blocks
    const uint32_t fx = __shfl(x, 63, 64), fy = __shfl(y, 63, 64);
    const uint64_t n = __shfl(len, 63, 64);
    if (container == 1) return fx;
    return ((1u + fx) % 65521u) | ((uint32_t)((n % 65521u + fy) % 65521u) << 16);  // a = 1 + A, b = n + B
}

// debug / wrapper seam: checksum of one buffer cut into 65535-byte units (flate_hip_checksum)
__global__ __launch_bounds__(64) void k_fold_checksum(const uint32_t* __restrict__ cks_part, uint32_t n_blocks,
                                                      int container, fl_crc_consts cc, uint32_t* __restrict__ out) {
    const uint32_t v = fl_fold_checksums(cks_part, 0, n_blocks, container, cc, threadIdx.x);
    if (threadIdx.x == 0) out[0] = v;
}

// ------------------------------------------------------------------ offsets
// Bit offset of every block inside its chunk's stream.  A Huffman block moves the
// offset by its exact size; a stored block first pads to a byte boundary
// (block_writer.zig:283-291).  Both are maps off -> (has ? ceil8(off + a) + c : off + a);
// they compose associatively, so a chunk with thousands of blocks is a wave scan.
struct fl_offmap {
    uint64_t a, c;
    uint32_t has;
};
__device__ __forceinline__ fl_offmap fl_offmap_compose(fl_offmap f, fl_offmap g) {  // f first, then g
    fl_offmap r;
    if (!f.has) {
        r.a = f.a + g.a;
        r.c = g.c;
        r.has = g.has;
    } else if (!g.has) {
        r.a = f.a;
        r.c = f.c + g.a;
        r.has = 1;
    } else {
        r.a = f.a;
        r.c = ((f.c + g.a + 7) & ~7ull) + g.c;
        r.has = 1;
    }
    return r;
}
__device__ __forceinline__ uint64_t fl_offmap_apply(fl_offmap f, uint64_t off) {
    return f.has ? (((off + f.a + 7) & ~7ull) + f.c) : off + f.a;
}

// One workgroup per chunk: block offsets, container header / footer bytes, out_len, status.  A chunk of the chunk path has two
// blocks and gets one wave; a long huffman-only / store-only stream has thousands (config #4: 2049) and gets 16 -- every wave a
// contiguous range of the blocks: its composed map first, the waves' maps composed in order through LDS, then the range again
// with what lies before it applied (one wave over 2049 blocks: 0.19 ms of a 1.2 ms step, most of it the checksum fold: 33
// blocks per lane, a 32-step polynomial product each).
#define FL_OFFS_MAX_WAVES 16
__global__ __launch_bounds__(64 * FL_OFFS_MAX_WAVES) void k_offsets(const fl_chunk* __restrict__ chunks, fl_params prm,
                                                fl_crc_consts cc, fl_block_plan* __restrict__ plans,
                                                const uint32_t* __restrict__ cks_part,
                                                uint8_t* __restrict__ out, uint64_t* __restrict__ out_len,
                                                int32_t* __restrict__ status) {
    __shared__ fl_offmap wtot[FL_OFFS_MAX_WAVES];
    __shared__ uint32_t wx[FL_OFFS_MAX_WAVES], wy[FL_OFFS_MAX_WAVES];
    __shared__ uint64_t wl[FL_OFFS_MAX_WAVES];
    __shared__ uint32_t q1_sh;  // a block boundary of this chunk loses or repeats bytes (quirk Q1)
    const uint32_t c = blockIdx.x;
    const fl_chunk ck = chunks[c];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    if (ck.skip) return;  // host already wrote status / out_len
    if (threadIdx.x == 0) q1_sh = 0;
    __syncthreads();
    const uint32_t hdr_bytes = prm.container == 1 ? 10u : (prm.container == 2 ? 2u : 0u);
    const uint32_t ftr_bytes = ck.unfinished ? 0u : (prm.container == 1 ? 8u : (prm.container == 2 ? 4u : 0u));
    const uint64_t base = (ck.out_off + hdr_bytes) * 8;
    // this wave's blocks: [r0, r1), a multiple of 64 per wave
    const uint32_t R = (((ck.n_blocks + W - 1) / W) + 63u) & ~63u;
    const uint32_t r0 = min(wave * R, ck.n_blocks), r1 = min(r0 + R, ck.n_blocks);

    fl_offmap ident;
    ident.a = 0;
    ident.c = 0;
    ident.has = 0;
    // the composition of the range's blocks behind `pre`; with `write`, every block gets its bit offset on the way
    auto pass = [&](fl_offmap pre, bool write) {
        fl_offmap run = pre;
        for (uint32_t b0 = r0; b0 < r1; b0 += 64) {
            const uint32_t j = b0 + lane;
            fl_offmap m = ident;
            fl_block_plan* plan = nullptr;
            if (j < r1) {
                plan = &plans[ck.first_block + j];
                if (plan->valid) {
                    if (plan->type == FL_BLOCK_STORED) {
                        m.a = 3;
                        m.c = 32 + 8ull * plan->in_len;
                        m.has = 1;
                    } else {
                        m.a = plan->size_bits;
                    }
                }
            }
            // inclusive scan of the composition across lanes
            fl_offmap inc = m;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                fl_offmap o;
                o.a = __shfl_up(inc.a, d, 64);
                o.c = __shfl_up(inc.c, d, 64);
                o.has = __shfl_up(inc.has, d, 64);
                if (lane >= (uint32_t)d) inc = fl_offmap_compose(o, inc);
            }
            if (write) {
                // exclusive = inclusive of the previous lane
                fl_offmap exc;
                exc.a = __shfl_up(inc.a, 1, 64);
                exc.c = __shfl_up(inc.c, 1, 64);
                exc.has = __shfl_up(inc.has, 1, 64);
                if (lane == 0) exc = ident;
                if (plan && plan->valid) {
                    plan->bit_off = fl_offmap_apply(fl_offmap_compose(run, exc), base);
                    // Q1: this block's slice ends q1_gap bytes before the bytes its tokens cover and the next block's starts
                    // there.  One of the two stored, the other Huffman coded: the reference's stream loses those bytes (this
                    // block stored) or holds them twice (the next one stored) -- reported, not repaired (fl_block_plan.q1_gap)
                    if (plan->q1_gap && j + 1 < ck.n_blocks) {
                        const fl_block_plan* nx = &plans[ck.first_block + j + 1];
                        if (nx->valid && (plan->type == FL_BLOCK_STORED) != (nx->type == FL_BLOCK_STORED)) q1_sh = 1u;
                    }
                }
            }
            fl_offmap last;
            last.a = __shfl(inc.a, 63, 64);
            last.c = __shfl(inc.c, 63, 64);
            last.has = __shfl(inc.has, 63, 64);
            run = fl_offmap_compose(run, last);
        }
        return run;
    };
    fl_offmap run;
    if (W == 1) {
        run = pass(ident, true);
    } else {
        const fl_offmap mine = pass(ident, false);
        if (lane == 0) wtot[wave] = mine;
        __syncthreads();
        fl_offmap pre = ident;
        for (uint32_t x = 0; x < wave; x++) pre = fl_offmap_compose(pre, wtot[x]);
        run = pass(pre, true);  // (the fields come from L2 this time)
        for (uint32_t x = wave + 1; x < W; x++) run = fl_offmap_compose(run, wtot[x]);  // every wave: the whole chunk
    }
    const uint64_t end_bits = fl_offmap_apply(run, base);
    const uint64_t body_end = (end_bits + 7) >> 3;  // bit_writer.flush pads the last byte (bit_writer.zig:46-61)
    const uint64_t total = body_end + ftr_bytes - ck.out_off;
    const bool fits = total <= ck.out_cap;

    // checksum over the whole chunk: fold the per-block parts
    uint32_t cks = 0;
    if (prm.container != 0 && threadIdx.x == 0) {
        if (prm.mode >= 4 && !prm.stream) {
            cks = cks_part[2 * (uint64_t)ck.first_block];
            if (prm.container == 2) {
                const uint32_t A = cks & 0xffff, B = cks >> 16, n = ck.in_len;
                const uint32_t a = (1 + A) % 65521u;
                const uint32_t bsum = (uint32_t)(((uint64_t)n + B) % 65521u);  // b = 0 + 1*n + B
                cks = a | (bsum << 16);
            }
        }
    }
    if (prm.container != 0 && !(prm.mode >= 4 && !prm.stream)) {
        // Every lane folds a run of consecutive blocks (Horner: one polynomial product per block), then moves its result to
        // the END of the chunk -- crc(A || B) = crc(A) x^(8 |B|) + crc(B): the run's part of the chunk's CRC is
        // crc(run) x^(8 * bytes behind the run); Adler-32: A = sum of the A's, B = sum of (B + A * bytes behind) -- and the
        // parts are simply added up: no scan of polynomial products (a 6-step scan of up-to-27-product powers was most of
        // this kernel's 0.19 ms on a stream of 2049 blocks).
        const uint32_t nbw = r1 - r0, per = (nbw + 63) / 64;
        const uint32_t j0 = r0 + min(lane * per, nbw), j1 = min(j0 + per, r1);
        uint32_t x = 0, y = 0, len = 0;  // crc | (A, B) and bytes of this lane's run (a chunk has fewer than 2^32 bytes)
        for (uint32_t j = j0; j < j1; j++) {
            const uint32_t pc = cks_part[2 * (uint64_t)(ck.first_block + j)];
            const uint32_t pl = cks_part[2 * (uint64_t)(ck.first_block + j) + 1];
            if (prm.container == 1) {
                const uint32_t shf = pl == FL_BLOCK_BYTES ? cc.pow65535 : fl_crc_xpow8n(cc.xpow8, pl);
                x = fl_crc_mulmod(x, shf) ^ pc;
            } else {
                y = (uint32_t)(((uint64_t)y + (uint64_t)x * pl + (pc >> 16)) % 65521u);
                x = (x + (pc & 0xffff)) % 65521u;
            }
            len += pl;
        }
        const uint32_t incl = fl_wave_incl_scan_dpp(len);  // bytes of the wave's range up to and including this lane's run
        if (lane == 63) wl[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < W; w++) {
            if (w < wave) before += (uint32_t)wl[w];
            all += (uint32_t)wl[w];
        }
        const uint32_t behind = all - (before + incl);
        uint32_t px, py = 0;
        if (prm.container == 1) {
            px = fl_wave_xor(len ? fl_crc_mulmod(x, fl_crc_xpow8n(cc.xpow8, behind)) : 0u);
        } else {
            px = fl_wave_sum(x);                                                                  // (64 terms below 65521 each)
            py = fl_wave_sum((uint32_t)(((uint64_t)y + (uint64_t)x * (behind % 65521u)) % 65521u));
        }
        if (lane == 0) {
            wx[wave] = px;
            wy[wave] = py;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t fx = 0, fy = 0;
            for (uint32_t w = 0; w < W; w++) {
                if (prm.container == 1) {
                    fx ^= wx[w];
                } else {
                    fx = (fx + wx[w]) % 65521u;
                    fy = (fy + wy[w]) % 65521u;
                }
            }
            cks = prm.container == 1 ? fx : (((1u + fx) % 65521u) | ((uint32_t)((all % 65521u + fy) % 65521u) << 16));  // a = 1 + A, b = n + B
        }
    }
    __syncthreads();  // (q1_sh)
    if (threadIdx.x == 0) {
        out_len[c] = fits ? total : 0;
        // FLATE_HIP_ST_OUTPUT_TOO_SMALL; FLATE_HIP_ST_REFERENCE_Q1_STREAM: the bytes are the reference's, and do not inflate to the input
        status[c] = fits ? (q1_sh ? 102 : 0) : 100;
        if (fits) {
            uint8_t* o = out + ck.out_off;
            if (prm.container == 1) {  // container.zig:64
                const uint8_t h[10] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0x03};
                for (int i = 0; i < 10; i++) o[i] = h[i];
                uint8_t* f = out + body_end;  // container.zig:92-96
                for (int i = 0; i < 4 && ftr_bytes; i++) f[i] = (uint8_t)(cks >> (8 * i));
                for (int i = 0; i < 4 && ftr_bytes; i++) f[4 + i] = (uint8_t)(ck.in_len >> (8 * i));
            } else if (prm.container == 2) {  // container.zig:78, 104
                o[0] = 0x78;
                o[1] = 0x9c;
                uint8_t* f = out + body_end;
                for (int i = 0; i < 4 && ftr_bytes; i++) f[i] = (uint8_t)(cks >> (8 * (3 - i)));
            }
        }
    }
    if (W > 1) __syncthreads();  // (no wave clears a plan another wave's second pass still reads)
    if (!fits)  // (the same verdict in every thread)
        for (uint32_t j = threadIdx.x; j < ck.n_blocks; j += blockDim.x) plans[ck.first_block + j].valid = 0;
}

// ------------------------------------------------------------------ encode
// Wave-parallel bit packer.  A block is a sequence of "items": the header bytes
// the planner produced, then one item per token (levels 4..9) or per input byte
// (huffman-only), then the end-of-block code (block_writer.zig:492-520, 563-571).
// The workgroup's 4 waves own contiguous quarters of the item sequence; a first
// pass sums the bit lengths so each wave knows where its bits start, the second
// pass packs 64 items at a time: wave prefix sum of lengths, ds_or into an LDS
// staging window, then whole dwords go out coalesced.  Only the first and last
// dword of a wave's range can be shared with a neighbour; those are atomic ORs
// into the pre-zeroed output.
#define FL_ENC_WAVES 4
#define FL_STG_DW 128  // (64 items of 60 bits at most + what is carried over)
// items that hold the block's symbols: a token each, or four bytes each (huffman-only)
#define FL_ENC_UNITS(TOKENS, n_sym) ((TOKENS) ? (n_sym) : (((n_sym) + 3u) >> 2))

struct fl_item {
    uint64_t v;
    uint32_t n;
};

template <bool TOKENS>
__device__ __forceinline__ fl_item fl_block_item(uint32_t i, uint32_t n_hdr, uint32_t hdr_nbits, uint32_t n_sym,
                                                 const uint8_t* __restrict__ hdr,
                                                 const uint8_t* __restrict__ bytes,
                                                 const uint32_t* __restrict__ toks, const uint32_t* lit_lds,
                                                 const uint32_t* dist_lds) {
    fl_item it;
    it.v = 0;
    it.n = 0;
    if (i < n_hdr) {
        const uint32_t rem = hdr_nbits - 8 * i;
        it.n = rem < 8 ? rem : 8;
        it.v = hdr[i] & ((1u << it.n) - 1);
    } else if (i < n_hdr + FL_ENC_UNITS(TOKENS, n_sym)) {
        const uint32_t k = i - n_hdr;
        if (!TOKENS) {
            // FOUR bytes an item (huffman-only: 4 x 15 bits at most): the scan, the placing and the LDS traffic of the packing
            // are paid per item, not per byte (round 5: config #4's k_encode 0.35 ms)
            uint64_t v = 0;
            uint32_t n = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                if (4 * k + j < n_sym) {
                    const uint32_t e = lit_lds[bytes[4 * k + j]];
                    v |= (uint64_t)(e & 0xffff) << n;
                    n += e >> 16;
                }
            }
            it.v = v;
            it.n = n;
        } else {
            const uint32_t t = toks[k];
            if (!FL_TOK_IS_MATCH(t)) {
                const uint32_t e = lit_lds[FL_TOK_LENLIT(t)];
                it.v = e & 0xffff;
                it.n = e >> 16;
            } else {
                const uint32_t ll = FL_TOK_LENLIT(t);
                const uint32_t li = fl_len_index(ll);
                const uint32_t le = lit_lds[257 + li];
                uint64_t v = le & 0xffff;
                uint32_t n = le >> 16;
                const uint32_t leb = fl_len_extra_bits(li);
                v |= (uint64_t)(ll - fl_len_base_scaled(li)) << n;
                n += leb;
                const uint32_t d = FL_TOK_DIST0(t);
                const uint32_t dc = fl_dist_code(d);
                const uint32_t de = dist_lds[dc];
                v |= (uint64_t)(de & 0xffff) << n;
                n += de >> 16;
                const uint32_t deb = fl_dist_extra_bits(dc);
                v |= (uint64_t)(d - fl_dist_base_scaled(dc)) << n;
                n += deb;
                it.v = v;
                it.n = n;
            }
        }
    } else if (i == n_hdr + FL_ENC_UNITS(TOKENS, n_sym)) {
        const uint32_t e = lit_lds[FL_EOB];
        it.v = e & 0xffff;
        it.n = e >> 16;
    }
    return it;
}

template <bool TOKENS>
__global__ __launch_bounds__(64 * FL_ENC_WAVES) void k_encode(const uint8_t* __restrict__ in,
                                                              const fl_chunk* __restrict__ chunks,
                                                              const uint32_t* __restrict__ blk_chunk,
                                                              const fl_block_plan* __restrict__ plans,
                                                              const uint32_t* __restrict__ tokens /* [chunk][65536] */,
                                                              uint32_t* __restrict__ out32) {
    __shared__ uint32_t lit_lds[FL_NUM_LIT + 2];
    __shared__ uint32_t dist_lds[FL_NUM_DIST + 2];
    __shared__ uint32_t wave_bits[FL_ENC_WAVES];
    __shared__ uint32_t stg[FL_ENC_WAVES][FL_STG_DW];

    const uint32_t b = blockIdx.x;
    const fl_block_plan* plan = &plans[b];
    if (!plan->valid) return;
    const uint32_t cidx = blk_chunk[b];
    const fl_chunk ck = chunks[cidx];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t bit_off = plan->bit_off;
    const uint8_t* src = in + ck.in_off;

    if (plan->type == FL_BLOCK_STORED) {
        // storedHeader + bytes (block_writer.zig:283-291, 385-388)
        const uint32_t len = plan->in_len;
        const uint64_t p = (bit_off + 3 + 7) >> 3;
        if (tid == 0) {
            if (plan->final_block) fl_atomic_or_bits(out32, bit_off, 1, 1);
            const uint32_t lw = (len & 0xffff) | ((~len & 0xffff) << 16);
            fl_atomic_or_bits(out32, p * 8, lw, 32);
        }
        fl_copy_bytes(out32, p + 4, src + plan->in_start, len, tid, 64 * FL_ENC_WAVES);
        return;
    }

    for (uint32_t i = tid; i < FL_NUM_LIT; i += 64 * FL_ENC_WAVES)
        lit_lds[i] = (uint32_t)plan->lit[i].code | ((uint32_t)plan->lit[i].len << 16);
    if (tid < FL_NUM_DIST) dist_lds[tid] = (uint32_t)plan->dist[tid].code | ((uint32_t)plan->dist[tid].len << 16);
    for (uint32_t i = lane; i < FL_STG_DW; i += 64) stg[wave][i] = 0;
    __syncthreads();

    const uint32_t hdr_nbits = plan->hdr_nbits;
    const uint32_t n_hdr = (hdr_nbits + 7) >> 3;
    const uint32_t n_sym = plan->tok_count;
    const uint32_t n_items = n_hdr + FL_ENC_UNITS(TOKENS, n_sym) + 1;
    const uint8_t* bytes = src + plan->tok_start;
    const uint32_t* toks = TOKENS ? tokens + ck.pos_off + plan->tok_start : nullptr;
    const uint8_t* hdr = plan->hdr;

    // contiguous item range of this wave, a multiple of 64 items
    uint32_t per = (n_items + FL_ENC_WAVES - 1) / FL_ENC_WAVES;
    per = (per + 63) & ~63u;
    const uint32_t i0 = min(n_items, wave * per), i1 = min(n_items, i0 + per);

    // pass 1: bits per wave
    uint32_t nb = 0;
    for (uint32_t i = i0 + lane; i < i1; i += 64)
        nb += fl_block_item<TOKENS>(i, n_hdr, hdr_nbits, n_sym, hdr, bytes, toks, lit_lds, dist_lds).n;
    nb = fl_wave_sum(nb);
    if (lane == 0) wave_bits[wave] = nb;
    __syncthreads();
    uint64_t cur = bit_off;
    for (uint32_t w = 0; w < wave; w++) cur += wave_bits[w];
    if (i0 >= i1) return;

    // pass 2: pack
    const uint64_t first_dw = cur >> 5;
    uint32_t* sw = stg[wave];
    for (uint32_t ib = i0; ib < i1; ib += 64) {
        const uint32_t i = ib + lane;
        fl_item it;
        it.v = 0;
        it.n = 0;
        if (i < i1) it = fl_block_item<TOKENS>(i, n_hdr, hdr_nbits, n_sym, hdr, bytes, toks, lit_lds, dist_lds);
        const uint32_t incl = fl_wave_incl_scan(it.n, lane);
        const uint32_t total = __shfl(incl, 63, 64);
        const uint64_t base_dw = cur >> 5;
        if (it.n) {
            const uint32_t rel = (uint32_t)(cur - (base_dw << 5)) + (incl - it.n);
            const uint32_t dw = rel >> 5, sh = rel & 31;
            const uint64_t a = it.v << sh;
            const uint32_t hi = sh ? (uint32_t)(it.v >> (64 - sh)) : 0u;
            if ((uint32_t)a) atomicOr(&sw[dw], (uint32_t)a);
            if ((uint32_t)(a >> 32)) atomicOr(&sw[dw + 1], (uint32_t)(a >> 32));
            if (hi) atomicOr(&sw[dw + 2], hi);
        }
        fl_lds_order();
        const uint64_t end = cur + total;
        const uint32_t nd = (uint32_t)((end >> 5) - base_dw);  // complete dwords
        for (uint32_t k = lane; k < nd; k += 64) {
            const uint32_t v = sw[k];
            if (base_dw + k == first_dw) {
                if (v) atomicOr(&out32[base_dw + k], v);
            } else {
                out32[base_dw + k] = v;
            }
        }
        const uint32_t carry = sw[nd];
        fl_lds_order();
        // clear the window, keep the partial dword as the new first one
        for (uint32_t k = lane; k <= nd + 2 && k < FL_STG_DW; k += 64) sw[k] = 0;
        fl_lds_order();
        if (lane == 0) sw[0] = carry;
        fl_lds_order();
        cur = end;
    }
    if ((cur & 31) && lane == 0) {
        const uint32_t v = sw[0];
        if (v) atomicOr(&out32[cur >> 5], v);
    }
}

// The same, one WAVE per block (batches of many blocks: the chunk path): no split of a block among waves, so no
// first pass over the tokens to find out where each wave's bits start -- half the instructions.  A wave's chain is
// four times as long, four times as many blocks are in flight.
template <bool TOKENS>
__global__ __launch_bounds__(64 * FL_ENC_WAVES) void k_encode_wave(const uint8_t* __restrict__ in,
                                                              const fl_chunk* __restrict__ chunks,
                                                              const uint32_t* __restrict__ blk_chunk,
                                                              const fl_block_plan* __restrict__ plans,
                                                              const uint32_t* __restrict__ tokens /* [chunk][65536] */,
                                                              uint32_t* __restrict__ out32, uint32_t n_blocks,
                                                              uint32_t pair_slots) {
    __shared__ uint32_t lit_all[FL_ENC_WAVES][FL_NUM_LIT + 2];
    __shared__ uint32_t dist_all[FL_ENC_WAVES][FL_NUM_DIST + 2];
    __shared__ uint32_t stg[FL_ENC_WAVES][FL_STG_DW];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t b = blockIdx.x * FL_ENC_WAVES + wave;
    if (b >= n_blocks) return;  // (no workgroup barrier below: every wave is on its own)
    if (pair_slots) {
        // chunk path: two plan slots per chunk and the second one is rarely used: all first slots before the second
        // ones, so that resident waves are waves with work (as in k_plan)
        const uint32_t half = n_blocks >> 1;
        b = b < half ? 2 * b : 2 * (b - half) + 1;
    }
    const fl_block_plan* plan = &plans[b];
    if (!plan->valid) return;
    const uint32_t cidx = blk_chunk[b];
    const fl_chunk ck = chunks[cidx];
    uint32_t* lit_lds = lit_all[wave];
    uint32_t* dist_lds = dist_all[wave];
    const uint64_t bit_off = plan->bit_off;
    const uint8_t* src = in + ck.in_off;

    if (plan->type == FL_BLOCK_STORED) {
        // storedHeader + bytes (block_writer.zig:283-291, 385-388)
        const uint32_t len = plan->in_len;
        const uint64_t p = (bit_off + 3 + 7) >> 3;
        if (lane == 0) {
            if (plan->final_block) fl_atomic_or_bits(out32, bit_off, 1, 1);
            const uint32_t lw = (len & 0xffff) | ((~len & 0xffff) << 16);
            fl_atomic_or_bits(out32, p * 8, lw, 32);
        }
        fl_copy_bytes(out32, p + 4, src + plan->in_start, len, lane, 64);
        return;
    }

    for (uint32_t i = lane; i < FL_NUM_LIT; i += 64)
        lit_lds[i] = (uint32_t)plan->lit[i].code | ((uint32_t)plan->lit[i].len << 16);
    if (lane < FL_NUM_DIST) dist_lds[lane] = (uint32_t)plan->dist[lane].code | ((uint32_t)plan->dist[lane].len << 16);
    for (uint32_t i = lane; i < FL_STG_DW; i += 64) stg[wave][i] = 0;
    fl_lds_order();

    const uint32_t hdr_nbits = plan->hdr_nbits;
    const uint32_t n_hdr = (hdr_nbits + 7) >> 3;
    const uint32_t n_sym = plan->tok_count;
    const uint32_t n_items = n_hdr + FL_ENC_UNITS(TOKENS, n_sym) + 1;
    const uint8_t* bytes = src + plan->tok_start;
    const uint32_t* toks = TOKENS ? tokens + ck.pos_off + plan->tok_start : nullptr;
    const uint8_t* hdr = plan->hdr;

    (void)n_items;
    uint64_t cur = bit_off;

    // pack 64 items (one a lane, it.n = 0: none) behind `cur`: prefix sum of the lengths, ds_or into the staging window, whole
    // dwords out; the block's first dword is shared with the block before it (an atomic OR into the cleared output)
    const uint64_t first_dw = cur >> 5;
    uint32_t* sw = stg[wave];
    auto pack = [&](const fl_item& it) {
        const uint32_t incl = fl_wave_incl_scan(it.n, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint64_t base_dw = cur >> 5;
        if (it.n) {
            const uint32_t rel = (uint32_t)(cur - (base_dw << 5)) + (incl - it.n);
            const uint32_t dw = rel >> 5, sh = rel & 31;
            const uint64_t a = it.v << sh;
            const uint32_t hi = sh ? (uint32_t)(it.v >> (64 - sh)) : 0u;
            if ((uint32_t)a) atomicOr(&sw[dw], (uint32_t)a);
            if ((uint32_t)(a >> 32)) atomicOr(&sw[dw + 1], (uint32_t)(a >> 32));
            if (hi) atomicOr(&sw[dw + 2], hi);
        }
        fl_lds_order();
        const uint64_t end = cur + total;
        const uint32_t nd = (uint32_t)((end >> 5) - base_dw);  // complete dwords (at most 120: two rounds of the lanes)
        {
            const uint32_t v0 = lane < nd ? sw[lane] : 0u, v1 = lane + 64u < nd ? sw[lane + 64u] : 0u;
            if (lane < nd) {
                if (base_dw + lane == first_dw) {
                    if (v0) atomicOr(&out32[base_dw + lane], v0);
                } else {
                    out32[base_dw + lane] = v0;
                }
            }
            if (lane + 64u < nd) out32[base_dw + lane + 64u] = v1;  // (never the block's first dword)
        }
        const uint32_t carry = sw[nd];
        fl_lds_order();
        // clear the window, keep the partial dword as the new first one
        if (lane <= nd + 2) sw[lane] = lane == 0 ? carry : 0u;
        if (lane + 64u <= nd + 2 && lane + 64u < FL_STG_DW) sw[lane + 64u] = 0u;
        fl_lds_order();
        cur = end;
    };
    if (!TOKENS) {
        // (huffman-only blocks of a batch this large: the generic items, 64 at a time)
        for (uint32_t ib = 0; ib < n_items; ib += 64) {
            fl_item it;
            it.v = 0;
            it.n = 0;
            if (ib + lane < n_items) it = fl_block_item<TOKENS>(ib + lane, n_hdr, hdr_nbits, n_sym, hdr, bytes, toks, lit_lds, dist_lds);
            pack(it);
        }
    } else {
        // Round 6: three loops instead of one over "items" -- the header's bytes, the tokens, the end-of-block code -- so that the
        // tokens' loop carries no branch for the other two, and a group's tokens are requested while the group before is packed
        // (the load sat in the loop with its own wait: 1.08 -> see profiles/r06_config2_kernel_stats.csv).
        for (uint32_t ib = 0; ib < n_hdr; ib += 64) {
            fl_item it;
            it.v = 0;
            it.n = 0;
            const uint32_t i = ib + lane;
            if (i < n_hdr) {
                const uint32_t rem = hdr_nbits - 8 * i;
                it.n = rem < 8 ? rem : 8;
                it.v = hdr[i] & ((1u << it.n) - 1);
            }
            pack(it);
        }
        const uint32_t eob = lit_lds[FL_EOB];
        uint32_t t_next = lane < n_sym ? toks[lane] : 0u;
        for (uint32_t k0 = 0; k0 < n_sym; k0 += 64) {
            const uint32_t t = t_next;
            t_next = k0 + 64 + lane < n_sym ? toks[k0 + 64 + lane] : 0u;
            fl_item it;
            it.v = 0;
            it.n = 0;
            const uint32_t k = k0 + lane;
            if (k < n_sym) {
                if (!FL_TOK_IS_MATCH(t)) {
                    const uint32_t e = lit_lds[FL_TOK_LENLIT(t)];
                    it.v = e & 0xffff;
                    it.n = e >> 16;
                } else {
                    const uint32_t ll = FL_TOK_LENLIT(t);
                    const uint32_t li = fl_len_index(ll);
                    const uint32_t le = lit_lds[257 + li];
                    uint64_t v = le & 0xffff;
                    uint32_t n = le >> 16;
                    v |= (uint64_t)(ll - fl_len_base_scaled(li)) << n;
                    n += fl_len_extra_bits(li);
                    const uint32_t d = FL_TOK_DIST0(t);
                    const uint32_t dc = fl_dist_code(d);
                    const uint32_t de = dist_lds[dc];
                    v |= (uint64_t)(de & 0xffff) << n;
                    n += de >> 16;
                    v |= (uint64_t)(d - fl_dist_base_scaled(dc)) << n;
                    n += fl_dist_extra_bits(dc);
                    it.v = v;
                    it.n = n;
                }
            } else if (k == n_sym) {  // the end-of-block code rides in the last group when a lane is free
                it.v = eob & 0xffff;
                it.n = eob >> 16;
            }
            pack(it);
        }
        if ((n_sym & 63u) == 0) {  // every lane of the last group held a token (or there was none): a group of its own
            fl_item it;
            it.v = lane == 0 ? (eob & 0xffff) : 0u;
            it.n = lane == 0 ? (eob >> 16) : 0u;
            pack(it);
        }
    }
    if ((cur & 31) && lane == 0) {
        const uint32_t v = sw[0];
        if (v) atomicOr(&out32[cur >> 5], v);
    }
}

// ------------------------------------------------------------------ packing
// dst_off = exclusive scan of out_len (one workgroup; n is at most a few 100k)
__global__ __launch_bounds__(1024) void k_scan_lens(const uint64_t* __restrict__ len, uint32_t n,
                                                    uint64_t* __restrict__ dst_off) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint64_t v = i < n ? len[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t off = carry;
        for (uint32_t w = 0; w < wave; w++) off += wsum[w];
        if (i < n) dst_off[i] = off + inc - v;
        __syncthreads();
        if (tid == 1023) carry = off + inc;
        __syncthreads();
    }
    if (tid == 0) dst_off[n] = carry;
}

// gridDim.y workgroups per stream, each over slices of 16 KiB taken round robin (a batch of a few long
// streams fills the chip as well as one of many short ones): interior destination bytes as 16-byte
// stores from unaligned source dwords, the (at most 15 + 15) edge bytes of the stream one by one --
// neighbouring streams never share a byte.
#define FL_GATHER_SLICE 16384u
__device__ __forceinline__ void fl_gather_one(const uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off,
                                              const uint64_t* __restrict__ out_len, uint8_t* __restrict__ dst,
                                              const uint64_t* __restrict__ dst_off, uint32_t c, uint32_t by, uint32_t ny,
                                              uint64_t skip = 0) {
    const uint32_t tid = threadIdx.x;
    const uint64_t n0 = out_len[c];
    if (n0 <= skip) return;
    const uint64_t n = n0 - skip;
    const uint8_t* s = out + out_off[c] + skip;
    uint8_t* d = dst + dst_off[c] + skip;
    const uint64_t head = min(n, (uint64_t)((16 - ((uintptr_t)d & 15)) & 15));
    const uint64_t nq = (n - head) >> 4;  // 16-byte units
    const uint64_t tail0 = head + 16 * nq;
    if (by == 0) {
        if (tid < head) d[tid] = s[tid];
        if (tid < 16 && tail0 + tid < n) d[tail0 + tid] = s[tail0 + tid];
    }
    uint4* d16 = (uint4*)(d + head);
    const uint8_t* s0 = s + head;
    const uint32_t sh = (uint32_t)((uintptr_t)s0 & 3);
    const uint32_t* sa = (const uint32_t*)(s0 - sh);  // aligned dwords: unit i = dwords 4 i .. 4 i + 4 shifted by sh bytes
    const uint64_t per = FL_GATHER_SLICE / 16;
    for (uint64_t u0 = (uint64_t)by * per; u0 < nq; u0 += (uint64_t)ny * per) {
        const uint64_t u1 = min(u0 + per, nq);
        for (uint64_t i = u0 + tid; i < u1; i += 256) {
            const uint32_t* w = sa + 4 * i;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            uint4 v;
            if (sh) {
                const uint32_t w4 = w[4];  // (holds bytes of this unit: inside the stream)
                v.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
                v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
                v.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
                v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
            } else {
                v = make_uint4(w0, w1, w2, w3);
            }
            d16[i] = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_gather_copy(const uint8_t* __restrict__ out,
                                                     const uint64_t* __restrict__ out_off,
                                                     const uint64_t* __restrict__ out_len,
                                                     uint8_t* __restrict__ dst, const uint64_t* __restrict__ dst_off) {
    fl_gather_one(out, out_off, out_len, dst, dst_off, blockIdx.x, blockIdx.y, gridDim.y);
}
// The same copy by a FEW workgroups that take the streams in turn: the destination is pinned host memory, and a grid that
// floods the write path with stores that wait for the link stalls every other kernel's stores behind them (the next
// sub-batch's kernels did not start before the copy had ended: rocprofv3 timeline, tools/e2e_timeline.py).
// Of the first `urows` slots the DMA engine's rectangle copy has taken the first `skip` bytes: only what lies behind them.
// `len_dst` (may be null): the streams' lengths go home the same way (pinned host memory the device can write), so that no
// DMA copy that waits for this sub-batch sits in the engine's queue in front of the next sub-batch's input.
__global__ __launch_bounds__(256) void k_copy_slots(const uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off,
                                                    const uint64_t* __restrict__ out_len, uint8_t* __restrict__ dst,
                                                    const uint64_t* __restrict__ dst_off, uint32_t n, uint32_t urows, uint64_t skip,
                                                    uint64_t* __restrict__ len_dst) {
    for (uint32_t c = blockIdx.x; c < n; c += gridDim.x) fl_gather_one(out, out_off, out_len, dst, dst_off, c, 0, 1, c < urows ? skip : 0);
    if (len_dst)
        for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) len_dst[c] = out_len[c];
}
