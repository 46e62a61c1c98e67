// flate_common.h -- constants, token format, RFC 1951 tables and the serial
// "block planner" (Huffman code construction, code-length RLE, block-type
// decision, dynamic header bits) shared by the HIP kernels.
//
// The planner functions are plain C++ marked FL_HD so that the very same
// source is (a) what one lane per block executes on the GPU and (b) unit
// testable on the CPU build (tests/cpu_shim, sanitizers run there; GPU ASan is
// not available on this pool).  The CPU build is test infrastructure only --
// libflate_hip.so has no CPU execution path.
//
// Reference behaviour reproduced here (file:line in /root/reference/src/flate):
//   huffman_encoder.zig:62-278   generate / bitCounts / assignEncodingAndSize
//   block_writer.zig:78-171      generateCodegen
//   block_writer.zig:179-229     dynamicSize / fixedSize / storedSizeFits
//   block_writer.zig:237-300     dynamicHeader / fixedHeader
//   block_writer.zig:307-383     write        (token blocks, levels 4..9)
//   block_writer.zig:524-572     huffmanBlock (huffman-only mode)
//   Token.zig:58-81,114-276      length / distance code tables
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FL_HD __host__ __device__ inline
#else
#define FL_HD inline
#endif

// ---- token format (same as the oracle / the debug seam in flate_hip.h) ----
// bit 23: kind (1 = match); bits 15..22: literal byte or length-3; bits 0..14: distance-1
#define FL_TOK_LIT(b) ((uint32_t)(b) << 15)
#define FL_TOK_MATCH(dist, len) ((1u << 23) | ((uint32_t)((len)-3) << 15) | (uint32_t)((dist)-1))
#define FL_TOK_IS_MATCH(t) (((t) >> 23) & 1u)
#define FL_TOK_LENLIT(t) (((t) >> 15) & 0xffu)
#define FL_TOK_DIST0(t) ((t)&0x7fffu)

// consts.zig
// On the GPU the planner is executed by all 64 lanes of a wave in lock step, redundantly (same
// data, same control flow, same stores); only the two sorts of the Huffman code construction use
// the lanes for what they are.  On the CPU (tests/cpu_shim) it is plain serial code.
#if defined(__HIP_DEVICE_COMPILE__)
#define FL_PLAN_PARALLEL 1
#define FL_PLAN_LANE() (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define FL_PLAN_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
// loops whose iterations are independent: strided over the lanes on the GPU; sums are
// completed with FL_PLAN_REDUCE
#define FL_PLAN_FOR(i, a, b) for (uint32_t i = (a) + FL_PLAN_LANE(); i < (b); i += 64)
static __device__ __forceinline__ uint32_t fl_plan_wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}
#define FL_PLAN_REDUCE(v) (v) = fl_plan_wave_sum(v)
#else
#define FL_PLAN_PARALLEL 0
// CPU build: which form of the Huffman bit counts fl_huff_generate runs (tests flip it)
static int fl_plan_cpu_use_pm = 0;
#define FL_PLAN_CPU_USE_PM fl_plan_cpu_use_pm
#define FL_PLAN_FOR(i, a, b) for (uint32_t i = (a); i < (b); i++)
#define FL_PLAN_REDUCE(v)
#define FL_PLAN_SYNC()
#endif

#if defined(__HIP_DEVICE_COMPILE__) && defined(FL_PLAN_PROF)
extern __device__ uint64_t g_fl_prof[];
#define FL_PLAN_MARK(slot)                                                                          \
    do {                                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fl_prof[slot] = __builtin_readcyclecounter();    \
    } while (0)
#else
#define FL_PLAN_MARK(slot)
#endif
#define FL_MAX_TOKENS 32768u     // consts.zig:7  tokens per block
#define FL_MIN_MATCH 4u          // consts.zig:12
#define FL_MAX_MATCH 258u        // consts.zig:13
#define FL_MAX_DIST 32768u       // consts.zig:16
#define FL_MAX_STORE 65535u      // consts.zig:46
#define FL_NUM_LIT 286
#define FL_NUM_DIST 30
#define FL_NUM_CG 19
#define FL_EOB 256
#define FL_END_MARK 255

enum { FL_BLOCK_STORED = 0, FL_BLOCK_FIXED = 1, FL_BLOCK_DYNAMIC = 2 };

struct fl_hcode {
    uint16_t code;  // bit-reversed (LSB-first) as the reference stores it
    uint16_t len;
};

// ---- RFC 1951 3.2.5 tables (Token.zig:143-276) ----
FL_HD uint32_t fl_len_extra_bits(uint32_t idx) {  // idx = length code - 257
    return idx < 8 ? 0u : (idx == 28 ? 0u : (idx - 4) >> 2);
}
FL_HD uint32_t fl_len_base_scaled(uint32_t idx) {  // base - 3
    if (idx < 8) return idx;
    if (idx == 28) return 255;
    uint32_t e = (idx - 4) >> 2;
    return ((4 + ((idx - 4) & 3)) << e);
}
// length-3 (0..255) -> code index 0..28 (Token.zig:114-141)
FL_HD uint32_t fl_len_index(uint32_t len_lit) {
    if (len_lit < 8) return len_lit;
    if (len_lit == 255) return 28;
    uint32_t hb = 31u - (uint32_t)__builtin_clz(len_lit);  // >= 3
    return ((hb - 1) << 2) + ((len_lit >> (hb - 2)) & 3);
}
FL_HD uint32_t fl_dist_extra_bits(uint32_t code) { return code < 4 ? 0u : (code - 2) >> 1; }
FL_HD uint32_t fl_dist_base_scaled(uint32_t code) {  // base - 1
    if (code < 4) return code;
    uint32_t e = (code - 2) >> 1;
    return (2 + (code & 1)) << e;
}
// distance-1 (0..32767) -> code 0..29 (Token.zig:70-81)
FL_HD uint32_t fl_dist_code(uint32_t d) {
    if (d < 4) return d;
    uint32_t hb = 31u - (uint32_t)__builtin_clz(d);  // >= 2
    return (hb << 1) + ((d >> (hb - 1)) & 1);
}

FL_HD uint16_t fl_bit_reverse(uint16_t v, uint32_t n) {  // huffman_encoder.zig:455-458
    uint32_t x = v;
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
    x = ((x & 0x00ffu) << 8) | ((x >> 8) & 0x00ffu);
    return (uint16_t)(x >> (16 - n));
}

// fixed codes (huffman_encoder.zig:298-338)
FL_HD fl_hcode fl_fixed_lit_code(uint32_t ch) {
    uint32_t bits, size;
    if (ch <= 143) { bits = ch + 48; size = 8; }
    else if (ch <= 255) { bits = ch + 400 - 144; size = 9; }
    else if (ch <= 279) { bits = ch - 256; size = 7; }
    else { bits = ch + 192 - 280; size = 8; }
    fl_hcode c;
    c.code = fl_bit_reverse((uint16_t)bits, size);
    c.len = (uint16_t)size;
    return c;
}
FL_HD fl_hcode fl_fixed_dist_code(uint32_t ch) {
    fl_hcode c;
    c.code = fl_bit_reverse((uint16_t)ch, 5);
    c.len = 5;
    return c;
}

// consts.zig:30
FL_HD uint32_t fl_codegen_order(uint32_t i) {
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return order[i];
}

// ---- per-block scratch: lives in LDS on the GPU (one per planning wave) ----
struct fl_level_info {
    uint32_t level, last_freq, next_char_freq, next_pair_freq, needed;
};
struct fl_plan_ws {
    uint16_t lit_freq[FL_NUM_LIT];
    uint16_t dist_freq[FL_NUM_DIST];
    uint16_t cg_freq[FL_NUM_CG];
    fl_hcode lit_codes[FL_NUM_LIT];
    fl_hcode dist_codes[FL_NUM_DIST];
    fl_hcode cg_codes[FL_NUM_CG];
    uint8_t codegen[FL_NUM_LIT + FL_NUM_DIST + 2];
    // Huffman construction scratch
    uint16_t list_sym[FL_NUM_LIT + 1];
    uint16_t list_freq[FL_NUM_LIT + 1];
    uint32_t bit_count[17];
#if !FL_PLAN_PARALLEL
    fl_level_info levels[18];  // the reference's lazy loop (the CPU build only: cross-check of the package-merge form; on the
    uint32_t leaf_counts[17][16];  // device they would cost a wave 1.5 KB of LDS: 16 instead of 20 planner waves per CU)
#endif
    // package-merge form of the same computation (fl_huff_bit_counts_pm)
    uint32_t pm_w[2 * FL_NUM_LIT];   // weights of the current level's list (2n - 2 items)
    uint32_t pm_p[FL_NUM_LIT];       // packages = pair sums of the previous level's list
    uint32_t pm_mask[16][18];        // per level: bit r = item r of the list is a leaf
    uint32_t pm_c[16];               // per level (<= 15 bits: 0..15): leaves among the items the solution takes (16, not 17: the struct is 8192 bytes on the device, 20 planner waves per CU)
};

// what the planner hands to the encode kernel (global memory, one per block)
#define FL_HDR_BYTES 640  // dynamic header <= 17 + 19*3 + 316*(7+7) bits = 4498 bits
struct fl_block_plan {
    uint32_t type;       // FL_BLOCK_*
    uint32_t size_bits;  // exact size of the encoded block (Huffman types); stored: see fl_stored_bits
    uint32_t hdr_nbits;  // bits in hdr[] (block header incl. BFINAL/BTYPE)
    uint32_t final_block;
    uint32_t in_start;   // stored-block source: chunk-relative byte range
    uint32_t in_len;
    uint32_t tok_start;  // chunk-relative token index (token blocks) / byte index (huffman-only)
    uint32_t tok_count;
    uint32_t valid;
    uint32_t no_input;   // token blocks: the raw input slice is gone (window slide since the last flush)
    // Quirk Q1 (deflate.zig:227-230 -> 268-270 -> SlidingWindow.zig:119-123; the window advances only at deflate.zig:193):
    // when the block's 32768th token is a match, the slice [in_start, in_start + in_len) ends q1_gap bytes before the bytes
    // its tokens cover, and the next block's slice starts that early.  Harmless while both blocks are Huffman coded or both
    // stored; with exactly one of them stored the reference's stream loses or repeats those bytes (k_offsets reports it).
    uint32_t q1_gap;
    uint64_t bit_off;  // absolute bit offset in `out`, filled by the offset scan
    uint8_t hdr[FL_HDR_BYTES];
    fl_hcode lit[FL_NUM_LIT];
    fl_hcode dist[FL_NUM_DIST];
};

// ---- Huffman code construction -------------------------------------------------
// huffman_encoder.zig:122-247.  Quirk Q3 kept verbatim: the exhausted-leaf
// sentinel is 65535 (maxInt(u16), :189,282-287) whereas the "out of leaves and
// pairs" test (:170) compares with maxInt(i32) and therefore never fires;
// comparisons are strict `<` on u32.
#if !FL_PLAN_PARALLEL
FL_HD void fl_huff_bit_counts(fl_plan_ws* ws, uint32_t n, uint32_t max_bits) {
    const uint16_t* freq = ws->list_freq;
    fl_level_info* levels = ws->levels;
    if (max_bits > n - 1) max_bits = n - 1;
#if FL_PLAN_PARALLEL
    {
        const uint32_t lane = FL_PLAN_LANE();
        if (lane < 18) {
            levels[lane].level = 0; levels[lane].last_freq = 0; levels[lane].next_char_freq = 0;
            levels[lane].next_pair_freq = 0; levels[lane].needed = 0;
        }
        for (uint32_t i = lane; i < 17 * 16; i += 64) (&ws->leaf_counts[0][0])[i] = 0;
        FL_PLAN_SYNC();
    }
#else
    for (uint32_t i = 0; i < 18; i++) {
        levels[i].level = 0; levels[i].last_freq = 0; levels[i].next_char_freq = 0;
        levels[i].next_pair_freq = 0; levels[i].needed = 0;
    }
    for (uint32_t i = 0; i < 17; i++)
        for (uint32_t j = 0; j < 16; j++) ws->leaf_counts[i][j] = 0;
#endif
    for (uint32_t level = 1; level <= max_bits; level++) {
        levels[level].level = level;
        levels[level].last_freq = freq[1];
        levels[level].next_char_freq = freq[2];
        levels[level].next_pair_freq = (uint32_t)freq[0] + freq[1];
        levels[level].needed = 0;
        ws->leaf_counts[level][level] = 2;
        if (level == 1) levels[level].next_pair_freq = 0x7fffffffu;
    }
    levels[max_bits].needed = 2 * n - 4;
    uint32_t level = max_bits;
    for (;;) {
        fl_level_info* l = &levels[level];
        if (l->next_pair_freq == 0x7fffffffu && l->next_char_freq == 0x7fffffffu) {
            l->needed = 0;
            levels[level + 1].next_pair_freq = 0x7fffffffu;
            level += 1;
            continue;
        }
        uint32_t prev_freq = l->last_freq;
        if (l->next_char_freq < l->next_pair_freq) {
            uint32_t next = ws->leaf_counts[level][level] + 1;
            l->last_freq = l->next_char_freq;
            ws->leaf_counts[level][level] = next;
            l->next_char_freq = (next >= n) ? 65535u : (uint32_t)freq[next];
        } else {
            l->last_freq = l->next_pair_freq;
#if FL_PLAN_PARALLEL
            {
                const uint32_t lane = FL_PLAN_LANE();
                if (lane < level) ws->leaf_counts[level][lane] = ws->leaf_counts[level - 1][lane];
                FL_PLAN_SYNC();
            }
#else
            for (uint32_t j = 0; j < level; j++) ws->leaf_counts[level][j] = ws->leaf_counts[level - 1][j];
#endif
            levels[l->level - 1].needed = 2;
        }
        l->needed -= 1;
        if (l->needed == 0) {
            if (l->level == max_bits) break;
            levels[l->level + 1].next_pair_freq = prev_freq + l->last_freq;
            level += 1;
        } else {
            while (levels[level - 1].needed > 0) {
                level -= 1;
                if (level == 0) break;
            }
        }
    }
    uint32_t bits = 1;
    for (uint32_t i = 0; i < 17; i++) ws->bit_count[i] = 0;
    for (uint32_t lv = max_bits; lv > 0; lv--) {
        ws->bit_count[bits] = ws->leaf_counts[max_bits][lv] - ws->leaf_counts[max_bits][lv - 1];
        bits++;
    }
}

#endif

// The same bit counts without the serial walk.  The reference's loop is the boundary form of
// package-merge (huffman_encoder.zig:122-247): level k's list is the merge of the leaves with the
// pair sums ("packages") of level k-1's list, a package going first when the weights tie
// (`next_char_freq < next_pair_freq` is strict, :172), and the solution takes the first 2n-2
// items of the top level, for every package among them two more items one level down.  Q3 lives
// on as leaves of weight 65535 that follow the real ones (:189).  Here every level's list is
// built outright, each item finding its place by binary search -- the items of a level are
// independent, so the wave's lanes share them; 15 levels of about 850 searches replace some
// 8000 dependent steps.  Checked against the loop above on the CPU (tests/test_planner_cpu.py).
#if FL_PLAN_PARALLEL
#define FL_PLAN_OR(p, v) atomicOr((p), (v))
#else
#define FL_PLAN_OR(p, v) (*(p) |= (v))
#endif
FL_HD void fl_huff_bit_counts_pm(fl_plan_ws* ws, uint32_t n, uint32_t max_bits) {
    const uint16_t* freq = ws->list_freq;  // ascending, n >= 3 entries
    if (max_bits > n - 1) max_bits = n - 1;
    const uint32_t T = 2 * n - 2, H = n - 1;  // items per list, packages per level
    uint32_t* W = ws->pm_w;
    uint32_t* P = ws->pm_p;
    FL_PLAN_FOR(i, 0, T) W[i] = i < n ? (uint32_t)freq[i] : 65535u;  // level 1: leaves only
    FL_PLAN_FOR(i, 0, 16 * 18) (&ws->pm_mask[0][0])[i] = 0;
    FL_PLAN_SYNC();
    for (uint32_t k = 2; k <= max_bits; k++) {
        FL_PLAN_FOR(j, 0, H) P[j] = W[2 * j] + W[2 * j + 1];
        FL_PLAN_SYNC();
        FL_PLAN_FOR(i, 0, T) {  // leaf i goes behind the packages that weigh no more
            const uint32_t lw = i < n ? (uint32_t)freq[i] : 65535u;
            uint32_t lo = 0, hi = H;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (P[mid] <= lw) lo = mid + 1; else hi = mid;
            }
            const uint32_t r = i + lo;
            if (r < T) {
                W[r] = lw;
                FL_PLAN_OR(&ws->pm_mask[k][r >> 5], 1u << (r & 31));
            }
        }
        FL_PLAN_FOR(j, 0, H) {  // package j goes behind the leaves that weigh less
            const uint32_t pw = P[j];
            if (pw <= 65535u) {  // (a heavier one lies behind all the 65535-leaves: never taken)
                uint32_t lo = 0, hi = n;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)freq[mid] < pw) lo = mid + 1; else hi = mid;
                }
                const uint32_t r = j + lo;
                if (r < T) W[r] = pw;
            }
        }
        FL_PLAN_SYNC();
    }
    // top down: of the m items taken at level k, c are leaves; the packages take two items each below
    uint32_t m = T;
    for (uint32_t k = max_bits; k >= 1; k--) {
        uint32_t c = 0;
        if (k == 1) {
            c = m;
        } else {
            FL_PLAN_FOR(w, 0, 18) {
                if (32 * w < m) {
                    uint32_t bits = ws->pm_mask[k][w];
                    if (m - 32 * w < 32) bits &= (1u << (m - 32 * w)) - 1u;
                    c += (uint32_t)__builtin_popcount(bits);
                }
            }
            FL_PLAN_REDUCE(c);
        }
        ws->pm_c[k] = c;
        m = 2 * (m - c);
    }
    ws->pm_c[0] = 0;
    FL_PLAN_SYNC();
    // huffman_encoder.zig:239-246: symbols in the lists of levels lv..max only have max - lv + 1 bits
    FL_PLAN_FOR(b, 0, 17) {
        uint32_t v = 0;
        if (b >= 1 && b <= max_bits) v = ws->pm_c[max_bits - b + 1] - ws->pm_c[max_bits - b];
        ws->bit_count[b] = v;
    }
    FL_PLAN_SYNC();
}

// huffman_encoder.zig:62-95 + 251-278.
FL_HD void fl_huff_generate(fl_plan_ws* ws, const uint16_t* freq, uint32_t nfreq,
                            uint32_t max_bits, fl_hcode* codes) {
    uint16_t* lsym = ws->list_sym;
    uint16_t* lfrq = ws->list_freq;
    uint32_t count = 0;
#if FL_PLAN_PARALLEL
    {
        const uint32_t lane = FL_PLAN_LANE();
        const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
        for (uint32_t base = 0; base < nfreq; base += 64) {  // stream compaction, 64 symbols per step
            const uint32_t i = base + lane;
            const uint32_t f = i < nfreq ? freq[i] : 0u;
            const uint64_t used = __ballot(f != 0);
            if (f != 0) {
                const uint32_t at = count + (uint32_t)__popcll(used & below);
                lsym[at] = (uint16_t)i;
                lfrq[at] = (uint16_t)f;
            } else if (i < nfreq) {
                codes[i].len = 0;
                codes[i].code = 0;
            }
            count += (uint32_t)__popcll(used);
        }
        FL_PLAN_SYNC();
    }
#else
    for (uint32_t i = 0; i < nfreq; i++) {
        if (freq[i] != 0) {
            lsym[count] = (uint16_t)i;
            lfrq[count] = freq[i];
            count++;
        } else {
            codes[i].len = 0;
            codes[i].code = 0;
        }
    }
#endif
    if (count <= 2) {
        for (uint32_t i = 0; i < count; i++) {
            codes[lsym[i]].code = (uint16_t)i;
            codes[lsym[i]].len = 1;
        }
        return;
    }
    // sort by (freq, symbol) (:355-361); the list is in symbol order, so the sort has to be
    // stable on freq alone
#if FL_PLAN_PARALLEL
    {
        // rank sort: every lane places its own (at most 5) elements: rank = how many elements
        // come before it.  All reads happen before the first write.
        const uint32_t lane = FL_PLAN_LANE();
        uint16_t ms[5], mf[5];
        uint32_t rk[5];
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) {
            const uint32_t e = lane + 64 * k;
            ms[k] = e < count ? lsym[e] : (uint16_t)0;
            mf[k] = e < count ? lfrq[e] : (uint16_t)0;
            rk[k] = 0;
        }
        for (uint32_t j = 0; j < count; j++) {
            const uint32_t fj = lfrq[j];
#pragma unroll
            for (uint32_t k = 0; k < 5; k++) rk[k] += (fj < mf[k] || (fj == mf[k] && j < lane + 64 * k)) ? 1u : 0u;
        }
        FL_PLAN_SYNC();
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) {
            if (lane + 64 * k < count) {
                lsym[rk[k]] = ms[k];
                lfrq[rk[k]] = mf[k];
            }
        }
        FL_PLAN_SYNC();
    }
#else
    for (uint32_t i = 1; i < count; i++) {  // insertion sort stands in for std.mem.sort: the order is total
        uint16_t s = lsym[i], f = lfrq[i];
        uint32_t j = i;
        while (j > 0 && lfrq[j - 1] > f) {
            lsym[j] = lsym[j - 1];
            lfrq[j] = lfrq[j - 1];
            j--;
        }
        lsym[j] = s;
        lfrq[j] = f;
    }
#endif
    if (nfreq == FL_NUM_LIT) FL_PLAN_MARK(49);
#if FL_PLAN_PARALLEL
    fl_huff_bit_counts_pm(ws, count, max_bits);
#else
    if (FL_PLAN_CPU_USE_PM)
        fl_huff_bit_counts_pm(ws, count, max_bits);
    else
        fl_huff_bit_counts(ws, count, max_bits);
#endif
    if (nfreq == FL_NUM_LIT) FL_PLAN_MARK(50);
    uint32_t used_bits = max_bits > count - 1 ? count - 1 : max_bits;
    uint32_t code = 0;
    uint32_t list_len = count;
    for (uint32_t n = 0; n <= used_bits; n++) {
        code = (code << 1) & 0xffffu;
        uint32_t bits = ws->bit_count[n];
        if (n == 0 || bits == 0) continue;
        // the `bits` most frequent remaining symbols get length n, codes in symbol order
        uint32_t lo = list_len - bits;
#if FL_PLAN_PARALLEL
        {
            // no need to sort: a symbol's code is the first code of the length plus the number
            // of smaller symbols in the chunk
            const uint32_t lane = FL_PLAN_LANE();
            for (uint32_t e = lo + lane; e < list_len; e += 64) {
                const uint32_t sy = lsym[e];
                uint32_t r = 0;
                for (uint32_t j = lo; j < list_len; j++) r += lsym[j] < sy ? 1u : 0u;
                codes[sy].code = fl_bit_reverse((uint16_t)((code + r) & 0xffffu), n);
                codes[sy].len = (uint16_t)n;
            }
            code = (code + bits) & 0xffffu;
            FL_PLAN_SYNC();
        }
#else
        for (uint32_t i = lo + 1; i < list_len; i++) {  // sort chunk by symbol
            uint16_t s = lsym[i];
            uint32_t j = i;
            while (j > lo && lsym[j - 1] > s) {
                lsym[j] = lsym[j - 1];
                j--;
            }
            lsym[j] = s;
        }
        for (uint32_t k = lo; k < list_len; k++) {
            codes[lsym[k]].code = fl_bit_reverse((uint16_t)code, n);
            codes[lsym[k]].len = (uint16_t)n;
            code = (code + 1) & 0xffffu;
        }
#endif
        list_len -= bits;
    }
}

FL_HD uint32_t fl_huff_bit_length(const fl_hcode* codes, const uint16_t* freq, uint32_t n) {
    uint32_t total = 0;  // huffman_encoder.zig:97-105
    FL_PLAN_FOR(i, 0, n)
        if (freq[i] != 0) total += (uint32_t)freq[i] * codes[i].len;
    FL_PLAN_REDUCE(total);
    return total;
}

// One run of `count` equal code lengths `size` as generateCodegen writes it (block_writer.zig:118-165): the bytes go to `out`
// (nullptr: only counted), the code-length codes used are counted in freq3 = {how often `size` itself, code 16, code 17 | 18 << 8}.
FL_HD uint32_t fl_codegen_run(uint32_t size, int32_t count, uint8_t* out, uint32_t* n_size, uint32_t* n_16, uint32_t* n_17, uint32_t* n_18) {
    uint32_t o = 0, k_size = 0, k16 = 0, k17 = 0, k18 = 0;
    if (size != 0) {
        if (out) out[o] = (uint8_t)size;
        o++;
        k_size++;
        count--;
        while (count >= 3) {
            const int32_t n = count < 6 ? count : 6;
            if (out) {
                out[o] = 16;
                out[o + 1] = (uint8_t)(n - 3);
            }
            o += 2;
            k16++;
            count -= n;
        }
    } else {
        while (count >= 11) {
            const int32_t n = count < 138 ? count : 138;
            if (out) {
                out[o] = 18;
                out[o + 1] = (uint8_t)(n - 11);
            }
            o += 2;
            k18++;
            count -= n;
        }
        if (count >= 3) {
            if (out) {
                out[o] = 17;
                out[o + 1] = (uint8_t)(count - 3);
            }
            o += 2;
            k17++;
            count = 0;
        }
    }
    for (; count > 0; count--) {
        if (out) out[o] = (uint8_t)size;
        o++;
        k_size++;
    }
    *n_size = k_size;
    *n_16 = k16;
    *n_17 = k17;
    *n_18 = k18;
    return o;
}

#if FL_PLAN_PARALLEL
// The same on the GPU, by the wave's lanes (round 5: the serial form below was 40 % of the planner's time on a block without
// matches -- 316 lengths, a dependent LDS load and a store each).  A RUN of equal lengths becomes its bytes independently of the
// others; where they go is a prefix sum over the runs.  Slot i = 64 k + lane holds length i: a run is written by the lane of
// its LAST length (run start by a running maximum over the run starts, offsets by a running sum of the runs' byte counts, both
// scans per group of 64 with a carry).  The lengths are read before anything is written: the output overwrites them.
static __device__ __forceinline__ uint32_t fl_plan_scan_max(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v = v > o ? v : o;
    }
    return v;
}
static __device__ __forceinline__ uint32_t fl_plan_scan_sum(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
FL_HD void fl_generate_codegen(fl_plan_ws* ws, uint32_t num_literals, uint32_t num_distances,
                               const fl_hcode* lit_codes, const fl_hcode* dist_codes) {
    const uint32_t lane = FL_PLAN_LANE();
    const uint32_t n = num_literals + num_distances;
    uint8_t* codegen = ws->codegen;
    uint32_t* cnt32 = ws->pm_p;  // (scratch of the Huffman construction: free here) 19 counters
    FL_PLAN_FOR(i, 0, num_literals) codegen[i] = (uint8_t)lit_codes[i].len;
    FL_PLAN_FOR(i, 0, num_distances) codegen[num_literals + i] = (uint8_t)dist_codes[i].len;
    FL_PLAN_FOR(i, 0, FL_NUM_CG) cnt32[i] = 0;
    FL_PLAN_SYNC();
    constexpr uint32_t G = (FL_NUM_LIT + FL_NUM_DIST + 63) / 64;  // groups of 64 slots
    uint32_t cur[G], start_at[G], nbytes[G], off[G];
    bool is_end[G];
    uint32_t carry_max = 0;
#pragma unroll
    for (uint32_t k = 0; k < G; k++) {
        const uint32_t i = 64 * k + lane;
        const bool in = i < n;
        cur[k] = in ? codegen[i] : 0xffffu;
        const uint32_t prev = (in && i) ? codegen[i - 1] : 0xfffeu, next = (in && i + 1 < n) ? codegen[i + 1] : 0xfffdu;
        is_end[k] = in && next != cur[k];
        const uint32_t st = (in && prev != cur[k]) ? i + 1 : 0u;  // (+ 1: 0 = not a start)
        const uint32_t m = fl_plan_scan_max(st, lane);
        start_at[k] = (m > carry_max ? m : carry_max) - 1u;  // the start of the run slot i lies in
        const uint32_t gmax = (uint32_t)__shfl((int)m, 63, 64);
        carry_max = gmax > carry_max ? gmax : carry_max;
    }
    uint32_t carry_sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < G; k++) {
        const uint32_t i = 64 * k + lane;
        uint32_t a, b, c, d;
        nbytes[k] = is_end[k] ? fl_codegen_run(cur[k], (int32_t)(i - start_at[k] + 1u), nullptr, &a, &b, &c, &d) : 0u;
        const uint32_t incl = fl_plan_scan_sum(nbytes[k], lane);
        off[k] = carry_sum + incl - nbytes[k];
        carry_sum += (uint32_t)__shfl((int)incl, 63, 64);
    }
    FL_PLAN_SYNC();  // (every lane has read its lengths)
#pragma unroll
    for (uint32_t k = 0; k < G; k++) {
        if (is_end[k]) {
            const uint32_t i = 64 * k + lane;
            uint32_t k_size, k16, k17, k18;
            (void)fl_codegen_run(cur[k], (int32_t)(i - start_at[k] + 1u), codegen + off[k], &k_size, &k16, &k17, &k18);
            if (k_size) atomicAdd(&cnt32[cur[k]], k_size);
            if (k16) atomicAdd(&cnt32[16], k16);
            if (k17) atomicAdd(&cnt32[17], k17);
            if (k18) atomicAdd(&cnt32[18], k18);
        }
    }
    if (lane == 0) codegen[carry_sum] = FL_END_MARK;
    FL_PLAN_SYNC();
    FL_PLAN_FOR(i, 0, FL_NUM_CG) ws->cg_freq[i] = (uint16_t)cnt32[i];
    FL_PLAN_SYNC();
}
#else
// block_writer.zig:78-171.  lit_lens / dist_lens are read through the code tables.
FL_HD void fl_generate_codegen(fl_plan_ws* ws, uint32_t num_literals, uint32_t num_distances,
                               const fl_hcode* lit_codes, const fl_hcode* dist_codes) {
    for (uint32_t i = 0; i < FL_NUM_CG; i++) ws->cg_freq[i] = 0;
    uint8_t* codegen = ws->codegen;
    for (uint32_t i = 0; i < num_literals; i++) codegen[i] = (uint8_t)lit_codes[i].len;
    for (uint32_t i = 0; i < num_distances; i++) codegen[num_literals + i] = (uint8_t)dist_codes[i].len;
    codegen[num_literals + num_distances] = FL_END_MARK;

    uint32_t size = codegen[0];
    int32_t count = 1;
    uint32_t out_index = 0;
    for (uint32_t in_index = 1; size != FL_END_MARK; in_index++) {
        uint32_t next_size = codegen[in_index];
        if (next_size == size) {
            count++;
            continue;
        }
        if (size != 0) {
            codegen[out_index++] = (uint8_t)size;
            ws->cg_freq[size]++;
            count--;
            while (count >= 3) {
                int32_t n = 6;
                if (n > count) n = count;
                codegen[out_index++] = 16;
                codegen[out_index++] = (uint8_t)(n - 3);
                ws->cg_freq[16]++;
                count -= n;
            }
        } else {
            while (count >= 11) {
                int32_t n = 138;
                if (n > count) n = count;
                codegen[out_index++] = 18;
                codegen[out_index++] = (uint8_t)(n - 11);
                ws->cg_freq[18]++;
                count -= n;
            }
            if (count >= 3) {
                codegen[out_index++] = 17;
                codegen[out_index++] = (uint8_t)(count - 3);
                ws->cg_freq[17]++;
                count = 0;
            }
        }
        count--;
        for (; count >= 0; count--) {
            codegen[out_index++] = (uint8_t)size;
            ws->cg_freq[size]++;
        }
        size = next_size;
        count = 1;
    }
    codegen[out_index] = FL_END_MARK;
}
#endif

// tiny LSB-first bit sink for the block header (bit_writer.zig:63-79 semantics)
struct fl_hdr_writer {
    uint8_t* buf;
    uint64_t acc;
    uint32_t nacc;
    uint32_t nbits;
};
FL_HD void fl_hdr_put(fl_hdr_writer* w, uint32_t v, uint32_t n) {
    w->acc |= (uint64_t)v << w->nacc;
    w->nacc += n;
    w->nbits += n;
    while (w->nacc >= 8) {
        *w->buf++ = (uint8_t)w->acc;
        w->acc >>= 8;
        w->nacc -= 8;
    }
}
FL_HD void fl_hdr_finish(fl_hdr_writer* w) {
    if (w->nacc) *w->buf++ = (uint8_t)w->acc;
}

// block_writer.zig:179-203 (without the extra-bits / body terms)
FL_HD uint32_t fl_dynamic_header_size(fl_plan_ws* ws, uint32_t* num_codegens_out) {
    uint32_t num_codegens = FL_NUM_CG;
    while (num_codegens > 4 && ws->cg_freq[fl_codegen_order(num_codegens - 1)] == 0) num_codegens--;
    *num_codegens_out = num_codegens;
    return 3 + 5 + 5 + 4 + 3 * num_codegens + fl_huff_bit_length(ws->cg_codes, ws->cg_freq, FL_NUM_CG) +
           (uint32_t)ws->cg_freq[16] * 2 + (uint32_t)ws->cg_freq[17] * 3 + (uint32_t)ws->cg_freq[18] * 7;
}

// block_writer.zig:237-281
FL_HD void fl_emit_dynamic_header(fl_plan_ws* ws, fl_hdr_writer* w, uint32_t num_literals,
                                  uint32_t num_distances, uint32_t num_codegens, uint32_t eof) {
    fl_hdr_put(w, eof ? 5u : 4u, 3);
    fl_hdr_put(w, num_literals - 257, 5);
    fl_hdr_put(w, num_distances - 1, 5);
    fl_hdr_put(w, num_codegens - 4, 4);
    for (uint32_t i = 0; i < num_codegens; i++) fl_hdr_put(w, ws->cg_codes[fl_codegen_order(i)].len, 3);
    uint32_t i = 0;
    for (;;) {
        uint32_t cw = ws->codegen[i++];
        if (cw == FL_END_MARK) break;
        fl_hdr_put(w, ws->cg_codes[cw].code, ws->cg_codes[cw].len);
        if (cw == 16) fl_hdr_put(w, ws->codegen[i++], 2);
        else if (cw == 17) fl_hdr_put(w, ws->codegen[i++], 3);
        else if (cw == 18) fl_hdr_put(w, ws->codegen[i++], 7);
    }
}

// size in bits a stored block occupies when it starts at absolute bit `off`
// (block_writer.zig:283-291: 3 header bits, pad to byte, LEN, NLEN, bytes)
FL_HD uint64_t fl_stored_end(uint64_t off, uint32_t len) {
    uint64_t p = (off + 3 + 7) & ~7ull;
    return p + 32 + 8ull * len;
}

// ---- planner for a token block: BlockWriter.write, block_writer.zig:307-383 ----
// ws->lit_freq / dist_freq hold the token histogram WITHOUT the end-of-block
// count (added here, :464).  `in_len` is the length of the optional raw input
// slice; FL_NO_INPUT stands for the Zig `null` (a window slide happened since the
// last flush, SlidingWindow.zig:119-123 -- never the case for chunks <= 65535
// bytes, but the reference's golden "-noinput" vectors exercise it).
// `dynamic_only` selects the reference's other token-block writer, BlockWriter.dynamicBlock
// (block_writer.zig:395-432: no fixed-code candidate, extra bits left out of the estimate, stored
// when the input does not shrink by 1/16).  The reference's compressor never calls it; its golden
// ".dyn" vectors do (block_writer.zig:645), and so does the debug seam flate_hip_debug_write_block.
#define FL_NO_INPUT 0xffffffffu
FL_HD void fl_plan_token_block(fl_plan_ws* ws, fl_block_plan* plan, uint32_t in_len, uint32_t eof,
                               bool dynamic_only = false) {
    ws->lit_freq[FL_EOB] += 1;
    uint32_t num_literals = FL_NUM_LIT;
    while (ws->lit_freq[num_literals - 1] == 0) num_literals--;
    uint32_t num_distances = FL_NUM_DIST;
    while (num_distances > 0 && ws->dist_freq[num_distances - 1] == 0) num_distances--;
    bool phantom_dist = false;
    if (num_distances == 0) {  // block_writer.zig:476-481
        ws->dist_freq[0] = 1;
        num_distances = 1;
        phantom_dist = true;
    }
    fl_huff_generate(ws, ws->lit_freq, FL_NUM_LIT, 15, ws->lit_codes);
    fl_huff_generate(ws, ws->dist_freq, FL_NUM_DIST, 15, ws->dist_codes);

    // storedSizeFits (:221-229)
    const bool storable = in_len != FL_NO_INPUT && in_len <= FL_MAX_STORE;
    const uint32_t stored_size = storable ? (in_len + 5) * 8 : 0;
    uint32_t real_extra_bits = 0;
    FL_PLAN_FOR(lc, 257 + 8, num_literals)
        real_extra_bits += (uint32_t)ws->lit_freq[lc] * fl_len_extra_bits(lc - 257);
    FL_PLAN_FOR(dc, 4, num_distances)
        real_extra_bits += (uint32_t)ws->dist_freq[dc] * fl_dist_extra_bits(dc);
    FL_PLAN_REDUCE(real_extra_bits);
    // the estimates only include the extra bits when a stored block is possible (:317-334)
    const uint32_t extra_bits = (storable && !dynamic_only) ? real_extra_bits : 0;
    // fixedSize (:206-211)
    uint32_t fixed_sum = 0;
    FL_PLAN_FOR(i, 0, FL_NUM_LIT)
        if (ws->lit_freq[i]) fixed_sum += (uint32_t)ws->lit_freq[i] * fl_fixed_lit_code(i).len;
    FL_PLAN_FOR(i, 0, FL_NUM_DIST)
        if (ws->dist_freq[i]) fixed_sum += (uint32_t)ws->dist_freq[i] * 5u;
    FL_PLAN_REDUCE(fixed_sum);
    const uint32_t fixed_bits = 3 + extra_bits + fixed_sum;
    uint32_t size = fixed_bits;
    uint32_t type = FL_BLOCK_FIXED;

    fl_generate_codegen(ws, num_literals, num_distances, ws->lit_codes, ws->dist_codes);
    fl_huff_generate(ws, ws->cg_freq, FL_NUM_CG, 7, ws->cg_codes);
    uint32_t num_codegens;
    uint32_t dyn_size = fl_dynamic_header_size(ws, &num_codegens) +
                        fl_huff_bit_length(ws->lit_codes, ws->lit_freq, FL_NUM_LIT) +
                        fl_huff_bit_length(ws->dist_codes, ws->dist_freq, FL_NUM_DIST) + extra_bits;
    if (dyn_size < size || dynamic_only) {  // ties go to fixed (:362)
        size = dyn_size;
        type = FL_BLOCK_DYNAMIC;
    }
    if (dynamic_only) {
        if (storable && stored_size < size + (size >> 4)) type = FL_BLOCK_STORED;  // :424
    } else if (storable && stored_size < size) {
        type = FL_BLOCK_STORED;  // ties go to Huffman (:369)
    }

    plan->type = type;
    // The reference's estimate counts the phantom distance symbol of a block
    // without matches (dist_freq[0] = 1); the emitted block does not contain it.
    plan->size_bits = size - extra_bits + real_extra_bits -
                      (phantom_dist ? (type == FL_BLOCK_DYNAMIC ? ws->dist_codes[0].len : 5u) : 0u);
    plan->final_block = eof;
    plan->in_len = in_len;
    fl_hdr_writer w;
    w.buf = plan->hdr;
    w.acc = 0;
    w.nacc = 0;
    w.nbits = 0;
    if (type == FL_BLOCK_DYNAMIC) {
        fl_emit_dynamic_header(ws, &w, num_literals, num_distances, num_codegens, eof);
        FL_PLAN_FOR(i, 0, FL_NUM_LIT) plan->lit[i] = ws->lit_codes[i];
        FL_PLAN_FOR(i, 0, FL_NUM_DIST) plan->dist[i] = ws->dist_codes[i];
    } else if (type == FL_BLOCK_FIXED) {
        fl_hdr_put(&w, eof ? 3u : 2u, 3);  // fixedHeader :293-300
        FL_PLAN_FOR(i, 0, FL_NUM_LIT) plan->lit[i] = fl_fixed_lit_code(i);
        FL_PLAN_FOR(i, 0, FL_NUM_DIST) plan->dist[i] = fl_fixed_dist_code(i);
    }
    fl_hdr_finish(&w);
    plan->hdr_nbits = w.nbits;
}

// ---- planner for a huffman-only block: huffmanBlock, block_writer.zig:524-572 ----
// ws->lit_freq[0..255] holds the byte histogram of the block's input.
FL_HD void fl_plan_huffman_block(fl_plan_ws* ws, fl_block_plan* plan, uint32_t in_len, uint32_t eof) {
    FL_PLAN_MARK(48);
    FL_PLAN_FOR(i, 256, FL_NUM_LIT) ws->lit_freq[i] = 0;
    FL_PLAN_SYNC();
    ws->lit_freq[FL_EOB] = 1;
    const uint32_t num_literals = FL_EOB + 1;
    const uint32_t num_distances = 1;
    // huff_distance (huffman_encoder.zig:340-348): symbol 0 with a 1-bit code
    FL_PLAN_FOR(i, 0, FL_NUM_DIST) {
        ws->dist_freq[i] = 0;
        ws->dist_codes[i].code = 0;
        ws->dist_codes[i].len = 0;
    }
    FL_PLAN_SYNC();
    ws->dist_freq[0] = 1;
    ws->dist_codes[0].len = 1;
    fl_huff_generate(ws, ws->lit_freq, FL_NUM_LIT, 15, ws->lit_codes);
    FL_PLAN_MARK(51);
    fl_generate_codegen(ws, num_literals, num_distances, ws->lit_codes, ws->dist_codes);
    FL_PLAN_MARK(52);
    fl_huff_generate(ws, ws->cg_freq, FL_NUM_CG, 7, ws->cg_codes);
    FL_PLAN_MARK(53);
    uint32_t num_codegens;
    uint32_t size = fl_dynamic_header_size(ws, &num_codegens) +
                    fl_huff_bit_length(ws->lit_codes, ws->lit_freq, FL_NUM_LIT) +
                    fl_huff_bit_length(ws->dist_codes, ws->dist_freq, FL_NUM_DIST);
    const bool storable = in_len <= FL_MAX_STORE;
    const uint32_t ssize = storable ? (in_len + 5) * 8 : 0;
    uint32_t type = FL_BLOCK_DYNAMIC;
    if (storable && ssize < (size + (size >> 4))) type = FL_BLOCK_STORED;  // :558

    plan->type = type;
    // the estimate counts one phantom distance bit (dist_freq[0] = 1, :531); the
    // emitted block is header + literal codes + end-of-block
    plan->size_bits = size - 1;
    plan->final_block = eof;
    plan->in_len = in_len;
    fl_hdr_writer w;
    w.buf = plan->hdr;
    w.acc = 0;
    w.nacc = 0;
    w.nbits = 0;
    if (type == FL_BLOCK_DYNAMIC) {
        fl_emit_dynamic_header(ws, &w, num_literals, num_distances, num_codegens, eof);
        FL_PLAN_FOR(i, 0, FL_NUM_LIT) plan->lit[i] = ws->lit_codes[i];
    }
    fl_hdr_finish(&w);
    plan->hdr_nbits = w.nbits;
    FL_PLAN_MARK(54);
}
