/*
 * flate_oracle.c -- CPU oracle for the DEFLATE hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of ianic/flate's algorithm (reference under
 * /root/reference, Zig).  Not a product path: only tests/, smoke() and
 * bench.py's cpu_baseline leg may use it (see flate_oracle.h).
 * Parity status: PINNED against the reference's own golden vectors
 * (tests/test_oracle_*.py).
 *
 * Structure follows the reference so that each piece can be audited:
 *   sink / checksums          container.zig:168-206 (+ Zig std Crc32/Adler32)
 *   token code tables         Token.zig:58-276
 *   bit writer                bit_writer.zig:10-99
 *   Huffman code builder      huffman_encoder.zig:62-348
 *   block writer              block_writer.zig:78-585
 *   hash chains               Lookup.zig:12-84
 *   sliding window            SlidingWindow.zig:18-123
 *   tokenizer                 deflate.zig:121-373
 *   huffman/store compressors deflate.zig:449-529
 *   inflate                   inflate.zig:43-355, huffman_decoder.zig:71-175,
 *                             bit_reader.zig:18-219, CircularBuffer.zig:44-75
 */
#include "flate_oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ sink */

static void sink_reserve(fo_sink* s, size_t extra) {
    if (s->len + extra <= s->cap) return;
    size_t ncap = s->cap ? s->cap * 2 : 4096;
    while (ncap < s->len + extra) ncap *= 2;
    s->data = (uint8_t*)realloc(s->data, ncap);
    if (!s->data) abort();
    s->cap = ncap;
}
static void sink_write(fo_sink* s, const uint8_t* p, size_t n) {
    if (n == 0) return;
    sink_reserve(s, n);
    memcpy(s->data + s->len, p, n);
    s->len += n;
}
void fo_sink_free(fo_sink* s) {
    free(s->data);
    s->data = NULL;
    s->len = s->cap = 0;
}

/* ------------------------------------------------------------- checksums */
/* CRC-32/IEEE 802.3 (reflected 0xEDB88320) and Adler-32: what Zig std's
 * std.hash.Crc32 / std.hash.Adler32 compute (container.zig:170-171).  Pins:
 * flate.zig:370,375; inflate.zig:409,431,466. */

static uint32_t crc_table[256];
static int crc_table_ready = 0;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_table[i] = c;
    }
    crc_table_ready = 1;
}
uint32_t fo_crc32(uint32_t crc, const uint8_t* p, size_t n) {
    if (!crc_table_ready) crc_init();
    uint32_t c = crc ^ 0xffffffffu;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
uint32_t fo_adler32(uint32_t adler, const uint8_t* p, size_t n) {
    uint32_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
    while (n > 0) {
        size_t k = n < 5552 ? n : 5552;
        for (size_t i = 0; i < k; i++) {
            a += p[i];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        p += k;
        n -= k;
    }
    return (b << 16) | a;
}

/* container.zig:168-206 */
typedef struct {
    int container;
    uint32_t state; /* crc or adler running value */
    uint64_t bytes;
} hasher_t;
static void hasher_init(hasher_t* h, int container) {
    h->container = container;
    h->state = (container == FO_ZLIB) ? 1u : 0u;
    h->bytes = 0;
}
static void hasher_update(hasher_t* h, const uint8_t* p, size_t n) {
    if (h->container == FO_GZIP)
        h->state = fo_crc32(h->state, p, n);
    else if (h->container == FO_ZLIB)
        h->state = fo_adler32(h->state, p, n);
    else
        return;
    h->bytes += n;
}

/* container.zig:53-83 */
static void write_container_header(int container, fo_sink* s) {
    if (container == FO_GZIP) {
        static const uint8_t h[10] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0x03};
        sink_write(s, h, 10);
    } else if (container == FO_ZLIB) {
        static const uint8_t h[2] = {0x78, 0x9c};
        sink_write(s, h, 2);
    }
}
/* container.zig:85-109 */
static void write_container_footer(hasher_t* h, fo_sink* s) {
    uint8_t b[8];
    if (h->container == FO_GZIP) {
        uint32_t c = h->state, n = (uint32_t)h->bytes;
        b[0] = c; b[1] = c >> 8; b[2] = c >> 16; b[3] = c >> 24;
        b[4] = n; b[5] = n >> 8; b[6] = n >> 16; b[7] = n >> 24;
        sink_write(s, b, 8);
    } else if (h->container == FO_ZLIB) {
        uint32_t c = h->state;
        b[0] = c >> 24; b[1] = c >> 16; b[2] = c >> 8; b[3] = c;
        sink_write(s, b, 4);
    }
}

/* ------------------------------------------------------ token code tables */
/* RFC 1951 3.2.5 tables as used by Token.zig:114-276. */

static const uint8_t len_base_scaled[29] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,
                                            12, 14, 16, 20, 24, 28, 32, 40, 48, 56,
                                            64, 80, 96, 112, 128, 160, 192, 224, 255};
static const uint8_t len_extra_bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                           2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base_scaled[30] = {
    0x0000, 0x0001, 0x0002, 0x0003, 0x0004, 0x0006, 0x0008, 0x000c, 0x0010, 0x0018,
    0x0020, 0x0030, 0x0040, 0x0060, 0x0080, 0x00c0, 0x0100, 0x0180, 0x0200, 0x0300,
    0x0400, 0x0600, 0x0800, 0x0c00, 0x1000, 0x1800, 0x2000, 0x3000, 0x4000, 0x6000};
static const uint8_t dist_extra_bits[30] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,  4,  5,  5,  6,
                                            6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint8_t len_index[256];   /* Token.zig:114-141 match_lengths_index */
static uint8_t dist_index[32768]; /* Token.zig:70-81 distanceCode, flattened */
static int tok_tables_ready = 0;
static void tok_tables_init(void) {
    for (int v = 0; v < 256; v++) {
        int idx = 0;
        for (int i = 0; i < 29; i++)
            if (len_base_scaled[i] <= v) idx = i;
        len_index[v] = (uint8_t)idx;
    }
    for (int d = 0; d < 32768; d++) {
        int idx = 0;
        for (int i = 0; i < 30; i++)
            if (dist_base_scaled[i] <= d) idx = i;
        dist_index[d] = (uint8_t)idx;
    }
    tok_tables_ready = 1;
}
#define TOK_INIT() do { if (!tok_tables_ready) tok_tables_init(); } while (0)

uint16_t fo_length_code(uint8_t len_lit) { TOK_INIT(); return (uint16_t)(257 + len_index[len_lit]); }
uint8_t fo_distance_code(uint16_t dist0) { TOK_INIT(); return dist_index[dist0 & 0x7fff]; }
uint8_t fo_length_extra_bits(uint16_t code) { return len_extra_bits[code - 257]; }
uint8_t fo_distance_extra_bits(uint8_t code) { return dist_extra_bits[code]; }

/* ------------------------------------------------------------ bit writer */
/* bit_writer.zig:63-97.  The reference's 48-bit spill / 240-byte staging is
 * not observable; the output is the LSB-first bit concatenation, zero padded
 * to a byte at every flush. */
typedef struct {
    fo_sink* out;
    uint64_t bits;
    uint32_t nbits;
} bitw_t;
static void bw_write_bits(bitw_t* w, uint32_t b, uint32_t nb) {
    w->bits |= (uint64_t)b << w->nbits;
    w->nbits += nb;
    if (w->nbits >= 32) { /* spill 4 whole bytes at a time (the reference spills 6, :66-78) */
        fo_sink* s = w->out;
        if (s->len + 4 > s->cap) sink_reserve(s, 4);
        uint32_t v = (uint32_t)w->bits;
        s->data[s->len] = (uint8_t)v;
        s->data[s->len + 1] = (uint8_t)(v >> 8);
        s->data[s->len + 2] = (uint8_t)(v >> 16);
        s->data[s->len + 3] = (uint8_t)(v >> 24);
        s->len += 4;
        w->bits >>= 32;
        w->nbits -= 32;
    }
}
/* bit_writer.zig:46-61 */
static void bw_flush(bitw_t* w) {
    while (w->nbits != 0) {
        uint8_t byte = (uint8_t)w->bits;
        sink_write(w->out, &byte, 1);
        w->bits >>= 8;
        w->nbits = w->nbits > 8 ? w->nbits - 8 : 0;
    }
    w->bits = 0;
}
/* bit_writer.zig:81-97 (UnfinishedBits cannot happen: callers flush first) */
static void bw_write_bytes(bitw_t* w, const uint8_t* p, size_t n) {
    if (w->nbits & 7) abort();
    bw_flush(w); /* whole buffered bytes first (bit_writer.zig:86-91) */
    sink_write(w->out, p, n);
}

/* ---------------------------------------------------- Huffman code builder */
/* huffman_encoder.zig */
typedef struct { uint16_t code, len; } hcode_t;
typedef struct { hcode_t codes[286]; } henc_t;
typedef struct { uint16_t literal, freq; } lnode_t;

static uint16_t bit_reverse16(uint16_t v, unsigned n) { /* huffman_encoder.zig:455-458 */
    uint16_t r = 0;
    for (unsigned i = 0; i < 16; i++)
        if (v & (1u << i)) r |= (uint16_t)(1u << (15 - i));
    return (uint16_t)(r >> (16 - n));
}

static int cmp_by_freq(const void* a, const void* b) { /* huffman_encoder.zig:355-361 */
    const lnode_t* x = (const lnode_t*)a;
    const lnode_t* y = (const lnode_t*)b;
    if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
    return (x->literal > y->literal) - (x->literal < y->literal);
}
static int cmp_by_literal(const void* a, const void* b) { /* huffman_encoder.zig:350-353 */
    const lnode_t* x = (const lnode_t*)a;
    const lnode_t* y = (const lnode_t*)b;
    return (x->literal > y->literal) - (x->literal < y->literal);
}

/* huffman_encoder.zig:122-247.  Quirk Q3 kept: the exhausted-leaf sentinel is
 * 65535 (maxInt(u16), :189,282-287) while the "ran out of both" test compares
 * against maxInt(i32) (:170), so that test never fires; all comparisons are
 * strict `<` on u32. */
static void huff_bit_counts(const lnode_t* list, uint32_t n, uint32_t max_bits,
                            uint32_t* bit_count /* [17] */) {
    struct level_info {
        uint32_t level, last_freq, next_char_freq, next_pair_freq, needed;
    } levels[18];
    uint32_t leaf_counts[17][16];
    memset(levels, 0, sizeof levels);
    memset(leaf_counts, 0, sizeof leaf_counts);

    if (max_bits > n - 1) max_bits = n - 1;

    for (uint32_t level = 1; level <= max_bits; level++) {
        levels[level].level = level;
        levels[level].last_freq = list[1].freq;
        levels[level].next_char_freq = list[2].freq;
        levels[level].next_pair_freq = (uint32_t)list[0].freq + list[1].freq;
        levels[level].needed = 0;
        leaf_counts[level][level] = 2;
        if (level == 1) levels[level].next_pair_freq = INT32_MAX;
    }
    levels[max_bits].needed = 2 * n - 4;

    uint32_t level = max_bits;
    for (;;) {
        struct level_info* l = &levels[level];
        if (l->next_pair_freq == (uint32_t)INT32_MAX && l->next_char_freq == (uint32_t)INT32_MAX) {
            l->needed = 0;
            levels[level + 1].next_pair_freq = INT32_MAX;
            level += 1;
            continue;
        }
        uint32_t prev_freq = l->last_freq;
        if (l->next_char_freq < l->next_pair_freq) {
            uint32_t next = leaf_counts[level][level] + 1;
            l->last_freq = l->next_char_freq;
            leaf_counts[level][level] = next;
            l->next_char_freq = (next >= n) ? 65535u : list[next].freq;
        } else {
            l->last_freq = l->next_pair_freq;
            memcpy(leaf_counts[level], leaf_counts[level - 1], level * sizeof(uint32_t));
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
    if (leaf_counts[max_bits][max_bits] != n) { /* reference asserts this (:229) */
        fprintf(stderr, "flate_oracle: bitCounts invariant broken (n=%u)\n", n);
        abort();
    }
    uint32_t bits = 1;
    const uint32_t* counts = leaf_counts[max_bits];
    for (uint32_t lv = max_bits; lv > 0; lv--) {
        bit_count[bits] = counts[lv] - counts[lv - 1];
        bits++;
    }
    bit_count[0] = 0;
    for (uint32_t i = max_bits + 1; i < 17; i++) bit_count[i] = 0;
}

/* huffman_encoder.zig:62-95 + 251-278 */
static void henc_generate(henc_t* e, const uint16_t* freq, int nfreq, uint32_t max_bits) {
    lnode_t list[287];
    uint32_t count = 0;
    for (int i = 0; i < nfreq; i++) {
        if (freq[i] != 0) {
            list[count].literal = (uint16_t)i;
            list[count].freq = freq[i];
            count++;
        } else {
            e->codes[i].len = 0;
        }
    }
    if (count <= 2) {
        for (uint32_t i = 0; i < count; i++) {
            e->codes[list[i].literal].code = (uint16_t)i;
            e->codes[list[i].literal].len = 1;
        }
        return;
    }
    qsort(list, count, sizeof(lnode_t), cmp_by_freq);

    uint32_t bit_count[17];
    huff_bit_counts(list, count, max_bits, bit_count);
    uint32_t used_bits = max_bits > count - 1 ? count - 1 : max_bits;

    uint16_t code = 0;
    uint32_t list_len = count;
    for (uint32_t n = 0; n <= used_bits; n++) {
        code = (uint16_t)(code << 1);
        uint32_t bits = bit_count[n];
        if (n == 0 || bits == 0) continue;
        lnode_t* chunk = list + (list_len - bits);
        qsort(chunk, bits, sizeof(lnode_t), cmp_by_literal);
        for (uint32_t k = 0; k < bits; k++) {
            e->codes[chunk[k].literal].code = bit_reverse16(code, n);
            e->codes[chunk[k].literal].len = (uint16_t)n;
            code++;
        }
        list_len -= bits;
    }
}

/* huffman_encoder.zig:97-105 */
static uint32_t henc_bit_length(const henc_t* e, const uint16_t* freq, int nfreq) {
    uint32_t total = 0;
    for (int i = 0; i < nfreq; i++)
        if (freq[i] != 0) total += (uint32_t)freq[i] * e->codes[i].len;
    return total;
}

/* huffman_encoder.zig:298-330 */
static void henc_fixed_literal(henc_t* e) {
    for (uint16_t ch = 0; ch < 286; ch++) {
        uint16_t bits, size;
        if (ch <= 143) { bits = ch + 48; size = 8; }
        else if (ch <= 255) { bits = ch + 400 - 144; size = 9; }
        else if (ch <= 279) { bits = ch - 256; size = 7; }
        else { bits = ch + 192 - 280; size = 8; }
        e->codes[ch].code = bit_reverse16(bits, size);
        e->codes[ch].len = size;
    }
}
/* huffman_encoder.zig:332-338 */
static void henc_fixed_distance(henc_t* e) {
    for (uint16_t ch = 0; ch < 30; ch++) {
        e->codes[ch].code = bit_reverse16(ch, 5);
        e->codes[ch].len = 5;
    }
}
/* huffman_encoder.zig:340-348 */
static void henc_huffman_distance(henc_t* e) {
    uint16_t freq[30];
    memset(freq, 0, sizeof freq);
    freq[0] = 1;
    memset(e, 0, sizeof *e);
    henc_generate(e, freq, 30, 15);
}

void fo_huffman_generate(const uint16_t* freq, int n, int max_bits, uint16_t* codes,
                         uint16_t* lens) {
    henc_t e;
    memset(&e, 0, sizeof e);
    henc_generate(&e, freq, n, (uint32_t)max_bits);
    for (int i = 0; i < n; i++) {
        codes[i] = e.codes[i].len ? e.codes[i].code : 0;
        lens[i] = e.codes[i].len;
    }
}
void fo_fixed_literal_codes(uint16_t codes[286], uint16_t lens[286]) {
    henc_t e;
    henc_fixed_literal(&e);
    for (int i = 0; i < 286; i++) {
        codes[i] = e.codes[i].code;
        lens[i] = e.codes[i].len;
    }
}

/* ------------------------------------------------------------ block writer */
/* block_writer.zig */
static const uint8_t codegen_order[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                          11, 4,  12, 3, 13, 2, 14, 1, 15}; /* consts.zig:30 */
#define END_CODE_MARK 255
#define END_BLOCK_MARKER 256
#define MAX_STORE_BLOCK 65535

typedef struct {
    bitw_t bw;
    uint16_t codegen_freq[19];
    uint16_t literal_freq[286];
    uint16_t distance_freq[30];
    uint8_t codegen[286 + 30 + 1];
    henc_t literal_encoding, distance_encoding, codegen_encoding;
    henc_t fixed_literal_encoding, fixed_distance_encoding, huff_distance;
    /* test seam: token log (deflate.zig:578-608) */
    int log_tokens;
    uint32_t* tok_log;
    size_t tok_log_len, tok_log_cap;
} blockw_t;

static void blockw_init(blockw_t* b, fo_sink* out) { /* block_writer.zig:38-45 */
    memset(b, 0, sizeof *b);
    b->bw.out = out;
    henc_fixed_literal(&b->fixed_literal_encoding);
    henc_fixed_distance(&b->fixed_distance_encoding);
    henc_huffman_distance(&b->huff_distance);
    TOK_INIT();
}

static void blockw_write_code(blockw_t* b, hcode_t c) { bw_write_bits(&b->bw, c.code, c.len); }

/* block_writer.zig:78-171 */
static void blockw_generate_codegen(blockw_t* b, uint32_t num_literals, uint32_t num_distances,
                                    const henc_t* lit_enc, const henc_t* dist_enc) {
    memset(b->codegen_freq, 0, sizeof b->codegen_freq);
    uint8_t* codegen = b->codegen;
    for (uint32_t i = 0; i < num_literals; i++) codegen[i] = (uint8_t)lit_enc->codes[i].len;
    for (uint32_t i = 0; i < num_distances; i++)
        codegen[num_literals + i] = (uint8_t)dist_enc->codes[i].len;
    codegen[num_literals + num_distances] = END_CODE_MARK;

    uint8_t size = codegen[0];
    int32_t count = 1;
    uint32_t out_index = 0;
    for (uint32_t in_index = 1; size != END_CODE_MARK; in_index++) {
        uint8_t next_size = codegen[in_index];
        if (next_size == size) {
            count++;
            continue;
        }
        if (size != 0) {
            codegen[out_index++] = size;
            b->codegen_freq[size]++;
            count--;
            while (count >= 3) {
                int32_t n = 6;
                if (n > count) n = count;
                codegen[out_index++] = 16;
                codegen[out_index++] = (uint8_t)(n - 3);
                b->codegen_freq[16]++;
                count -= n;
            }
        } else {
            while (count >= 11) {
                int32_t n = 138;
                if (n > count) n = count;
                codegen[out_index++] = 18;
                codegen[out_index++] = (uint8_t)(n - 11);
                b->codegen_freq[18]++;
                count -= n;
            }
            if (count >= 3) {
                codegen[out_index++] = 17;
                codegen[out_index++] = (uint8_t)(count - 3);
                b->codegen_freq[17]++;
                count = 0;
            }
        }
        count--;
        for (; count >= 0; count--) {
            codegen[out_index++] = size;
            b->codegen_freq[size]++;
        }
        size = next_size;
        count = 1;
    }
    codegen[out_index] = END_CODE_MARK;
}

/* block_writer.zig:179-203 */
static uint32_t blockw_dynamic_size(blockw_t* b, const henc_t* lit_enc, const henc_t* dist_enc,
                                    uint32_t extra_bits, uint32_t* num_codegens_out) {
    uint32_t num_codegens = 19;
    while (num_codegens > 4 && b->codegen_freq[codegen_order[num_codegens - 1]] == 0)
        num_codegens--;
    uint32_t header = 3 + 5 + 5 + 4 + 3 * num_codegens +
                      henc_bit_length(&b->codegen_encoding, b->codegen_freq, 19) +
                      (uint32_t)b->codegen_freq[16] * 2 + (uint32_t)b->codegen_freq[17] * 3 +
                      (uint32_t)b->codegen_freq[18] * 7;
    uint32_t size = header + henc_bit_length(lit_enc, b->literal_freq, 286) +
                    henc_bit_length(dist_enc, b->distance_freq, 30) + extra_bits;
    *num_codegens_out = num_codegens;
    return size;
}

/* block_writer.zig:206-211 */
static uint32_t blockw_fixed_size(blockw_t* b, uint32_t extra_bits) {
    return 3 + henc_bit_length(&b->fixed_literal_encoding, b->literal_freq, 286) +
           henc_bit_length(&b->fixed_distance_encoding, b->distance_freq, 30) + extra_bits;
}

/* block_writer.zig:237-281 */
static void blockw_dynamic_header(blockw_t* b, uint32_t num_literals, uint32_t num_distances,
                                  uint32_t num_codegens, int eof) {
    bw_write_bits(&b->bw, eof ? 5 : 4, 3);
    bw_write_bits(&b->bw, num_literals - 257, 5);
    bw_write_bits(&b->bw, num_distances - 1, 5);
    bw_write_bits(&b->bw, num_codegens - 4, 4);
    for (uint32_t i = 0; i < num_codegens; i++)
        bw_write_bits(&b->bw, b->codegen_encoding.codes[codegen_order[i]].len, 3);
    uint32_t i = 0;
    for (;;) {
        uint32_t code_word = b->codegen[i++];
        if (code_word == END_CODE_MARK) break;
        blockw_write_code(b, b->codegen_encoding.codes[code_word]);
        switch (code_word) {
            case 16: bw_write_bits(&b->bw, b->codegen[i++], 2); break;
            case 17: bw_write_bits(&b->bw, b->codegen[i++], 3); break;
            case 18: bw_write_bits(&b->bw, b->codegen[i++], 7); break;
            default: break;
        }
    }
}

/* block_writer.zig:283-291, 385-388 */
static void blockw_stored_block(blockw_t* b, const uint8_t* input, size_t len, int eof) {
    if (len > 65535) abort();
    bw_write_bits(&b->bw, eof ? 1 : 0, 3);
    bw_flush(&b->bw);
    uint16_t l = (uint16_t)len;
    bw_write_bits(&b->bw, l, 16);
    bw_write_bits(&b->bw, (uint16_t)~l, 16);
    bw_write_bytes(&b->bw, input, len);
}

/* block_writer.zig:444-488 */
static void blockw_index_tokens(blockw_t* b, const uint32_t* tokens, size_t ntok,
                                uint32_t* num_literals_out, uint32_t* num_distances_out) {
    memset(b->literal_freq, 0, sizeof b->literal_freq);
    memset(b->distance_freq, 0, sizeof b->distance_freq);
    for (size_t i = 0; i < ntok; i++) {
        uint32_t t = tokens[i];
        if (!FO_TOK_IS_MATCH(t)) {
            b->literal_freq[FO_TOK_LENLIT(t)]++;
            continue;
        }
        b->literal_freq[257 + len_index[FO_TOK_LENLIT(t)]]++;
        b->distance_freq[dist_index[FO_TOK_DIST0(t)]]++;
    }
    b->literal_freq[END_BLOCK_MARKER]++;
    uint32_t num_literals = 286;
    while (b->literal_freq[num_literals - 1] == 0) num_literals--;
    uint32_t num_distances = 30;
    while (num_distances > 0 && b->distance_freq[num_distances - 1] == 0) num_distances--;
    if (num_distances == 0) {
        b->distance_freq[0] = 1;
        num_distances = 1;
    }
    henc_generate(&b->literal_encoding, b->literal_freq, 286, 15);
    henc_generate(&b->distance_encoding, b->distance_freq, 30, 15);
    *num_literals_out = num_literals;
    *num_distances_out = num_distances;
}

/* block_writer.zig:492-520 */
static void blockw_write_tokens(blockw_t* b, const uint32_t* tokens, size_t ntok,
                                const hcode_t* le_codes, const hcode_t* oe_codes) {
    for (size_t i = 0; i < ntok; i++) {
        uint32_t t = tokens[i];
        if (!FO_TOK_IS_MATCH(t)) {
            blockw_write_code(b, le_codes[FO_TOK_LENLIT(t)]);
            continue;
        }
        uint32_t ll = FO_TOK_LENLIT(t);
        uint32_t li = len_index[ll];
        blockw_write_code(b, le_codes[257 + li]);
        if (len_extra_bits[li] > 0) bw_write_bits(&b->bw, ll - len_base_scaled[li], len_extra_bits[li]);
        uint32_t d = FO_TOK_DIST0(t);
        uint32_t di = dist_index[d];
        blockw_write_code(b, oe_codes[di]);
        if (dist_extra_bits[di] > 0) bw_write_bits(&b->bw, d - dist_base_scaled[di], dist_extra_bits[di]);
    }
    blockw_write_code(b, le_codes[END_BLOCK_MARKER]);
}

static void blockw_log(blockw_t* b, const uint32_t* tokens, size_t ntok) {
    if (!b->log_tokens || ntok == 0) return;
    if (b->tok_log_len + ntok > b->tok_log_cap) {
        size_t ncap = b->tok_log_cap ? b->tok_log_cap * 2 : 65536;
        while (ncap < b->tok_log_len + ntok) ncap *= 2;
        b->tok_log = (uint32_t*)realloc(b->tok_log, ncap * sizeof(uint32_t));
        if (!b->tok_log) abort();
        b->tok_log_cap = ncap;
    }
    memcpy(b->tok_log + b->tok_log_len, tokens, ntok * sizeof(uint32_t));
    b->tok_log_len += ntok;
}

/* block_writer.zig:307-383.  input == NULL <=> Zig null. */
static void blockw_write(blockw_t* b, const uint32_t* tokens, size_t ntok, int eof,
                         const uint8_t* input, size_t input_len, int has_input) {
    blockw_log(b, tokens, ntok);
    uint32_t num_literals, num_distances;
    blockw_index_tokens(b, tokens, ntok, &num_literals, &num_distances);

    uint32_t extra_bits = 0;
    /* storedSizeFits, block_writer.zig:221-229 */
    int storable = has_input && input_len <= MAX_STORE_BLOCK;
    uint32_t stored_size = storable ? (uint32_t)((input_len + 5) * 8) : 0;

    if (storable) {
        for (uint32_t lc = 257 + 8; lc < num_literals; lc++)
            extra_bits += (uint32_t)b->literal_freq[lc] * len_extra_bits[lc - 257];
        for (uint32_t dc = 4; dc < num_distances; dc++)
            extra_bits += (uint32_t)b->distance_freq[dc] * dist_extra_bits[dc];
    }

    const henc_t* literal_encoding = &b->fixed_literal_encoding;
    const henc_t* distance_encoding = &b->fixed_distance_encoding;
    uint32_t size = blockw_fixed_size(b, extra_bits);

    uint32_t num_codegens = 0;
    blockw_generate_codegen(b, num_literals, num_distances, &b->literal_encoding,
                            &b->distance_encoding);
    henc_generate(&b->codegen_encoding, b->codegen_freq, 19, 7);
    uint32_t dyn_size = blockw_dynamic_size(b, &b->literal_encoding, &b->distance_encoding,
                                            extra_bits, &num_codegens);
    if (dyn_size < size) {
        size = dyn_size;
        literal_encoding = &b->literal_encoding;
        distance_encoding = &b->distance_encoding;
    }
    if (storable && stored_size < size) {
        blockw_stored_block(b, input, input_len, eof);
        return;
    }
    if (literal_encoding == &b->fixed_literal_encoding)
        bw_write_bits(&b->bw, eof ? 3 : 2, 3); /* fixedHeader :293-300 */
    else
        blockw_dynamic_header(b, num_literals, num_distances, num_codegens, eof);
    blockw_write_tokens(b, tokens, ntok, literal_encoding->codes, distance_encoding->codes);
}

/* block_writer.zig:395-433 */
static void blockw_dynamic_block(blockw_t* b, const uint32_t* tokens, size_t ntok, int eof,
                                 const uint8_t* input, size_t input_len, int has_input) {
    uint32_t num_literals, num_distances, num_codegens;
    blockw_index_tokens(b, tokens, ntok, &num_literals, &num_distances);
    blockw_generate_codegen(b, num_literals, num_distances, &b->literal_encoding,
                            &b->distance_encoding);
    henc_generate(&b->codegen_encoding, b->codegen_freq, 19, 7);
    uint32_t size = blockw_dynamic_size(b, &b->literal_encoding, &b->distance_encoding, 0,
                                        &num_codegens);
    int storable = has_input && input_len <= MAX_STORE_BLOCK;
    uint32_t ssize = storable ? (uint32_t)((input_len + 5) * 8) : 0;
    if (storable && ssize < (size + (size >> 4))) {
        blockw_stored_block(b, input, input_len, eof);
        return;
    }
    blockw_dynamic_header(b, num_literals, num_distances, num_codegens, eof);
    blockw_write_tokens(b, tokens, ntok, b->literal_encoding.codes, b->distance_encoding.codes);
}

/* block_writer.zig:524-585 */
static void blockw_huffman_block(blockw_t* b, const uint8_t* input, size_t input_len, int eof) {
    memset(b->literal_freq, 0, sizeof b->literal_freq);
    for (size_t i = 0; i < input_len; i++) b->literal_freq[input[i]]++;
    b->literal_freq[END_BLOCK_MARKER] = 1;
    const uint32_t num_literals = END_BLOCK_MARKER + 1;
    b->distance_freq[0] = 1;
    const uint32_t num_distances = 1;

    henc_generate(&b->literal_encoding, b->literal_freq, 286, 15);
    uint32_t num_codegens = 0;
    blockw_generate_codegen(b, num_literals, num_distances, &b->literal_encoding,
                            &b->huff_distance);
    henc_generate(&b->codegen_encoding, b->codegen_freq, 19, 7);
    uint32_t size = blockw_dynamic_size(b, &b->literal_encoding, &b->huff_distance, 0,
                                        &num_codegens);
    int storable = input_len <= MAX_STORE_BLOCK;
    uint32_t ssize = storable ? (uint32_t)((input_len + 5) * 8) : 0;
    if (storable && ssize < (size + (size >> 4))) {
        blockw_stored_block(b, input, input_len, eof);
        return;
    }
    blockw_dynamic_header(b, num_literals, num_distances, num_codegens, eof);
    const hcode_t* encoding = b->literal_encoding.codes;
    for (size_t i = 0; i < input_len; i++) bw_write_bits(&b->bw, encoding[input[i]].code, encoding[input[i]].len);
    blockw_write_code(b, encoding[END_BLOCK_MARKER]);
}

int fo_block_write(int fn, const uint32_t* tokens, size_t ntok, int eof, const uint8_t* input,
                   size_t input_len, int has_input, uint8_t* out, size_t cap, size_t* out_len) {
    fo_sink s = {0};
    blockw_t* b = (blockw_t*)malloc(sizeof *b);
    blockw_init(b, &s);
    if (fn == 0)
        blockw_write(b, tokens, ntok, eof, input, input_len, has_input);
    else if (fn == 1)
        blockw_dynamic_block(b, tokens, ntok, eof, input, input_len, has_input);
    else
        blockw_huffman_block(b, input, input_len, eof);
    bw_flush(&b->bw);
    free(b);
    int rc = 0;
    *out_len = s.len;
    if (s.len > cap)
        rc = FO_OUTPUT_TOO_SMALL;
    else if (s.len)
        memcpy(out, s.data, s.len);
    fo_sink_free(&s);
    return rc;
}

/* ------------------------------------------------------------ hash chains */
/* Lookup.zig */
#define HIST_LEN 32768u
#define WIN_LEN 65536u
#define MIN_MATCH 4u
#define MAX_MATCH 258u
#define MIN_LOOKAHEAD (MIN_MATCH + MAX_MATCH)
#define MAX_RP (WIN_LEN - MIN_LOOKAHEAD)
#define TOKENS_MAX 32768u

typedef struct {
    uint16_t head[32768];
    uint16_t chain[65536];
} lookup_t;

static inline uint32_t hashu(uint32_t v) { return (v * 0x9E3779B1u) >> 17; } /* Lookup.zig:82-84 */
uint32_t fo_hash4(const uint8_t* b) { /* Lookup.zig:75-80 */
    return hashu((uint32_t)b[3] | (uint32_t)b[2] << 8 | (uint32_t)b[1] << 16 | (uint32_t)b[0] << 24);
}
static inline uint16_t lookup_set(lookup_t* l, uint32_t h, uint16_t pos) { /* Lookup.zig:35-40 */
    uint16_t p = l->head[h];
    l->head[h] = pos;
    l->chain[pos] = p;
    return p;
}
static inline uint16_t lookup_add(lookup_t* l, const uint8_t* data, size_t len, uint16_t pos) {
    if (len < 4) return 0; /* Lookup.zig:23-27 */
    return lookup_set(l, fo_hash4(data), pos);
}
static void lookup_slide(lookup_t* l, uint16_t n) { /* Lookup.zig:43-51 */
    for (uint32_t i = 0; i < 32768; i++) l->head[i] = l->head[i] > n ? (uint16_t)(l->head[i] - n) : 0;
    for (uint32_t i = 0; i < n; i++) {
        uint16_t v = l->chain[i + n];
        l->chain[i] = v > n ? (uint16_t)(v - n) : 0;
    }
}
static void lookup_bulk_add(lookup_t* l, const uint8_t* data, size_t data_len, uint16_t len,
                            uint16_t pos) { /* Lookup.zig:55-72 */
    if (len == 0 || data_len < MIN_MATCH) return;
    uint32_t hb = (uint32_t)data[3] | (uint32_t)data[2] << 8 | (uint32_t)data[1] << 16 |
                  (uint32_t)data[0] << 24;
    lookup_set(l, hashu(hb), pos);
    uint16_t i = pos;
    size_t end = (size_t)len + 3 < data_len ? (size_t)len + 3 : data_len;
    for (size_t j = 4; j < end; j++) {
        hb = (hb << 8) | data[j];
        i++;
        lookup_set(l, hashu(hb), i);
    }
}

void fo_lookup_add_all(const uint8_t* data, size_t n, uint16_t* prev_out, uint16_t* head_out,
                       uint16_t* chain_out) {
    lookup_t* l = (lookup_t*)calloc(1, sizeof *l);
    for (size_t i = 0; i < n && i < 65536; i++)
        prev_out[i] = lookup_add(l, data + i, n - i, (uint16_t)i);
    if (head_out) memcpy(head_out, l->head, sizeof l->head);
    if (chain_out) memcpy(chain_out, l->chain, sizeof l->chain);
    free(l);
}
void fo_lookup_bulk_add(const uint8_t* data, size_t n, uint16_t* head_out, uint16_t* chain_out) {
    lookup_t* l = (lookup_t*)calloc(1, sizeof *l);
    lookup_bulk_add(l, data, n, (uint16_t)n, 0);
    memcpy(head_out, l->head, sizeof l->head);
    memcpy(chain_out, l->chain, sizeof l->chain);
    free(l);
}

/* --------------------------------------------------------- sliding window */
/* SlidingWindow.zig */
typedef struct {
    uint8_t buffer[WIN_LEN];
    size_t wp, rp;
    long fp;
} window_t;

static uint16_t win_slide(window_t* w) { /* SlidingWindow.zig:36-44 */
    if (!(w->rp >= MAX_RP && w->wp >= w->rp)) abort();
    size_t n = w->wp - HIST_LEN;
    memmove(w->buffer, w->buffer + HIST_LEN, n);
    w->rp -= HIST_LEN;
    w->wp -= HIST_LEN;
    w->fp -= (long)HIST_LEN;
    return (uint16_t)n;
}
/* SlidingWindow.zig:81-104 */
static uint16_t win_match_raw(const uint8_t* buffer, size_t wp, uint16_t prev_pos,
                              uint16_t curr_pos, uint16_t min_len) {
    size_t max_len = wp - curr_pos;
    if (max_len > MAX_MATCH) max_len = MAX_MATCH;
    const uint8_t* prev_lh = buffer + prev_pos;
    const uint8_t* curr_lh = buffer + curr_pos;
    size_t i = min_len;
    if (i > 0) {
        if (max_len <= i) return 0;
        for (;;) {
            if (prev_lh[i] != curr_lh[i]) return 0;
            if (i == 0) break;
            i--;
        }
        i = min_len;
    }
    while (i < max_len) {
        if (prev_lh[i] != curr_lh[i]) break;
        i++;
    }
    return i >= MIN_MATCH ? (uint16_t)i : 0;
}
uint16_t fo_window_match(const uint8_t* data, size_t wp, uint16_t prev_pos, uint16_t curr_pos,
                         uint16_t min_len) {
    return win_match_raw(data, wp, prev_pos, curr_pos, min_len);
}

/* -------------------------------------------------------------- tokenizer */
/* deflate.zig:35-53 */
typedef struct { uint16_t good, lazy, nice, chain; } level_args_t;
static level_args_t level_args(int level) {
    switch (level) {
        case 4: return (level_args_t){4, 4, 16, 16};
        case 5: return (level_args_t){8, 16, 32, 32};
        case 6: return (level_args_t){8, 16, 128, 128};
        case 7: return (level_args_t){8, 32, 128, 256};
        case 8: return (level_args_t){32, 128, 258, 1024};
        case 9: return (level_args_t){32, 258, 258, 4096};
        default: abort();
    }
}

enum { FLUSH_NONE = 0, FLUSH_FLUSH = 1, FLUSH_FINAL = 2 };

struct fo_deflate {
    int container, mode;
    fo_sink out;
    hasher_t hasher;
    blockw_t bw;
    /* levels 4..9: deflate.zig:122-134 */
    lookup_t lookup;
    window_t win;
    uint32_t tokens[TOKENS_MAX];
    size_t tokens_pos;
    level_args_t level;
    int has_prev_match, has_prev_literal;
    uint32_t prev_match;
    uint8_t prev_literal;
    size_t q1_pending; /* bytes of the token being added that the window has not advanced over yet (quirk Q1, below) */
    /* what every flushTokens handed to the block writer, in stream offsets (tests: the window-slide path against the
     * independent model of tests/golden/make_slide_fixtures.py): tokens, final, has input, slice start, slice length */
    uint64_t slid_total;
    uint64_t* blk_log;
    size_t blk_log_len, blk_log_cap;
    /* simple compressors: deflate.zig:456-457 */
    uint8_t sbuf[65535];
    size_t swp;
};

fo_deflate* fo_deflate_new(int container, int mode) {
    fo_deflate* d = (fo_deflate*)calloc(1, sizeof *d);
    if (!d) abort();
    d->container = container;
    d->mode = mode;
    hasher_init(&d->hasher, container);
    blockw_init(&d->bw, &d->out);
    if (mode >= 4) d->level = level_args(mode);
    write_container_header(container, &d->out); /* deflate.zig:144, 470 */
    return d;
}
void fo_deflate_free(fo_deflate* d) {
    if (!d) return;
    fo_sink_free(&d->out);
    free(d->bw.tok_log);
    free(d->blk_log);
    free(d);
}
const uint64_t* fo_deflate_block_log(const fo_deflate* d, size_t* count) {
    *count = d->blk_log_len / 5;
    return d->blk_log;
}
const uint8_t* fo_deflate_output(const fo_deflate* d, size_t* len) {
    *len = d->out.len;
    return d->out.data;
}
void fo_deflate_log_tokens(fo_deflate* d, int enable) { d->bw.log_tokens = enable; }
const uint32_t* fo_deflate_token_log(const fo_deflate* d, size_t* count) {
    *count = d->bw.tok_log_len;
    return d->bw.tok_log;
}

/* NOT the reference's behaviour, off by default (tests of FLATE_HIP_DEFLATE_REPAIR_Q1 only): the reference with its window
 * advanced (deflate.zig:193) BEFORE a full token block is flushed (deflate.zig:227-230), so that the slice handed to the block
 * writer is the bytes the tokens cover.  As written the slice ends before the bytes of a match that is the block's 32768th
 * token (quirk Q1). */
static int fo_q1_repair = 0;
void fo_set_q1_repair(int on) { fo_q1_repair = on; }

/* deflate.zig:268-288 */
static void df_flush_tokens(fo_deflate* d, int flush_opt) {
    /* win.tokensBuffer(): SlidingWindow.zig:119-123 */
    const size_t rp_eff = d->win.rp + (fo_q1_repair ? d->q1_pending : 0);
    int has_input = d->win.fp >= 0;
    const uint8_t* input = has_input ? d->win.buffer + d->win.fp : NULL;
    size_t input_len = has_input ? rp_eff - (size_t)d->win.fp : 0;
    if (d->bw.log_tokens) {
        if (d->blk_log_len + 5 > d->blk_log_cap) {
            d->blk_log_cap = d->blk_log_cap ? d->blk_log_cap * 2 : 320;
            d->blk_log = (uint64_t*)realloc(d->blk_log, d->blk_log_cap * sizeof(uint64_t));
            if (!d->blk_log) abort();
        }
        uint64_t* e = d->blk_log + d->blk_log_len;
        e[0] = d->tokens_pos;
        e[1] = flush_opt == FLUSH_FINAL;
        e[2] = (uint64_t)has_input;
        e[3] = has_input ? d->slid_total + (uint64_t)d->win.fp : 0;
        e[4] = input_len;
        d->blk_log_len += 5;
    }
    blockw_write(&d->bw, d->tokens, d->tokens_pos, flush_opt == FLUSH_FINAL, input, input_len,
                 has_input);
    if (flush_opt == FLUSH_FLUSH) blockw_stored_block(&d->bw, NULL, 0, 0);
    if (flush_opt != FLUSH_NONE) bw_flush(&d->bw.bw);
    d->tokens_pos = 0;
    d->win.fp = (long)rp_eff; /* SlidingWindow.zig:113-115 (rp_eff == rp unless the test-only repair is on) */
}
/* deflate.zig:227-230 */
static void df_add_token(fo_deflate* d, uint32_t t) {
    d->tokens[d->tokens_pos++] = t;
    if (d->tokens_pos == TOKENS_MAX) df_flush_tokens(d, FLUSH_NONE);
}
static void df_add_prev_literal(fo_deflate* d) { /* deflate.zig:214-216 */
    if (d->has_prev_literal) df_add_token(d, FO_TOK_LIT(d->prev_literal));
}
static uint16_t df_add_match(fo_deflate* d, uint32_t m, size_t not_advanced) { /* deflate.zig:220-225 */
    d->q1_pending = not_advanced;
    df_add_token(d, m);
    d->q1_pending = 0;
    d->has_prev_literal = 0;
    d->has_prev_match = 0;
    return (uint16_t)(FO_TOK_LENLIT(m) + 3);
}
#ifdef FO_STATS /* tools/lz_stats.c: counters of the reference walk, never built into the oracle library */
unsigned long long fo_stat_calls[3], fo_stat_cands[3], fo_stat_hist[3][14], fo_stat_lenhist[260], fo_stat_cmp8;
#define FO_STAT(x) x
#else
#define FO_STAT(x)
#endif
/* deflate.zig:233-266 */
static int df_find_match(fo_deflate* d, uint16_t pos, const uint8_t* lh, size_t lh_len,
                         uint16_t min_len, uint32_t* match_out) {
    uint16_t len = min_len;
    uint16_t prev_pos = lookup_add(&d->lookup, lh, lh_len, pos);
    int found = 0;
    size_t chain = d->level.chain;
    if (len >= d->level.good) chain >>= 2;
    FO_STAT(int kind = min_len == 0 ? 0 : (min_len >= d->level.good ? 2 : 1); unsigned nc = 0; fo_stat_calls[kind]++;)
    while (prev_pos > 0 && chain > 0) {
        uint16_t distance = (uint16_t)(pos - prev_pos);
        if (distance > 32768) break;
        FO_STAT(nc++;)
        uint16_t new_len = win_match_raw(d->win.buffer, d->win.wp, prev_pos, pos, len);
        FO_STAT(if (new_len >= 8) fo_stat_cmp8++;)
        if (new_len > len) {
            *match_out = FO_TOK_MATCH(distance, new_len);
            found = 1;
            if (new_len >= d->level.nice) {
                FO_STAT(fo_stat_cands[kind] += nc; { int b = 0; while ((1u << b) <= nc && b < 13) b++; fo_stat_hist[kind][b]++; })
                return 1;
            }
            len = new_len;
        }
        prev_pos = d->lookup.chain[prev_pos];
        chain--;
    }
    FO_STAT(fo_stat_cands[kind] += nc; { int b = 0; while ((1u << b) <= nc && b < 13) b++; fo_stat_hist[kind][b]++; })
    return found;
}
/* deflate.zig:154-205 */
static void df_tokenize(fo_deflate* d, int flush_opt) {
    int should_flush = flush_opt != FLUSH_NONE;
    for (;;) {
        /* activeLookahead: SlidingWindow.zig:56-60 */
        size_t lh_len = d->win.wp - d->win.rp;
        size_t min = should_flush ? 0 : MIN_LOOKAHEAD;
        if (!(lh_len > min)) break;
        const uint8_t* lh = d->win.buffer + d->win.rp;

        uint16_t step = 1;
        uint16_t pos = (uint16_t)d->win.rp;
        uint8_t literal = lh[0];
        uint16_t min_len = d->has_prev_match ? (uint16_t)(FO_TOK_LENLIT(d->prev_match) + 3) : 0;
        uint32_t match;
        if (df_find_match(d, pos, lh, lh_len, min_len, &match)) {
            df_add_prev_literal(d);
            if (FO_TOK_LENLIT(match) + 3 >= d->level.lazy) {
                step = df_add_match(d, match, FO_TOK_LENLIT(match) + 3u);
            } else {
                d->prev_literal = literal;
                d->has_prev_literal = 1;
                d->prev_match = match;
                d->has_prev_match = 1;
            }
        } else {
            if (d->has_prev_match) {
                step = (uint16_t)(df_add_match(d, d->prev_match, FO_TOK_LENLIT(d->prev_match) + 2u) - 1);
            } else {
                df_add_prev_literal(d);
                d->prev_literal = literal;
                d->has_prev_literal = 1;
            }
        }
        /* windowAdvance: deflate.zig:207-211 */
        lookup_bulk_add(&d->lookup, lh + 1, lh_len - 1, (uint16_t)(step - 1), (uint16_t)(pos + 1));
        d->win.rp += step;
    }
    if (should_flush) {
        if (d->has_prev_match) abort(); /* deflate.zig:199 */
        df_add_prev_literal(d);
        d->has_prev_literal = 0;
        df_flush_tokens(d, flush_opt);
    }
}

/* deflate.zig:304-321 driven by a fixed buffer reader (deflate.zig:363-367) */
static void df_compress(fo_deflate* d, const uint8_t* in, size_t n) {
    size_t off = 0;
    for (;;) {
        size_t room = WIN_LEN - d->win.wp;
        if (room == 0) {
            df_tokenize(d, FLUSH_NONE);
            uint16_t k = win_slide(&d->win); /* deflate.zig:291-294 */
            d->slid_total += HIST_LEN;
            lookup_slide(&d->lookup, k);
            continue;
        }
        size_t k = n - off < room ? n - off : room;
        memcpy(d->win.buffer + d->win.wp, in + off, k);
        hasher_update(&d->hasher, d->win.buffer + d->win.wp, k);
        d->win.wp += k;
        off += k;
        df_tokenize(d, FLUSH_NONE);
        if (k < room) break;
    }
}

/* deflate.zig:486-493 */
static void simple_flush_buffer(fo_deflate* d, int final) {
    if (d->mode == FO_MODE_HUFFMAN)
        blockw_huffman_block(&d->bw, d->sbuf, d->swp, final);
    else
        blockw_stored_block(&d->bw, d->sbuf, d->swp, final);
    d->swp = 0;
}
/* deflate.zig:498-511 */
static void simple_compress(fo_deflate* d, const uint8_t* in, size_t n) {
    size_t off = 0;
    for (;;) {
        size_t room = sizeof d->sbuf - d->swp;
        if (room == 0) {
            simple_flush_buffer(d, 0);
            continue;
        }
        size_t k = n - off < room ? n - off : room;
        memcpy(d->sbuf + d->swp, in + off, k);
        hasher_update(&d->hasher, d->sbuf + d->swp, k);
        d->swp += k;
        off += k;
        if (k < room) break;
    }
}

void fo_deflate_write(fo_deflate* d, const uint8_t* in, size_t n) {
    if (d->mode >= 4)
        df_compress(d, in, n);
    else
        simple_compress(d, in, n);
}
void fo_deflate_flush(fo_deflate* d) {
    if (d->mode >= 4) {
        df_tokenize(d, FLUSH_FLUSH); /* deflate.zig:335-337 */
    } else { /* deflate.zig:474-478 */
        simple_flush_buffer(d, 0);
        blockw_stored_block(&d->bw, NULL, 0, 0);
        bw_flush(&d->bw.bw);
    }
}
void fo_deflate_finish(fo_deflate* d) {
    if (d->mode >= 4) {
        df_tokenize(d, FLUSH_FINAL); /* deflate.zig:344-347 */
    } else { /* deflate.zig:480-484 */
        simple_flush_buffer(d, 1);
        bw_flush(&d->bw.bw);
    }
    write_container_footer(&d->hasher, &d->out);
}

size_t fo_compress_bound(size_t n) {
    /* stored blocks: 5 bytes per 65535 + one possibly-empty trailing block,
     * Huffman blocks never exceed stored by more than the header; add slack
     * for the Q1 corner (duplicated bytes around a stored/Huffman seam). */
    return n + (n / 32768 + 2) * 16 + 18 + 64 + n / 64;
}

int fo_compress(const uint8_t* in, size_t n, int container, int mode, uint8_t* out, size_t cap,
                size_t* out_len) {
    fo_deflate* d = fo_deflate_new(container, mode);
    fo_deflate_write(d, in, n);
    fo_deflate_finish(d);
    int rc = 0;
    *out_len = d->out.len;
    if (d->out.len > cap)
        rc = FO_OUTPUT_TOO_SMALL;
    else
        memcpy(out, d->out.data, d->out.len);
    fo_deflate_free(d);
    return rc;
}

int fo_tokenize(const uint8_t* in, size_t n, int level, uint32_t* tokens, size_t cap,
                size_t* count) {
    fo_deflate* d = fo_deflate_new(FO_RAW, level);
    fo_deflate_log_tokens(d, 1);
    fo_deflate_write(d, in, n);
    fo_deflate_flush(d);
    int rc = 0;
    *count = d->bw.tok_log_len;
    if (d->bw.tok_log_len > cap)
        rc = FO_OUTPUT_TOO_SMALL;
    else if (d->bw.tok_log_len)
        memcpy(tokens, d->bw.tok_log, d->bw.tok_log_len * sizeof(uint32_t));
    fo_deflate_free(d);
    return rc;
}

/* ---------------------------------------------------------------- inflate */
/* bit_reader.zig modelled by bit position: `fill(nice)` fails with
 * EndOfStream only when no bit is left at all (bit_reader.zig:59-67); `shift`
 * fails when more bits are consumed than remain (:159-163); peeks beyond the
 * end see zero bits. */
typedef struct {
    const uint8_t* data;
    uint64_t total_bits;
    uint64_t pos;
} bitr_t;

static inline int br_fill(bitr_t* r, unsigned nice) {
    if (nice > 0 && r->pos >= r->total_bits) return FO_END_OF_STREAM;
    return FO_OK;
}
static inline uint64_t br_peek(const bitr_t* r, unsigned n) { /* n <= 32 */
    uint64_t v = 0;
    uint64_t byte = r->pos >> 3;
    unsigned sh = (unsigned)(r->pos & 7);
    uint64_t nbytes = r->total_bits >> 3;
    for (unsigned i = 0; i < 6; i++) {
        uint64_t b = byte + i < nbytes ? r->data[byte + i] : 0;
        v |= b << (8 * i);
    }
    v >>= sh;
    return n >= 64 ? v : (v & ((1ull << n) - 1));
}
static inline int br_shift(bitr_t* r, unsigned n) {
    if (n > r->total_bits - r->pos) return FO_END_OF_STREAM;
    r->pos += n;
    return FO_OK;
}
/* readF(U, 0): fill(n), take, shift(n) -- bit_reader.zig:104-109 */
static inline int br_read(bitr_t* r, unsigned n, uint32_t* v) {
    int rc = br_fill(r, n);
    if (rc) return rc;
    *v = (uint32_t)br_peek(r, n);
    return br_shift(r, n);
}
static inline void br_align(bitr_t* r) { r->pos = (r->pos + 7) & ~7ull; } /* :189-192 */
static inline uint32_t rev_bits(uint32_t v, unsigned n) {
    uint32_t r = 0;
    for (unsigned i = 0; i < n; i++)
        if (v & (1u << i)) r |= 1u << (n - 1 - i);
    return r;
}

/* huffman_decoder.zig: canonical code over symbols ordered by
 * (code_bits, kind, symbol) == (code_bits, symbol index) (:20-28, 84). */
typedef struct {
    uint16_t count[16];
    uint16_t symbol[286];
    int max_code_bits;
} hdec_t;

/* checkCompletnes, huffman_decoder.zig:126-153 */
static int hdec_generate(hdec_t* d, const uint8_t* lens, int n, int alphabet, int max_code_bits) {
    if (alphabet == 286 && lens[256] == 0) return FO_MISSING_END_OF_BLOCK_CODE;
    memset(d->count, 0, sizeof d->count);
    d->max_code_bits = max_code_bits;
    int max = 0;
    for (int i = 0; i < n; i++) {
        if (lens[i] == 0) continue;
        if (lens[i] > max) max = lens[i];
        d->count[lens[i]]++;
    }
    if (max != 0) {
        long left = 1;
        for (int len = 1; len <= max_code_bits; len++) {
            left <<= 1;
            if (d->count[len] > left) return FO_OVERSUBSCRIBED_HUFFMAN_TREE;
            left -= d->count[len];
        }
        if (left > 0) {
            if (!(max_code_bits > 7 && max == d->count[1])) return FO_INCOMPLETE_HUFFMAN_TREE;
        }
    }
    uint16_t offs[17];
    offs[1] = 0;
    for (int len = 1; len < 16; len++) offs[len + 1] = (uint16_t)(offs[len] + d->count[len]);
    for (int i = 0; i < n; i++)
        if (lens[i] != 0) d->symbol[offs[lens[i]]++] = (uint16_t)i;
    return FO_OK;
}
/* find(), huffman_decoder.zig:156-175: `peek` holds max_code_bits stream bits
 * (LSB-first as read); returns the symbol whose code is a prefix. */
static int hdec_find(const hdec_t* d, uint32_t peek, uint16_t* sym, unsigned* code_bits) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= d->max_code_bits; len++) {
        code |= (int)(peek & 1);
        peek >>= 1;
        int count = d->count[len];
        if (code - count < first) {
            *sym = d->symbol[index + (code - first)];
            *code_bits = (unsigned)len;
            return FO_OK;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return FO_INVALID_CODE;
}

typedef struct {
    bitr_t bits;
    uint8_t* out;
    size_t cap, wp;
    hdec_t lit_dec, dst_dec;
    int flags;
} inflate_t;

static const uint16_t len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t dist_base[30] = {1,    2,    3,    4,    5,    7,    9,    13,    17,    25,
                                       33,   49,   65,   97,   129,  193,  257,  385,   513,   769,
                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};

/* CircularBuffer.zig:44-75 */
static int inf_write_match(inflate_t* s, uint32_t length, uint32_t distance) {
    if (s->wp < distance || length < 3 || length > 258 || distance < 1 || distance > 32768)
        return FO_INVALID_MATCH;
    if (s->wp + length > s->cap) return FO_OUTPUT_TOO_SMALL;
    for (uint32_t i = 0; i < length; i++) {
        s->out[s->wp] = s->out[s->wp - distance];
        s->wp++;
    }
    return FO_OK;
}
static inline int inf_write_lit(inflate_t* s, uint8_t b) {
    if (s->wp >= s->cap) return FO_OUTPUT_TOO_SMALL;
    s->out[s->wp++] = b;
    return FO_OK;
}

/* inflate.zig:123-140: callers have already fill()ed */
static int inf_decode_length(inflate_t* s, uint32_t code, uint32_t* length) {
    if (code > 28) return FO_INVALID_CODE;
    unsigned eb = len_extra_bits[code];
    *length = len_base[code];
    if (eb) {
        *length += (uint32_t)br_peek(&s->bits, eb);
        return br_shift(&s->bits, eb);
    }
    return FO_OK;
}
static int inf_decode_distance(inflate_t* s, uint32_t code, uint32_t* distance) {
    if (code > 29) return FO_INVALID_CODE;
    unsigned eb = dist_extra_bits[code];
    *distance = dist_base[code];
    if (eb) {
        *distance += (uint32_t)br_peek(&s->bits, eb);
        return br_shift(&s->bits, eb);
    }
    return FO_OK;
}

/* inflate.zig:89-102 */
static int inf_stored_block(inflate_t* s) {
    br_align(&s->bits);
    uint32_t len, nlen;
    int rc;
    if ((rc = br_read(&s->bits, 16, &len))) return rc;
    if ((rc = br_read(&s->bits, 16, &nlen))) return rc;
    if (len != ((~nlen) & 0xffff)) return FO_WRONG_STORED_BLOCK_NLEN;
    if ((uint64_t)len * 8 > s->bits.total_bits - s->bits.pos) return FO_END_OF_STREAM;
    if (s->wp + len > s->cap) return FO_OUTPUT_TOO_SMALL;
    memcpy(s->out + s->wp, s->bits.data + (s->bits.pos >> 3), len);
    s->wp += len;
    s->bits.pos += (uint64_t)len * 8;
    return FO_OK;
}

/* bit_reader.zig:205-217 */
static int inf_read_fixed_code(inflate_t* s, uint32_t* code) {
    int rc = br_fill(&s->bits, 9);
    if (rc) return rc;
    uint32_t code7 = rev_bits((uint32_t)br_peek(&s->bits, 7), 7);
    if ((rc = br_shift(&s->bits, 7))) return rc;
    uint32_t extra;
    if (code7 <= 0x17) {
        *code = code7 + 256;
    } else if (code7 <= 0x5f) {
        extra = (uint32_t)br_peek(&s->bits, 1);
        if ((rc = br_shift(&s->bits, 1))) return rc;
        *code = (code7 << 1) + extra - 0x30;
    } else if (code7 <= 0x63) {
        extra = (uint32_t)br_peek(&s->bits, 1);
        if ((rc = br_shift(&s->bits, 1))) return rc;
        *code = ((code7 - 0x60) << 1) + extra + 280;
    } else {
        extra = rev_bits((uint32_t)br_peek(&s->bits, 2), 2);
        if ((rc = br_shift(&s->bits, 2))) return rc;
        *code = ((code7 - 0x64) << 2) + extra + 144;
    }
    return FO_OK;
}

/* inflate.zig:104-121 */
static int inf_fixed_block(inflate_t* s) {
    for (;;) {
        uint32_t code;
        int rc = inf_read_fixed_code(s, &code);
        if (rc) return rc;
        if (code <= 255) {
            if ((rc = inf_write_lit(s, (uint8_t)code))) return rc;
        } else if (code == 256) {
            return FO_OK;
        } else if (code <= 285) {
            if ((rc = br_fill(&s->bits, 5 + 5 + 13))) return rc;
            uint32_t length, distance;
            if ((rc = inf_decode_length(s, code - 257, &length))) return rc;
            uint32_t dcode = rev_bits((uint32_t)br_peek(&s->bits, 5), 5);
            if ((rc = br_shift(&s->bits, 5))) return rc;
            if ((rc = inf_decode_distance(s, dcode, &distance))) return rc;
            if ((rc = inf_write_match(s, length, distance))) return rc;
        } else {
            return FO_INVALID_CODE;
        }
    }
}

/* inflate.zig:188-216 */
static int inf_dynamic_code_length(inflate_t* s, uint16_t code, uint8_t* lens, size_t lens_len,
                                   size_t pos, size_t* adv) {
    if (pos >= lens_len) return FO_INVALID_DYNAMIC_BLOCK_HEADER;
    uint32_t v;
    int rc;
    switch (code) {
        case 16:
            if ((rc = br_read(&s->bits, 2, &v))) return rc;
            v += 3;
            if (pos == 0 || pos + v > lens_len) return FO_INVALID_DYNAMIC_BLOCK_HEADER;
            for (uint32_t i = 0; i < v; i++) lens[pos + i] = lens[pos + i - 1];
            *adv = v;
            return FO_OK;
        case 17:
            if ((rc = br_read(&s->bits, 3, &v))) return rc;
            *adv = v + 3;
            return FO_OK;
        case 18:
            if ((rc = br_read(&s->bits, 7, &v))) return rc;
            *adv = v + 11;
            return FO_OK;
        default:
            if (code > 15) return FO_INVALID_DYNAMIC_BLOCK_HEADER;
            lens[pos] = (uint8_t)code;
            *adv = 1;
            return FO_OK;
    }
}

static int inf_read_lens(inflate_t* s, const hdec_t* cl_dec, uint8_t* lens, size_t lens_len,
                         size_t want, size_t boundary, int* crossed) {
    size_t pos = 0;
    while (pos < want) {
        int rc = br_fill(&s->bits, 7);
        if (rc) return rc;
        uint16_t sym;
        unsigned cb;
        if ((rc = hdec_find(cl_dec, (uint32_t)br_peek(&s->bits, 7), &sym, &cb))) return rc;
        if ((rc = br_shift(&s->bits, cb))) return rc;
        size_t adv;
        if (boundary && sym == 16 && pos == boundary) *crossed = 1;
        if ((rc = inf_dynamic_code_length(s, sym, lens, lens_len, pos, &adv))) return rc;
        if (boundary && pos < boundary && pos + adv > boundary) *crossed = 1;
        pos += adv;
    }
    if (pos > want) return FO_INVALID_DYNAMIC_BLOCK_HEADER;
    return FO_OK;
}

/* inflate.zig:144-184 */
static int inf_dynamic_block_header(inflate_t* s) {
    uint32_t v;
    int rc;
    if ((rc = br_read(&s->bits, 5, &v))) return rc;
    uint32_t hlit = v + 257;
    if ((rc = br_read(&s->bits, 5, &v))) return rc;
    uint32_t hdist = v + 1;
    if ((rc = br_read(&s->bits, 4, &v))) return rc;
    uint32_t hclen = v + 4;
    if (hlit > 286 || hdist > 30) return FO_INVALID_DYNAMIC_BLOCK_HEADER;

    uint8_t cl_lens[19];
    memset(cl_lens, 0, sizeof cl_lens);
    for (uint32_t i = 0; i < hclen; i++) {
        if ((rc = br_read(&s->bits, 3, &v))) return rc;
        cl_lens[codegen_order[i]] = (uint8_t)v;
    }
    hdec_t cl_dec;
    if ((rc = hdec_generate(&cl_dec, cl_lens, 19, 19, 7))) return rc;

    uint8_t lit_lens[286], dst_lens[30];
    memset(lit_lens, 0, sizeof lit_lens);
    memset(dst_lens, 0, sizeof dst_lens);
    int crossed = 0;
    if (s->flags & 1) {
        /* reference-exact (Q6): two separate lists, inflate.zig:161-180 */
        if ((rc = inf_read_lens(s, &cl_dec, lit_lens, 286, hlit, 0, &crossed))) return rc;
        if ((rc = inf_read_lens(s, &cl_dec, dst_lens, 30, hdist, 0, &crossed))) return rc;
    } else {
        /* RFC 1951 3.2.7: one list of hlit + hdist lengths (as puff.c:703-724).
         * A repeat that crosses the HLIT/HDIST boundary is where the reference
         * stops with InvalidDynamicBlockHeader; we go on, accept the header if
         * it is valid, and otherwise report the reference's error name. */
        uint8_t lens[286 + 30];
        memset(lens, 0, sizeof lens);
        rc = inf_read_lens(s, &cl_dec, lens, hlit + hdist, hlit + hdist, hlit, &crossed);
        if (rc) return crossed ? FO_INVALID_DYNAMIC_BLOCK_HEADER : rc;
        memcpy(lit_lens, lens, hlit);
        memcpy(dst_lens, lens + hlit, hdist);
    }
    if ((rc = hdec_generate(&s->lit_dec, lit_lens, 286, 286, 15)))
        return crossed ? FO_INVALID_DYNAMIC_BLOCK_HEADER : rc;
    if ((rc = hdec_generate(&s->dst_dec, dst_lens, 30, 30, 15)))
        return crossed ? FO_INVALID_DYNAMIC_BLOCK_HEADER : rc;
    return FO_OK;
}

/* inflate.zig:220-249 */
static int inf_dynamic_block(inflate_t* s) {
    for (;;) {
        int rc = br_fill(&s->bits, 15);
        if (rc) return rc;
        uint16_t sym;
        unsigned cb;
        if ((rc = hdec_find(&s->lit_dec, (uint32_t)br_peek(&s->bits, 15), &sym, &cb))) return rc;
        if ((rc = br_shift(&s->bits, cb))) return rc;
        if (sym < 256) {
            if ((rc = inf_write_lit(s, (uint8_t)sym))) return rc;
        } else if (sym == 256) {
            return FO_OK;
        } else {
            if ((rc = br_fill(&s->bits, 5 + 15 + 13))) return rc;
            uint32_t length, distance;
            if ((rc = inf_decode_length(s, sym - 257u, &length))) return rc;
            uint16_t dsym;
            if ((rc = hdec_find(&s->dst_dec, (uint32_t)br_peek(&s->bits, 15), &dsym, &cb))) return rc;
            if ((rc = br_shift(&s->bits, cb))) return rc;
            if ((rc = inf_decode_distance(s, dsym, &distance))) return rc;
            if ((rc = inf_write_match(s, length, distance))) return rc;
        }
    }
}

/* container.zig:119-152 */
static int inf_parse_header(inflate_t* s, int container) {
    uint32_t v;
    int rc;
    if (container == FO_GZIP) {
        uint32_t magic1, magic2, method, flags;
        if ((rc = br_read(&s->bits, 8, &magic1))) return rc;
        if ((rc = br_read(&s->bits, 8, &magic2))) return rc;
        if ((rc = br_read(&s->bits, 8, &method))) return rc;
        if ((rc = br_read(&s->bits, 8, &flags))) return rc;
        for (int i = 0; i < 6; i++)
            if ((rc = br_read(&s->bits, 8, &v))) return rc;
        if (magic1 != 0x1f || magic2 != 0x8b || method != 0x08) return FO_BAD_GZIP_HEADER;
        if (flags != 0) {
            if (flags & 0x04) {
                uint32_t lo, hi;
                if ((rc = br_fill(&s->bits, 16))) return rc;
                lo = (uint32_t)br_peek(&s->bits, 16);
                if ((rc = br_shift(&s->bits, 16))) return rc;
                hi = lo;
                for (uint32_t i = 0; i < hi; i++)
                    if ((rc = br_read(&s->bits, 8, &v))) return rc;
            }
            if (flags & 0x08) {
                do {
                    if ((rc = br_read(&s->bits, 8, &v))) return rc;
                } while (v != 0);
            }
            if (flags & 0x10) {
                do {
                    if ((rc = br_read(&s->bits, 8, &v))) return rc;
                } while (v != 0);
            }
            if (flags & 0x02) {
                for (int i = 0; i < 2; i++)
                    if ((rc = br_read(&s->bits, 8, &v))) return rc;
            }
        }
    } else if (container == FO_ZLIB) {
        uint32_t cm, cinfo;
        if ((rc = br_read(&s->bits, 4, &cm))) return rc;
        if ((rc = br_read(&s->bits, 4, &cinfo))) return rc;
        if ((rc = br_read(&s->bits, 8, &v))) return rc;
        if (cm != 8 || cinfo > 7) return FO_BAD_ZLIB_HEADER;
    }
    return FO_OK;
}

/* container.zig:154-166 */
static int inf_parse_footer(inflate_t* s, int container) {
    int rc;
    uint32_t v;
    if (container == FO_GZIP) {
        uint32_t crc = fo_crc32(0, s->out, s->wp);
        if ((rc = br_fill(&s->bits, 32))) return rc;
        v = (uint32_t)br_peek(&s->bits, 32);
        if ((rc = br_shift(&s->bits, 32))) return rc;
        if (v != crc) return FO_WRONG_GZIP_CHECKSUM;
        if ((rc = br_fill(&s->bits, 32))) return rc;
        v = (uint32_t)br_peek(&s->bits, 32);
        if ((rc = br_shift(&s->bits, 32))) return rc;
        if (v != (uint32_t)s->wp) return FO_WRONG_GZIP_SIZE;
    } else if (container == FO_ZLIB) {
        uint32_t ad = fo_adler32(1, s->out, s->wp);
        uint32_t swapped = (ad >> 24) | ((ad >> 8) & 0xff00) | ((ad << 8) & 0xff0000) | (ad << 24);
        if ((rc = br_fill(&s->bits, 32))) return rc;
        v = (uint32_t)br_peek(&s->bits, 32);
        if ((rc = br_shift(&s->bits, 32))) return rc;
        if (v != swapped) return FO_WRONG_ZLIB_CHECKSUM;
    }
    return FO_OK;
}

/* inflate.zig:251-280 */
int fo_decompress(const uint8_t* in, size_t n, int container, int flags, uint8_t* out, size_t cap,
                  size_t* out_len, size_t* consumed) {
    TOK_INIT();
    inflate_t* s = (inflate_t*)calloc(1, sizeof *s);
    s->bits.data = in;
    s->bits.total_bits = (uint64_t)n * 8;
    s->out = out;
    s->cap = cap;
    s->flags = flags;
    int rc = inf_parse_header(s, container);
    while (rc == FO_OK) {
        uint32_t bfinal, btype;
        if ((rc = br_read(&s->bits, 1, &bfinal))) break;
        if ((rc = br_read(&s->bits, 2, &btype))) break;
        if (btype == 2) {
            if ((rc = inf_dynamic_block_header(s))) break;
            rc = inf_dynamic_block(s);
        } else if (btype == 0) {
            rc = inf_stored_block(s);
        } else if (btype == 1) {
            rc = inf_fixed_block(s);
        } else {
            rc = FO_INVALID_BLOCK_TYPE;
        }
        if (rc) break;
        if (bfinal) {
            br_align(&s->bits);
            rc = inf_parse_footer(s, container);
            break;
        }
    }
    *out_len = s->wp;
    if (consumed) *consumed = (size_t)((s->bits.pos + 7) >> 3);
    free(s);
    return rc;
}

const char* fo_status_name(int status) {
    switch (status) {
        case FO_OK: return "Ok";
        case FO_END_OF_STREAM: return "EndOfStream";
        case FO_BAD_GZIP_HEADER: return "BadGzipHeader";
        case FO_BAD_ZLIB_HEADER: return "BadZlibHeader";
        case FO_WRONG_GZIP_CHECKSUM: return "WrongGzipChecksum";
        case FO_WRONG_GZIP_SIZE: return "WrongGzipSize";
        case FO_WRONG_ZLIB_CHECKSUM: return "WrongZlibChecksum";
        case FO_INVALID_CODE: return "InvalidCode";
        case FO_OVERSUBSCRIBED_HUFFMAN_TREE: return "OversubscribedHuffmanTree";
        case FO_INCOMPLETE_HUFFMAN_TREE: return "IncompleteHuffmanTree";
        case FO_MISSING_END_OF_BLOCK_CODE: return "MissingEndOfBlockCode";
        case FO_INVALID_MATCH: return "InvalidMatch";
        case FO_INVALID_BLOCK_TYPE: return "InvalidBlockType";
        case FO_WRONG_STORED_BLOCK_NLEN: return "WrongStoredBlockNlen";
        case FO_INVALID_DYNAMIC_BLOCK_HEADER: return "InvalidDynamicBlockHeader";
        case FO_OUTPUT_TOO_SMALL: return "OutputTooSmall";
        default: return "Unknown";
    }
}
