/*
 * flate_oracle.h -- CPU oracle for the DEFLATE hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a from-scratch plain-C restatement of the algorithm of ianic/flate
 * (reference @ /root/reference, Zig).  It exists only to CHECK the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product library (libflate_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check it against every
 * golden vector / known-answer the reference's own tests hold for this path
 * (token lists, 36 token counts, 32x3 compressed sizes, 85 golden block
 * checks, Huffman known-answers, inflate vectors, the 40-case fuzz error
 * table, header/footer error cases, multi-stream) -- see SURVEY.md section 8c.
 * The Zig reference itself cannot be built here (no zig toolchain); the only
 * compilable reference artefact, bin/puff/puff.c, is built into oracle/_ref/
 * and used as a differential inflater.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef FLATE_ORACLE_H
#define FLATE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* container tags -- container.zig:18-21 */
enum { FO_RAW = 0, FO_GZIP = 1, FO_ZLIB = 2 };

/* modes: 0 store-only, 1 huffman-only, 4..9 levels (deflate.zig:23-32, 401-434) */
enum { FO_MODE_STORE = 0, FO_MODE_HUFFMAN = 1 };

/* inflate status codes: 1:1 with the reference's error names
 * (bit_reader.zig:29, container.zig:45-51, huffman_decoder.zig:35-40,
 * inflate.zig:72-78).  Same numbering as include/flate_hip.h. */
enum {
    FO_OK = 0,
    FO_END_OF_STREAM = 1,
    FO_BAD_GZIP_HEADER = 2,
    FO_BAD_ZLIB_HEADER = 3,
    FO_WRONG_GZIP_CHECKSUM = 4,
    FO_WRONG_GZIP_SIZE = 5,
    FO_WRONG_ZLIB_CHECKSUM = 6,
    FO_INVALID_CODE = 7,
    FO_OVERSUBSCRIBED_HUFFMAN_TREE = 8,
    FO_INCOMPLETE_HUFFMAN_TREE = 9,
    FO_MISSING_END_OF_BLOCK_CODE = 10,
    FO_INVALID_MATCH = 11,
    FO_INVALID_BLOCK_TYPE = 12,
    FO_WRONG_STORED_BLOCK_NLEN = 13,
    FO_INVALID_DYNAMIC_BLOCK_HEADER = 14,
    FO_OUTPUT_TOO_SMALL = 100
};

/* Token encoding shared with the HIP path: bit 23 kind (1 = match),
 * bits 15..22 len_lit (literal byte, or length-3), bits 0..14 dist-1.
 * Mirrors Token.zig:18-22 field meaning (the Zig struct layout itself is not
 * ABI-stable). */
#define FO_TOK_LIT(b) ((uint32_t)(b) << 15)
#define FO_TOK_MATCH(dist, len) ((1u << 23) | ((uint32_t)((len) - 3) << 15) | (uint32_t)((dist) - 1))
#define FO_TOK_IS_MATCH(t) (((t) >> 23) & 1u)
#define FO_TOK_LENLIT(t) (((t) >> 15) & 0xffu)
#define FO_TOK_DIST0(t) ((t) & 0x7fffu)

/* ---- growable output sink (stands in for the Zig `writer: anytype`) ---- */
typedef struct fo_sink {
    uint8_t* data;
    size_t len, cap;
} fo_sink;
void fo_sink_free(fo_sink* s);

/* ---- streaming compressor (deflate.zig:121-373, 449-529) ---- */
typedef struct fo_deflate fo_deflate;
/* mode: 0 store, 1 huffman, 4..9 level.  Writes the container header. */
fo_deflate* fo_deflate_new(int container, int mode);
void fo_deflate_free(fo_deflate* d);
/* Deflate.write / SimpleCompressor.write: feed bytes (any split). */
void fo_deflate_write(fo_deflate* d, const uint8_t* in, size_t n);
/* Deflate.flush (deflate.zig:335) / SimpleCompressor.flush (:474). */
void fo_deflate_flush(fo_deflate* d);
/* Deflate.finish (deflate.zig:344) / SimpleCompressor.finish (:480). */
void fo_deflate_finish(fo_deflate* d);
/* bytes produced so far */
const uint8_t* fo_deflate_output(const fo_deflate* d, size_t* len);
/* test hook (TestTokenWriter / TokenDecoder seam, deflate.zig:578-608,682-719):
 * when enabled before any write, every token handed to the block writer is
 * also appended to an internal log. */
void fo_deflate_log_tokens(fo_deflate* d, int enable);
const uint32_t* fo_deflate_token_log(const fo_deflate* d, size_t* count);
/* with the token log on: per flushTokens five values -- tokens, final, has input, slice start (stream offset), slice length */
const uint64_t* fo_deflate_block_log(const fo_deflate* d, size_t* count);

/* one-shot: compress()+finish().  returns 0, or FO_OUTPUT_TOO_SMALL. */
int fo_compress(const uint8_t* in, size_t n, int container, int mode,
                uint8_t* out, size_t cap, size_t* out_len);
/* safe upper bound of the output size for any mode */
size_t fo_compress_bound(size_t n);
/* tokens the tokenizer produces for `in` at `level` (write + flush) */
int fo_tokenize(const uint8_t* in, size_t n, int level, uint32_t* tokens,
                size_t cap, size_t* count);

/* ---- block writer seam (block_writer.zig) ---- */
/* fn: 0 = write (:307), 1 = dynamicBlock (:395), 2 = huffmanBlock (:524).
 * input == NULL means "no input" (Zig null).  Output is the flushed block. */
int fo_block_write(int fn, const uint32_t* tokens, size_t ntok, int eof,
                   const uint8_t* input, size_t input_len, int has_input,
                   uint8_t* out, size_t cap, size_t* out_len);

/* ---- Huffman code builder (huffman_encoder.zig:62-278) ---- */
/* freq[n] -> codes[n] (bit-reversed, as stored by the reference), lens[n] */
void fo_huffman_generate(const uint16_t* freq, int n, int max_bits,
                         uint16_t* codes, uint16_t* lens);
/* fixed tables (huffman_encoder.zig:298-338) */
void fo_fixed_literal_codes(uint16_t codes[286], uint16_t lens[286]);

/* ---- hash chain + window unit hooks (Lookup.zig, SlidingWindow.zig) ---- */
uint32_t fo_hash4(const uint8_t* b);                          /* Lookup.zig:75-84 */
/* run add() for every position of data, return prev positions (Lookup.zig:23-27) */
void fo_lookup_add_all(const uint8_t* data, size_t n, uint16_t* prev_out,
                       uint16_t* head_out, uint16_t* chain_out);
void fo_lookup_bulk_add(const uint8_t* data, size_t n, uint16_t* head_out,
                        uint16_t* chain_out);                  /* Lookup.zig:55-72 */
uint16_t fo_window_match(const uint8_t* data, size_t wp, uint16_t prev_pos,
                         uint16_t curr_pos, uint16_t min_len); /* SlidingWindow.zig:81-104 */
/* Token.zig:58-81 */
uint16_t fo_length_code(uint8_t len_lit);
uint8_t fo_distance_code(uint16_t dist0);
uint8_t fo_length_extra_bits(uint16_t code);
uint8_t fo_distance_extra_bits(uint8_t code);

/* ---- inflate (inflate.zig) ---- */
/* flags bit0 = reference-strict dynamic header (quirk Q6, inflate.zig:161-180:
 * literal and distance code lengths decoded as two separate lists, so a
 * repeat code that runs across the HLIT/HDIST boundary -- legal per RFC 1951
 * 3.2.7 and emitted by the reference's own encoder -- is rejected with
 * InvalidDynamicBlockHeader).  Default (0): the lengths are one list (RFC,
 * as puff.c:703-724); a header with such a repeat is accepted when it is
 * valid, and reported as InvalidDynamicBlockHeader (the reference's answer)
 * when anything else in that header is wrong.  Both modes give the
 * reference's error name on all 40 inputs of its fuzz table.
 * consumed (optional) = bytes of `in` used by this stream. */
int fo_decompress(const uint8_t* in, size_t n, int container, int flags,
                  uint8_t* out, size_t cap, size_t* out_len, size_t* consumed);

/* ---- checksums (Zig std.hash.Crc32 / Adler32 at container.zig:170-171) ---- */
uint32_t fo_crc32(uint32_t crc, const uint8_t* p, size_t n);   /* start with 0 */
uint32_t fo_adler32(uint32_t adler, const uint8_t* p, size_t n); /* start with 1 */

const char* fo_status_name(int status);

#ifdef __cplusplus
}
#endif
#endif
