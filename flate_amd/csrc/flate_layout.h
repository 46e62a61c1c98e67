// flate_layout.h -- plain descriptors shared by the host code and the kernels (no HIP types):
// chunk / tile / piece / segment tables, call-wide parameters, CRC constants.
#pragma once
#include <stdint.h>

#include "flate_common.h"

#define FL_SEG 32768u  // whole-stream passes: positions per parse / emit segment and per match-tile step
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

// huffman-only / store-only streams with sync-flush points (flate_hip_compress_flush): the block
// table is built on the host, one entry per plan slot; ordinary batches compute the same from
// the chunk length (block j = bytes [65535 j, ...), deflate.zig:498-511).
struct fl_sblock {
    uint32_t start, len;  // chunk-relative byte range
    uint32_t flags;       // bit0: final block of the stream; bit1: the empty stored block after a flush
};

// call-wide constants
// fl_params.flags
#define FL_PRM_REPAIR_Q1 1u  // flate_hip_set_flags(FLATE_HIP_DEFLATE_REPAIR_Q1): a block is handed the bytes its tokens cover (below: q1_gap)
struct fl_params {
    uint32_t n_chunks;
    uint32_t n_blocks;
    int32_t container;  // 0 raw, 1 gzip, 2 zlib
    int32_t mode;       // 0 store, 1 huffman, 4..9
    // level args (deflate.zig:41-52)
    uint32_t good, lazy, nice, chain;
    uint32_t flags;   // FL_PRM_*
    uint32_t stream;  // non-zero: whole-stream pass (kernels_stream.h)
    uint32_t plan_dynamic_only;  // debug seam only: plan token blocks as BlockWriter.dynamicBlock does
};

// CRC-32 helper constants computed on the host once (reflected representation,
// x^0 = 0x80000000): xpow8[j] = x^(8 * 2^j) mod P, pow1024[m] = x^(8*1024*m) mod P.
struct fl_crc_consts {
    uint32_t xpow8[32];
    uint32_t pow1024[64];
    uint32_t pow65535;  // x^(8*65535)
};

