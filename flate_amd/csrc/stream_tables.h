// stream_tables.h -- host side of a whole-stream pass (kernels_stream.h): which tiles, pieces and
// segments a stream with its sync-flush points turns into, and the slide table.  Plain C++ (no
// HIP): flate_hip.hip uses it, and tests/cpu_shim compiles it for the CPU tests.
#pragma once
#include <stdint.h>

#include <vector>

#include "flate_layout.h"

// Sync-flush points of one stream (flate_hip_compress_flush); absent for ordinary batches.
struct FlushSpec {
    const uint64_t* pos;
    uint32_t n;
    bool finish;
};

// Host-built tables of one whole-stream pass (kernels_stream.h).
struct StreamTables {
    std::vector<fl_tile> tiles;
    std::vector<fl_piece> pieces;
    std::vector<fl_seg> segs;
    std::vector<uint32_t> fpts;   // flush points of all chunks
    std::vector<uint32_t> zones;  // per chunk: first position visited after slide j
    uint64_t npos = 0;
    bool any_flush = false;
};

// Adds chunk `i` (pass-local index) to the tables; fills the chunk's stream fields and n_blocks.
inline void add_stream_chunk(StreamTables& t, fl_chunk& c, uint32_t i, uint32_t first_block, const FlushSpec* fs) {
    const uint32_t N = c.in_len;
    c.pos_off = t.npos;
    t.npos += ((uint64_t)N + FL_SEG - 1) / FL_SEG * FL_SEG + FL_SEG;
    // flush points
    c.flush_off = (uint32_t)t.fpts.size();
    c.n_flush = fs ? fs->n : 0;
    for (uint32_t k = 0; k < c.n_flush; k++) t.fpts.push_back((uint32_t)fs->pos[k]);
    if (c.n_flush) t.any_flush = true;
    // slide j happens when the window is full for the j-th time; position p is visited after it
    // iff p is within min_lookahead of the window end (SlidingWindow.zig:56-60) -- unless a flush
    // that came before the window was full ran the tokenizer up to its own position first
    c.zone_off = (uint32_t)t.zones.size();
    c.n_slides = N >= 65536u ? (N - 65536u) / FL_SEG + 1 : 0;
    for (uint32_t j = 1; j <= c.n_slides; j++) {
        const uint32_t full = 65536u + FL_SEG * (j - 1);
        uint32_t z = full - (FL_MAX_MATCH + 4u);
        for (uint32_t k = 0; k < c.n_flush; k++) {
            const uint32_t f = t.fpts[c.flush_off + k];
            if (f < full && f > z) z = f;
        }
        t.zones.push_back(z);
    }
    auto zone_of = [&](uint32_t j, uint32_t w0) -> uint32_t {  // window-relative, 65536 = no such slide
        if (j > c.n_slides) return 65536u;
        return t.zones[c.zone_off + j - 1] - w0;
    };
    t.tiles.push_back(fl_tile{i, 0u, 0u, zone_of(1, 0)});
    for (uint32_t w0 = FL_SEG; w0 + FL_SEG < N; w0 += FL_SEG)
        t.tiles.push_back(fl_tile{i, w0, FL_SEG, zone_of(w0 / FL_SEG + 1, w0)});
    // pieces, their blocks and segments
    c.piece0 = (uint32_t)t.pieces.size();
    uint32_t nb = 0;
    auto add_piece = [&](uint32_t start, uint32_t end, uint32_t flags) {
        fl_piece pc{};
        pc.chunk = i;
        pc.start = start;
        pc.end = end;
        pc.first_block = first_block + nb;
        pc.n_blocks = (end - start) / FL_SEG + 1 + ((flags & 2) ? 1 : 0);  // deflate.zig:227-230 (+ marker)
        pc.seg0 = (uint32_t)t.segs.size();
        pc.flags = flags;
        const uint32_t pi = (uint32_t)t.pieces.size();
        if (start < end)  // (an empty piece -- flush called twice -- has no segment, just its empty block)
            for (uint64_t h0 = start & ~(FL_SEG - 1); h0 < end; h0 += FL_SEG) t.segs.push_back(fl_seg{pi, (uint32_t)h0});
        pc.n_seg = (uint32_t)t.segs.size() - pc.seg0;
        nb += pc.n_blocks;
        t.pieces.push_back(pc);
    };
    uint32_t prev = 0;
    for (uint32_t k = 0; k < c.n_flush; k++) {
        const uint32_t f = t.fpts[c.flush_off + k];
        add_piece(prev, f, 2u);
        prev = f;
    }
    if (!fs || fs->finish) add_piece(prev, N, 1u);
    c.n_piece = (uint32_t)t.pieces.size() - c.piece0;
    c.n_blocks = nb;
}

