// CPU build of the serial block planner in flate_amd/csrc/flate_common.h.
// TEST INFRASTRUCTURE ONLY: lets the exact source one GPU lane executes be
// checked (and sanitised) on the CPU against the oracle and the reference's
// golden block vectors.  Not linked into libflate_hip.so.
#include <string.h>

#include "../../flate_amd/csrc/flate_common.h"

extern "C" {

int shim_plan_sizeof() { return (int)sizeof(fl_block_plan); }

// mode 0: token block (BlockWriter.write); mode 1: huffman-only block; mode 2: token block as
// BlockWriter.dynamicBlock plans it.
void shim_plan_block(int mode, const uint16_t* lit_freq, const uint16_t* dist_freq, uint32_t in_len,
                     uint32_t eof, fl_block_plan* plan) {
    static fl_plan_ws ws;
    memset(&ws, 0xA5, sizeof ws);  // poison: the planner must not rely on zeroed scratch
    memcpy(ws.lit_freq, lit_freq, sizeof ws.lit_freq);
    memcpy(ws.dist_freq, dist_freq, sizeof ws.dist_freq);
    memset(plan, 0, sizeof *plan);
    if (mode == 0 || mode == 2)
        fl_plan_token_block(&ws, plan, in_len, eof, mode == 2);
    else
        fl_plan_huffman_block(&ws, plan, in_len, eof);
}

void shim_huff_generate(const uint16_t* freq, uint32_t n, uint32_t max_bits, uint16_t* codes, uint16_t* lens) {
    static fl_plan_ws ws;
    memset(&ws, 0xA5, sizeof ws);
    fl_hcode out[FL_NUM_LIT];
    fl_huff_generate(&ws, freq, n, max_bits, out);
    for (uint32_t i = 0; i < n; i++) {
        codes[i] = out[i].len ? out[i].code : 0;
        lens[i] = out[i].len;
    }
}

// which form of the Huffman bit counts the planner runs: 0 = the reference's lazy loop, 1 = package-merge
// (the form the GPU runs)
void shim_set_pm(int on) { fl_plan_cpu_use_pm = on; }

void shim_tables(uint8_t* len_index /*256*/, uint8_t* len_extra /*29*/, uint8_t* len_base /*29*/,
                 uint8_t* dist_code /*32768*/, uint8_t* dist_extra /*30*/, uint16_t* dist_base /*30*/) {
    for (uint32_t i = 0; i < 256; i++) len_index[i] = (uint8_t)fl_len_index(i);
    for (uint32_t i = 0; i < 29; i++) {
        len_extra[i] = (uint8_t)fl_len_extra_bits(i);
        len_base[i] = (uint8_t)fl_len_base_scaled(i);
    }
    for (uint32_t i = 0; i < 32768; i++) dist_code[i] = (uint8_t)fl_dist_code(i);
    for (uint32_t i = 0; i < 30; i++) {
        dist_extra[i] = (uint8_t)fl_dist_extra_bits(i);
        dist_base[i] = (uint16_t)fl_dist_base_scaled(i);
    }
}
}

// ---- host tables of a whole-stream pass (flate_amd/csrc/stream_tables.h) ----
#include "../../flate_amd/csrc/stream_tables.h"

extern "C" {
// Builds the tables of ONE stream of n bytes with the given flush points.  Returns the counts
// through the out parameters; arrays are filled up to their capacities.
int shim_stream_tables(uint32_t n, const uint64_t* flush_pos, uint32_t n_flush, int finish, uint32_t* n_blocks,
                       uint32_t* n_slides, uint32_t* zones, uint32_t zones_cap, uint32_t* tiles /* w0,tgt0,zone */,
                       uint32_t tiles_cap, uint32_t* n_tiles, uint32_t* pieces /* start,end,first_block,n_blocks,seg0,n_seg,flags */,
                       uint32_t pieces_cap, uint32_t* n_pieces, uint32_t* segs /* piece,h0 */, uint32_t segs_cap,
                       uint32_t* n_segs) {
    StreamTables t;
    fl_chunk c{};
    c.in_len = n;
    FlushSpec fs{flush_pos, n_flush, finish != 0};
    add_stream_chunk(t, c, 0, 0, (n_flush || !finish) ? &fs : nullptr);
    *n_blocks = c.n_blocks;
    *n_slides = c.n_slides;
    for (uint32_t i = 0; i < t.zones.size() && i < zones_cap; i++) zones[i] = t.zones[i];
    *n_tiles = (uint32_t)t.tiles.size();
    for (uint32_t i = 0; i < t.tiles.size() && i < tiles_cap; i++) {
        tiles[3 * i] = t.tiles[i].w0;
        tiles[3 * i + 1] = t.tiles[i].tgt0;
        tiles[3 * i + 2] = t.tiles[i].zone;
    }
    *n_pieces = (uint32_t)t.pieces.size();
    for (uint32_t i = 0; i < t.pieces.size() && i < pieces_cap; i++) {
        const fl_piece& p = t.pieces[i];
        const uint32_t v[7] = {p.start, p.end, p.first_block, p.n_blocks, p.seg0, p.n_seg, p.flags};
        for (int k = 0; k < 7; k++) pieces[7 * i + k] = v[k];
    }
    *n_segs = (uint32_t)t.segs.size();
    for (uint32_t i = 0; i < t.segs.size() && i < segs_cap; i++) {
        segs[2 * i] = t.segs[i].piece;
        segs[2 * i + 1] = t.segs[i].h0;
    }
    return 0;
}
}
