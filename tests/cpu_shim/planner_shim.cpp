// CPU build of the serial block planner in flate_amd/csrc/flate_common.h.
// TEST INFRASTRUCTURE ONLY: lets the exact source one GPU lane executes be
// checked (and sanitised) on the CPU against the oracle and the reference's
// golden block vectors.  Not linked into libflate_hip.so.
#include <string.h>

#include "../../flate_amd/csrc/flate_common.h"

extern "C" {

int shim_plan_sizeof() { return (int)sizeof(fl_block_plan); }

// mode 0: token block (BlockWriter.write); mode 1: huffman-only block.
void shim_plan_block(int mode, const uint16_t* lit_freq, const uint16_t* dist_freq, uint32_t in_len,
                     uint32_t eof, fl_block_plan* plan) {
    static fl_plan_ws ws;
    memset(&ws, 0xA5, sizeof ws);  // poison: the planner must not rely on zeroed scratch
    memcpy(ws.lit_freq, lit_freq, sizeof ws.lit_freq);
    memcpy(ws.dist_freq, dist_freq, sizeof ws.dist_freq);
    memset(plan, 0, sizeof *plan);
    if (mode == 0)
        fl_plan_token_block(&ws, plan, in_len, eof);
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
