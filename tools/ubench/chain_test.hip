// k_lz_chain against the reference's chain construction (Lookup.zig:35-40) on a few synthetic chunks
#include "../../flate_amd/csrc/kernels_parse.h"
#include <cstdio>
#include <vector>
#include <cstring>
int main() {
    const int NCH = 4;
    const uint32_t lens[NCH] = {65535, 65535, 1000, 40000};
    std::vector<uint8_t> in;
    std::vector<fl_chunk> ch(NCH);
    uint64_t x = 88172645463325252ull;
    for (int c = 0; c < NCH; c++) {
        memset(&ch[c], 0, sizeof(fl_chunk));
        ch[c].in_off = in.size(); ch[c].in_len = lens[c]; ch[c].pos_off = (uint64_t)c * 65536;
        for (uint32_t i = 0; i < lens[c]; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint8_t b = c == 0 ? (uint8_t)('a' + (x % 6)) : c == 1 ? (uint8_t)((i / 700) & 1 ? 0 : 'a' + (x % 3)) : (uint8_t)(x % 251);
            in.push_back(b);
        }
        in.push_back(0); in.push_back(0); in.push_back(0);  // odd alignment of the next chunk
    }
    in.resize(in.size() + 64);
    uint8_t* d_in; fl_chunk* d_ch; uint16_t* d_prev; uint32_t* d_cf; hipMalloc(&d_cf, 4 * NCH);
    hipMalloc(&d_in, in.size()); hipMalloc(&d_ch, sizeof(fl_chunk) * NCH); hipMalloc(&d_prev, 2 * 65536 * NCH);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice); hipMemcpy(d_ch, ch.data(), sizeof(fl_chunk) * NCH, hipMemcpyHostToDevice);
    hipMemset(d_prev, 0xee, 2 * 65536 * NCH);
    hipLaunchKernelGGL(k_lz_chain<false>, dim3(NCH), dim3(64 * FL_CHAIN_WAVES), 0, 0, d_in, d_ch, d_prev, d_cf, (uint32_t*)nullptr, (const uint32_t*)nullptr);
    hipError_t e = hipDeviceSynchronize();
    printf("kernel: %s\n", hipGetErrorString(e));
    std::vector<uint16_t> prev(65536 * NCH);
    hipMemcpy(prev.data(), d_prev, 2 * 65536 * NCH, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < NCH; c++) {
        std::vector<uint16_t> head(32768, 0);
        const uint8_t* s = in.data() + ch[c].in_off;
        for (uint32_t p = 0; p + 4 <= lens[c]; p++) {
            uint32_t v = (uint32_t)s[p + 3] | (uint32_t)s[p + 2] << 8 | (uint32_t)s[p + 1] << 16 | (uint32_t)s[p] << 24;
            uint32_t h = (v * 0x9E3779B1u) >> 17;
            uint16_t exp = head[h]; head[h] = (uint16_t)p;
            if (prev[c * 65536 + p] != exp) { if (bad < 10) printf("chunk %d pos %u: got %u expected %u\n", c, p, prev[c * 65536 + p], exp); bad++; }
        }
    }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
