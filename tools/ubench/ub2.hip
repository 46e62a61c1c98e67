// Round-3 micro-benchmarks: what one wave64 integer VALU / SALU / LDS instruction costs on gfx950, and
// whether LDS atomics of one instruction are applied in lane order.  The numbers decide whether the
// tokenizer kernels are bound by issue or by latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o ub2.bin ub2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

// 8 independent chains x 8 = 64 instructions per unrolled body
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(k2));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(k));
#define OP_MAX(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define OP_MUL(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define OP_ALIGNBYTE(i) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(k2));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(k2));
#define OP_CMPCND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(k), "v"(k2) : "vcc");
#define OP_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
#define OP_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
#define OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define OP_DPPADD(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define OP_SADD(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(sk));
#define OP_SAND64(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s64[i]) : "s"(sk64));
#define OP_READLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[i]) : "v"(a[i]));
#define OP_CMPBALLOT(i) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(s64[i]) : "v"(a[i]), "v"(k));
#define OP_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(k));
#define OP_SWIZ(i) asm volatile("ds_swizzle_b32 %0, %0 offset:0x8000" : "+v"(a[i]));

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t a[8];
    uint32_t s[8];
    uint64_t s64[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * (2 * i + 3); s[i] = i + out[0]; s64[i] = i + out[0]; }
    const uint32_t k = out[0] | 1, k2 = out[1] | 0x01020304;
    const uint32_t sk = __builtin_amdgcn_readfirstlane(k);
    const uint64_t sk64 = sk;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { BODY8(OP_ADD) }
        if (MODE == 1) { BODY8(OP_XOR) }
        if (MODE == 2) { BODY8(OP_ANDOR) }
        if (MODE == 3) { BODY8(OP_LSHLADD) }
        if (MODE == 4) { BODY8(OP_MAX) }
        if (MODE == 5) { BODY8(OP_MUL) }
        if (MODE == 6) { BODY8(OP_ALIGNBYTE) }
        if (MODE == 7) { BODY8(OP_PERM) }
        if (MODE == 8) { BODY8(OP_CMPCND) }
        if (MODE == 9) { BODY8(OP_FFBL) }
        if (MODE == 10) { BODY8(OP_BCNT) }
        if (MODE == 11) { BODY8(OP_DPP) }
        if (MODE == 12) { BODY8(OP_DPPADD) }
        if (MODE == 13) { BODY8(OP_SADD) }
        if (MODE == 14) { BODY8(OP_SAND64) }
        if (MODE == 15) { BODY8(OP_READLANE) }
        if (MODE == 16) { BODY8(OP_CMPBALLOT) }
        if (MODE == 17) { BODY8(OP_BPERM) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 18) { BODY8(OP_SWIZ) asm volatile("s_waitcnt lgkmcnt(0)"); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= a[i] ^ s[i] ^ (uint32_t)s64[i];
    out[2 + blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// dependent chain of one op kind: latency per instruction for a lone wave
template <int MODE>
__global__ void k_lat(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t a[1] = {threadIdx.x};
    const uint32_t k = out[0] | 1, k2 = out[1] | 0x01020304;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define L8(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)
        if (MODE == 0) { L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) L8(OP_ADD) }
        if (MODE == 5) { L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) L8(OP_MUL) }
        if (MODE == 11) { L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) L8(OP_DPP) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[2 + threadIdx.x] = a[0] ^ k2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS: dependent ds_read_b32 latency, atomic-add-with-return throughput from ONE wave (random banks),
// and the order in which one instruction's atomics to the same word are applied.
__global__ void k_lds_atomic_rate(uint32_t* out, int iters, unsigned long long* cyc) {
    __shared__ uint32_t tab[16384];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    uint32_t x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            acc += atomicAdd(&tab[x >> 18], 1u);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[2 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// every lane: old = atomicAdd(&w[key[lane]], 1): in lane order iff old == number of lower lanes with the same key.
// Several instructions back to back (no wait between them) must also keep program order.
__global__ void k_order(const uint32_t* keys, uint32_t* olds, int rounds) {
    __shared__ uint32_t w[4096];
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) w[i] = 0;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    for (int r = 0; r < rounds; r += 4) {
        const uint32_t k0 = keys[(r + 0) * 64 + threadIdx.x], k1 = keys[(r + 1) * 64 + threadIdx.x];
        const uint32_t k2 = keys[(r + 2) * 64 + threadIdx.x], k3 = keys[(r + 3) * 64 + threadIdx.x];
        const uint32_t o0 = atomicAdd(&w[k0], 1u);
        const uint32_t o1 = atomicAdd(&w[k1], 1u);
        const uint32_t o2 = atomicAdd(&w[k2], 1u);
        const uint32_t o3 = atomicAdd(&w[k3], 1u);
        olds[(r + 0) * 64 + threadIdx.x] = o0;
        olds[(r + 1) * 64 + threadIdx.x] = o1;
        olds[(r + 2) * 64 + threadIdx.x] = o2;
        olds[(r + 3) * 64 + threadIdx.x] = o3;
    }
}

template <int MODE>
static int run_rate(const char* name, uint32_t* d, unsigned long long* dc, int instr_per_body, double clk_ghz) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("%-22s", name);
    // waves per SIMD: block of 256 threads = 1 wave per SIMD; grid = 256 CUs x blocks
    const int cfg[4][2] = {{256, 256}, {512, 256}, {1024, 256}, {1024, 512}};  // {threads per block, blocks}: 1, 2, 4, 8 waves/SIMD
    for (int c = 0; c < 4; c++) {
        const int thr = cfg[c][0], blocks = cfg[c][1];
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(thr), 0, 0, d, 10, dc); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(thr), 0, 0, d, iters, dc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc; CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        const double waves_per_simd = (double)thr / 64 * blocks / 256 / 4;
        const double winstr_per_simd = waves_per_simd * iters * 64.0 * instr_per_body;
        // cycles per wave-instruction per SIMD, from the wall clock at the nominal clock and from the wave's own counter
        printf("  w/SIMD %.0f: %.2f cyc (wall@%.1fGHz) %.2f (counter)", waves_per_simd, ms * 1e6 * clk_ghz / winstr_per_simd,
               clk_ghz, (double)cyc / (iters * 64.0 * instr_per_body) / waves_per_simd);
    }
    printf("\n");
    return 0;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 4 * (1024 * 1024 + 16))); CK(hipMemset(d, 0, 64));
    uint32_t init[2] = {0x5a5a5a5b, 0x00010203}; CK(hipMemcpy(d, init, 8, hipMemcpyHostToDevice));
    unsigned long long* dc; CK(hipMalloc(&dc, 64));
    const double clk = 2.4;
    printf("cycles per wave64 instruction per SIMD (lower = faster); counter = __builtin_readcyclecounter of wave 0 / its instructions / waves per SIMD\n");
    run_rate<0>("v_add_u32", d, dc, 1, clk);
    run_rate<1>("v_xor_b32", d, dc, 1, clk);
    run_rate<2>("v_and_or_b32", d, dc, 1, clk);
    run_rate<3>("v_lshl_add_u32", d, dc, 1, clk);
    run_rate<4>("v_max_u32", d, dc, 1, clk);
    run_rate<5>("v_mul_lo_u32", d, dc, 1, clk);
    run_rate<6>("v_alignbyte_b32", d, dc, 1, clk);
    run_rate<7>("v_perm_b32", d, dc, 1, clk);
    run_rate<8>("v_cmp+v_cndmask (2)", d, dc, 2, clk);
    run_rate<9>("v_ffbl_b32", d, dc, 1, clk);
    run_rate<10>("v_bcnt_u32_b32", d, dc, 1, clk);
    run_rate<11>("v_mov_b32_dpp", d, dc, 1, clk);
    run_rate<12>("v_add_u32_dpp", d, dc, 1, clk);
    run_rate<13>("s_add_u32", d, dc, 1, clk);
    run_rate<14>("s_and_b64", d, dc, 1, clk);
    run_rate<15>("v_readlane_b32", d, dc, 1, clk);
    run_rate<16>("v_cmp -> sgpr pair", d, dc, 1, clk);
    run_rate<17>("ds_bpermute_b32", d, dc, 1, clk);
    run_rate<18>("ds_swizzle_b32", d, dc, 1, clk);
    {
        unsigned long long cyc;
        hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, d, 1000, dc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent v_add_u32, lone wave: %.2f counter ticks per instruction\n", (double)cyc / 64000.0);
        hipLaunchKernelGGL(k_lat<5>, dim3(1), dim3(64), 0, 0, d, 1000, dc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent v_mul_lo_u32, lone wave: %.2f\n", (double)cyc / 64000.0);
        hipLaunchKernelGGL(k_lat<11>, dim3(1), dim3(64), 0, 0, d, 1000, dc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent v_mov_b32_dpp, lone wave: %.2f\n", (double)cyc / 64000.0);
        // counter tick vs wall: a lone wave running a known number of ticks
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, d, 200000, dc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        printf("counter ticks per microsecond of wall time: %.1f (kernel %.3f ms, %llu ticks)\n", (double)cyc / (ms * 1e3), ms, cyc);
        hipLaunchKernelGGL(k_lds_atomic_rate, dim3(1), dim3(64), 0, 0, d, 2000, dc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("ds_add_rtn_u32 random addresses, one wave, 8 in flight: %.1f ticks per wave-instruction\n", (double)cyc / 16000.0);
    }
    // order of LDS atomics
    {
        const int rounds = 4096;
        std::vector<uint32_t> keys(rounds * 64), olds(rounds * 64);
        uint64_t x = 88172645463325252ull;
        for (int r = 0; r < rounds; r++) {
            const int kind = r % 4;  // all equal, few values, random in 4096, bank-conflicting strides
            for (int l = 0; l < 64; l++) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                keys[r * 64 + l] = kind == 0 ? (uint32_t)(r % 4096) : kind == 1 ? (uint32_t)(x % 5) : kind == 2 ? (uint32_t)(x % 4096) : (uint32_t)((x % 8) * 32 + (r % 32));
            }
        }
        uint32_t *dk, *dold; CK(hipMalloc(&dk, keys.size() * 4)); CK(hipMalloc(&dold, keys.size() * 4));
        CK(hipMemcpy(dk, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
        long bad = 0, total = 0;
        for (int nthr = 64; nthr <= 1024; nthr *= 4) {  // with idle neighbours too
            hipLaunchKernelGGL(k_order, dim3(64), dim3(nthr), 0, 0, dk, dold, rounds); CK(hipDeviceSynchronize());
            CK(hipMemcpy(olds.data(), dold, olds.size() * 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> cnt(4096, 0);
            for (int r = 0; r < rounds; r++)
                for (int l = 0; l < 64; l++) { const uint32_t kk = keys[r * 64 + l]; if (olds[r * 64 + l] != cnt[kk]) bad++; cnt[kk]++; total++; }
        }
        printf("LDS atomic order: %ld of %ld returned values differ from lane order / program order\n", bad, total);
    }
    return 0;
}
