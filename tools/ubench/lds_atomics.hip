// Round 6: what an ordered hash-table pass costs the LDS (k_lz_chain is bound by its returning atomics: 1024 wave instructions a
// chunk at about 46 cycles each).  Wave instructions per cycle and CU for the exchange as built (ds_mskor_rtn_b32 on 16384 words
// holding two 16-bit heads), a plain 32-bit exchange on 32768 words (ds_wrxchg_rtn_b32: one data operand), ds_max_rtn_u32, and the
// exchange without its return value -- random word addresses, 16 instructions in flight per wave, 8 or 16 waves a workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_atomics.bin tools/ubench/lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <int OP, int WORDS>
__global__ void k(uint32_t* out, int iters) {
    extern __shared__ uint32_t tab[];
    for (int i = threadIdx.x; i < WORDS; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint32_t r[16];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t w = (x >> 9) % WORDS;
            lds_u32* a = (lds_u32*)&tab[w];
            const uint32_t v = (uint32_t)it * 64u + threadIdx.x;
            if (OP == 0) asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=v"(r[s]) : "v"(a), "v"((x & 1u) ? 0xffff0000u : 0xffffu), "v"((x & 1u) ? v << 16 : (v & 0xffffu)) : "memory");
            else if (OP == 1) asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "=v"(r[s]) : "v"(a), "v"(v) : "memory");
            else if (OP == 2) asm volatile("ds_max_rtn_u32 %0, %1, %2" : "=v"(r[s]) : "v"(a), "v"(v) : "memory");
            else if (OP == 3) { asm volatile("ds_max_u32 %0, %1" : : "v"(a), "v"(v) : "memory"); r[s] = 0; }
            else { asm volatile("ds_read_b32 %0, %1" : "=v"(r[s]) : "v"(a) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 16; s++) acc += r[s];
    }
    const uint64_t dt = __builtin_readcyclecounter() - t0;
    if (threadIdx.x == 0) out[2 * blockIdx.x] = (uint32_t)dt;
    if (threadIdx.x == 0) out[2 * blockIdx.x + 1] = acc;
}
template <int OP, int WORDS>
void run(const char* name, int threads, int wg_per_cu) {
    uint32_t* d; hipMalloc(&d, 8 * 4096);
    const int iters = 200, grid = 256 * wg_per_cu;
    hipLaunchKernelGGL((k<OP, WORDS>), dim3(grid), dim3(threads), WORDS * 4, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<OP, WORDS>), dim3(grid), dim3(threads), WORDS * 4, 0, d, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double instr_per_cu = (double)iters * 16 * (threads / 64) * wg_per_cu;
    printf("%-28s %4d threads x %d WG/CU: %.3f ms, %.1f cycles (shader clock, one workgroup) per wave instruction, %.1f ns per wave instruction and CU\n", name, threads, wg_per_cu, ms,
           (double)h[0] / (iters * 16.0 * (threads / 64)), ms * 1e6 / instr_per_cu);
    hipFree(d);
}
int main() {
    run<0, 16384>("ds_mskor_rtn_b32 / 64 KiB", 512, 2);
    run<0, 16384>("ds_mskor_rtn_b32 / 64 KiB", 512, 1);
    run<1, 32768>("ds_wrxchg_rtn_b32 / 128 KiB", 1024, 1);
    run<1, 32768>("ds_wrxchg_rtn_b32 / 128 KiB", 512, 1);
    run<1, 16384>("ds_wrxchg_rtn_b32 / 64 KiB", 512, 2);
    run<2, 16384>("ds_max_rtn_u32 / 64 KiB", 512, 2);
    run<3, 16384>("ds_max_u32 (no return) / 64 KiB", 512, 2);
    run<4, 16384>("ds_read_b32 / 64 KiB", 512, 2);
    return 0;
}
