// what DS_MSKOR_RTN_B32 does on gfx950, and in which order the lanes of one instruction are served
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    __shared__ unsigned tab[64];
    tab[threadIdx.x] = 0x11112222u;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)&tab[0];
    unsigned addr = base + (threadIdx.x & 3) * 4, hs = (threadIdx.x & 4) ? 16 : 0, mask = 0xffffu << hs, val = (threadIdx.x + 1) << hs, old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n s_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(mask), "v"(val) : "memory");
    out[threadIdx.x] = old;
    __syncthreads();
    if (threadIdx.x < 4) out[64 + threadIdx.x] = tab[threadIdx.x];
    if (threadIdx.x == 0) out[68] = base;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 80); k<<<1, 64>>>(d); unsigned h[80]; hipMemcpy(h, d, 4 * 80, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) printf("lane %2d word %d half %d -> old %08x\n", i, i & 3, (i >> 2) & 1, h[i]);
    for (int i = 0; i < 4; i++) printf("final word %d = %08x\n", i, h[64 + i]);
    printf("lds base %08x\n", h[68]);
    return 0;
}
