// Micro-benchmarks that shaped the round-2 match finder: VALU integer rate, LDS random-gather rate,
// unaligned LDS reads.  Build: hipcc --offload-arch=gfx950 -O3 -o ub ub.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__global__ __launch_bounds__(1024) void k_valu(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19;
    const uint32_t k = out[0];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a0 = (a0 ^ k) + a1; a1 = max(a1, a2) ^ k; a2 = (a2 & k) | a3; a3 = a3 - a4;
            a4 = (a4 ^ k) + a5; a5 = max(a5, a6) ^ k; a6 = (a6 & k) | a7; a7 = a7 - a0;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x + 1] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// mode 0: u8 gather, 1: u16 gather, 2: b32 aligned gather, 3: b32 unaligned gather (memcpy), 4: two aligned b32 + alignbyte, 5: b32 consecutive (no conflicts)
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(uint32_t* out, int iters) {
    __shared__ uint32_t buf[16384 + 4];
    for (uint32_t i = threadIdx.x; i < 16384 + 4; i += 1024) buf[i] = i * 2654435761u;
    __syncthreads();
    const uint8_t* b8 = (const uint8_t*)buf;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            uint32_t a = (x >> 16);  // 0..65535
            if (MODE == 0) acc += b8[a];
            else if (MODE == 1) acc += ((const uint16_t*)buf)[a >> 1];
            else if (MODE == 2) acc += buf[a >> 2];
            else if (MODE == 3) { uint32_t v; __builtin_memcpy(&v, b8 + a, 4); acc += v; }
            else if (MODE == 4) { uint32_t i2 = a >> 2; acc += __builtin_amdgcn_alignbyte(buf[i2 + 1], buf[i2], a & 3); }
            else if (MODE == 5) acc += buf[((x >> 30) * 64 + (threadIdx.x & 63) + u * 256) & 16383];
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x + 1] = acc;
}

__global__ void k_unal(uint32_t* out) {
    __shared__ uint32_t buf[256];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) buf[i] = i * 0x01010101u + 0x03020100u * 0;  // byte pattern
    uint8_t* b8 = (uint8_t*)buf;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1024; i += 64) b8[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t a = threadIdx.x * 5 + 1;
    uint32_t v; __builtin_memcpy(&v, b8 + a, 4);
    uint32_t e = 0; for (int j = 0; j < 4; j++) e |= (uint32_t)(uint8_t)((a + j) * 7 + 3) << (8 * j);
    out[threadIdx.x] = (v == e);
}

// LDS atomic order inside one instruction: every lane exchanges its lane id into the same word
__global__ void k_xchg(uint32_t* out) {
    __shared__ uint32_t w[64];
    if (threadIdx.x < 64) w[threadIdx.x] = 1000;
    __syncthreads();
    uint32_t key = out[64 + threadIdx.x];  // key per lane
    uint32_t old = atomicExch(&w[key], threadIdx.x);
    out[threadIdx.x] = old;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 4 * (1024 * 2048 + 16))); CK(hipMemset(d, 0, 64));
    uint32_t one = 0x5a5a5a5a; CK(hipMemcpy(d, &one, 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const int grid = 512;  // 2 WG per CU
    for (int rep = 0; rep < 2; rep++) {
        int iters = 4000;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_valu, dim3(grid), dim3(1024), 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        double winstr = (double)grid * 16 * iters * 8 * 12;  // 12 VALU-ish ops per u (xor,add,max,xor,and_or,sub ...)
        printf("valu: %.3f ms, ~%.2f wave-instr/ns chip (if 12 ops/u); per CU per ns %.3f\n", ms, winstr / ms / 1e6, winstr / ms / 1e6 / 256);
    }
#define RUN(M) { int iters = 2000; hipLaunchKernelGGL(k_lds<M>, dim3(grid), dim3(1024), 0, 0, d, 10); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds<M>, dim3(grid), dim3(1024), 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        CK(hipEventElapsedTime(&ms, e0, e1)); double wi = (double)grid * 16 * iters * 8; \
        printf("lds mode %d: %.3f ms, %.3f ns per wave-gather per CU\n", M, ms, ms * 1e6 / (wi / 256)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    hipLaunchKernelGGL(k_unal, dim3(1), dim3(64), 0, 0, d + 1024); CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(256); CK(hipMemcpy(h.data(), d + 1024, 256, hipMemcpyDeviceToHost));
    int ok = 0; for (int i = 0; i < 64; i++) ok += h[i]; printf("unaligned ds_read_b32 correct lanes: %d / 64\n", ok);
    // xchg order
    for (int pat = 0; pat < 3; pat++) {
        std::vector<uint32_t> k(64); for (int i = 0; i < 64; i++) k[i] = pat == 0 ? 0 : pat == 1 ? (i & 3) : (i * 7 % 5);
        CK(hipMemcpy(d + 2048 + 64, k.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_xchg, dim3(1), dim3(64), 0, 0, d + 2048); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d + 2048, 256, hipMemcpyDeviceToHost));
        int inorder = 1; std::vector<int> last(64, 1000);
        for (int i = 0; i < 64; i++) { if ((int)h[i] != last[k[i]]) inorder = 0; last[k[i]] = i; }
        printf("xchg pattern %d: lane order %s; olds:", pat, inorder ? "yes" : "NO"); for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
    }
    return 0;
}
