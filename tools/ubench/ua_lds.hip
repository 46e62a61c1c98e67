// Unaligned LDS reads on gfx950: does ds_read_b32 / ds_read_b64 at a byte-unaligned address return the right bytes,
// and what does it cost against 3 aligned dwords + 2 v_alignbyte?   hipcc --offload-arch=gfx950 -O3 ua_lds.hip -o ua_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct __attribute__((packed)) u64p { uint64_t v; };
__global__ void k(const uint32_t* in, uint64_t* out, uint32_t mul, int mode, int iters, uint64_t* cyc) {
    __shared__ uint8_t buf[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((uint32_t*)buf)[i] = in[i];
    __syncthreads();
    uint32_t off = (threadIdx.x * mul) & 65527u;
    uint64_t acc = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint64_t v;
        if (mode == 0) {
            v = ((const u64p*)(buf + off))->v;
        } else {
            const uint32_t* w = (const uint32_t*)buf + (off >> 2);
            const uint32_t sh = off & 3;
            const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
            v = (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
        }
        acc += v;
        off = (off + (uint32_t)(v & 0xff) * 8 + 13) & 65519u;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
    std::vector<uint32_t> h(16384);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)s; }
    uint32_t* din; uint64_t *dout, *dcyc;
    hipMalloc(&din, 65536); hipMalloc(&dout, 2 * 1024 * 8 * 256); hipMalloc(&dcyc, 16);
    hipMemcpy(din, h.data(), 65536, hipMemcpyHostToDevice);
    std::vector<uint64_t> r0(1024), r1(1024);
    for (uint32_t mul : {1u, 3u, 8u, 37u}) {
        k<<<1, 1024>>>(din, dout, mul, 0, 1000, dcyc); hipMemcpy(r0.data(), dout, 8192, hipMemcpyDeviceToHost);
        k<<<1, 1024>>>(din, dout, mul, 1, 1000, dcyc); hipMemcpy(r1.data(), dout, 8192, hipMemcpyDeviceToHost);
        uint64_t c[2]; hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; i++) bad += r0[i] != r1[i];
        printf("mul %u: mismatches %d  cycles/iter unaligned-b64 %.1f  3xb32+alignbyte %.1f\n", mul, bad, c[0] / 1000.0, c[1] / 1000.0);
    }
    return 0;
}
