// What does a scattered global gather cost a CU on gfx950?  The round-4 match finder follows hash-chain links that
// live in global memory (L2-resident: a few hundred KB per workgroup); every step of a lane is a dependent 4- or
// 8-byte load at a random address of its chunk's region.  Measured here: cycles per lane-load per CU as a function of
// active lanes per wave, bytes per load, waves per CU, region size; and LDS gathers (ds_read_u16) for comparison.
//   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

// every workgroup chases pointers inside its own region of `region` dwords (index -> next index), `active` lanes
// of each wave take part; mode 0: 4-byte loads, 1: 8-byte loads (two dwords, the first is the link), 2: 16-byte loads
template <int MODE>
__global__ void k_gather(const uint32_t* __restrict__ links, uint32_t region, int active, int iters, uint32_t* out,
                         unsigned long long* cyc) {
    const uint32_t* base = links + (size_t)blockIdx.x * region;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (threadIdx.x * 2654435761u) % region;
    uint32_t acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((int)lane < active) {
        for (int it = 0; it < iters; it++) {
            uint32_t got;
            if (MODE == 0) {
                got = ((const uint16_t*)base)[idx * 2];   // 2-byte loads (the links are 16 bit)
            } else if (MODE == 1) {
                const uint2 v = *(const uint2*)(base + (idx & ~1u));
                got = v.x; acc += v.y;
            } else {
                const uint4 v = *(const uint4*)(base + (idx & ~3u));
                got = v.x; acc += v.y + v.z + v.w;
            }
            // the table holds zeros: the next index depends on the loaded value, but is a fresh pseudo-random position
            idx = ((idx + got) * 1664525u + 1013904223u + (uint32_t)it * 2654435761u) % region;
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx + acc;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

// the same chase through a 64 KiB table in LDS (16-bit links), for comparison
__global__ void k_lds(const uint32_t* __restrict__ links, int active, int iters, uint32_t* out, unsigned long long* cyc) {
    __shared__ uint16_t tab[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = (uint16_t)(links[i] & 32767u);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (threadIdx.x * 2654435761u) & 32767u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((int)lane < active)
        for (int it = 0; it < iters; it++) idx = tab[idx];
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

int main() {
    const int n_cu = 256;
    const uint32_t max_region = 1u << 18;  // dwords per workgroup (1 MiB)
    const int max_wg = n_cu * 2;
    std::vector<uint32_t> h((size_t)max_region * max_wg);
    uint64_t s = 88172645463325252ull;
    uint32_t *d, *out;
    unsigned long long* cyc;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&out, (size_t)max_wg * 1024 * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (uint32_t region : {1u << 14, 1u << 15, 1u << 16, 1u << 17}) {  // 64 KiB .. 512 KiB per workgroup
        for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = 0; }
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int wg_per_cu : {1, 2}) {
            for (int threads : {512, 1024}) {
                for (int active : {8, 32, 64}) {
                    for (int mode = 0; mode < 2; mode++) {
                        const int iters = 1000, grid = n_cu * wg_per_cu;
                        hipMemset(cyc, 0, 8);
                        // warm-up brings the regions into L2 / MALL
                        if (mode == 0) k_gather<0><<<grid, threads>>>(d, region, active, 50, out, cyc);
                        hipMemset(cyc, 0, 8);
                        hipEventRecord(e0);
                        if (mode == 0) k_gather<0><<<grid, threads>>>(d, region, active, iters, out, cyc);
                        if (mode == 1) k_gather<1><<<grid, threads>>>(d, region, active, iters, out, cyc);
                        if (mode == 2) k_gather<2><<<grid, threads>>>(d, region, active, iters, out, cyc);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms = 0;
                        hipEventElapsedTime(&ms, e0, e1);
                        unsigned long long c = 0;
                        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                        const double per_wg = (double)c / grid;  // cycles of one workgroup's loop
                        const double lane_loads_per_cu = (double)iters * active * (threads / 64) * wg_per_cu;
                        printf("total %4u MiB  region %4u KiB  wg/cu %d  threads %4d  active %2d  %2d B: %7.0f cyc per dependent load (latency), %5.2f cyc per lane-load per CU, %.3f ms\n",
                               (unsigned)((size_t)region * 4 * grid >> 20), region / 256, wg_per_cu, threads, active, mode == 0 ? 2 : 8, per_wg / iters, per_wg / lane_loads_per_cu * wg_per_cu / wg_per_cu * 1.0, ms);
                    }
                }
            }
        }
    }
    for (int threads : {512, 1024})
        for (int active : {8, 16, 32, 64}) {
            const int iters = 20000, grid = n_cu;
            hipMemset(cyc, 0, 8);
            k_lds<<<grid, threads>>>(d, active, iters, out, cyc);
            unsigned long long c = 0;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per_wg = (double)c / grid;
            printf("LDS u16 chase  threads %4d  active %2d: %6.1f cyc per dependent load, %5.3f cyc per lane-load per CU\n", threads, active,
                   per_wg / iters, per_wg / ((double)iters * active * (threads / 64)));
        }
    return 0;
}
