// Micro-benchmark behind DESIGN.md section 6 (config 5, flush phases): what one dependent round of random 8-byte table
// accesses costs a 512-thread workgroup whose table (64 MB) is far larger than the caches, as a function of the number of
// independent accesses a lane keeps in flight (K), with 1 or 2 workgroups per CU, for loads alone and for load -> atomic.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/random_access tests/micro/random_access.hip ; run: /tmp/random_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int K, int MODE> __global__ __launch_bounds__(512) void probe(uint64_t *tabs, uint32_t hbits, uint32_t rounds, unsigned long long *cyc, uint64_t *sink, uint32_t wbits)
{
    uint64_t *tab = tabs + ((uint64_t)blockIdx.x << hbits);
    const uint32_t mask = (1u << wbits) - 1u;      // accesses stay in a window of 2^wbits slots of the workgroup's table (moved every 16 rounds)
    uint64_t acc = 0; uint32_t st = mix(blockIdx.x * 977u + threadIdx.x);
    const unsigned long long t0 = clock64();
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t s[K]; uint64_t v[K];
        const uint32_t wbase = wbits < hbits ? (mix((r >> 4) * 2654435761u + blockIdx.x) & ((1u << hbits) - 1u) & ~mask) : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) { st = mix(st + 0x9E3779B9u + (uint32_t)acc); s[k] = wbase + (st & mask); }     // next addresses depend on the previous round's data
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __hip_atomic_load(&tab[s[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] += __hip_atomic_fetch_add(&tab[s[k]], (uint64_t)(v[k] & 1u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63u) == 0) atomicAdd(cyc, t1 - t0);
    if (acc == 0x1234567u) sink[0] = acc;
}
template <int K, int MODE> double run(uint64_t *tabs, uint32_t hbits, uint32_t grid, uint32_t rounds, unsigned long long *d_cyc, uint64_t *sink, uint32_t wbits = 23)
{
    hipMemset(d_cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<K, MODE>), dim3(grid), dim3(512), 0, 0, tabs, hbits, rounds, d_cyc, sink, wbits);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    const double per_round_us = ms * 1e3 / rounds;
    printf("grid %4u window 2^%u slots K %d %s: %.2f us per round (%.0f clock64 ticks), %.2f G accesses/s chip-wide\n", grid, wbits, K, MODE ? "load->atomic" : "load        ",
           per_round_us, (double)c / (grid * 8.0) / rounds, (double)grid * 512.0 * K * (MODE ? 2 : 1) / (per_round_us * 1e3));
    return per_round_us;
}
int main()
{
    const uint32_t hbits = 23, maxgrid = 512;
    uint64_t *tabs; unsigned long long *d_cyc; uint64_t *sink;
    if (hipMalloc(&tabs, ((size_t)maxgrid << hbits) * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tabs, 0, ((size_t)maxgrid << hbits) * 8);
    hipMalloc(&d_cyc, 8); hipMalloc(&sink, 8);
    const uint32_t rounds = 2000;
    for (uint32_t grid : {64u, 256u, 512u}) {
        run<1, 0>(tabs, hbits, grid, rounds, d_cyc, sink); run<2, 0>(tabs, hbits, grid, rounds, d_cyc, sink);
        run<4, 0>(tabs, hbits, grid, rounds, d_cyc, sink); run<8, 0>(tabs, hbits, grid, rounds, d_cyc, sink);
        run<1, 1>(tabs, hbits, grid, rounds, d_cyc, sink); run<2, 1>(tabs, hbits, grid, rounds, d_cyc, sink);
        run<4, 1>(tabs, hbits, grid, rounds, d_cyc, sink);
    }
    // locality: the same accesses confined to a window of the table (a segment) that moves every 16 rounds
    for (uint32_t wb : {10u, 13u, 15u, 17u, 20u}) { run<2, 0>(tabs, hbits, 512, rounds, d_cyc, sink, wb); run<2, 1>(tabs, hbits, 512, rounds, d_cyc, sink, wb); }
    return 0;
}
