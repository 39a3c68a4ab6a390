// Micro-benchmark behind DESIGN.md (config 5, the sweep's aggregation): throughput of the LDS operations a trip of the flat
// sweep issues -- random-address ds_add_u32 (no return), ds_cmpst_rtn_b32, ds_or_b32, ds_read_b64, ds_read_u16, ds_bpermute_b32 and a
// plain ds_write_b32 -- per CU, with 8 or 16 waves resident (one or two workgroups of 512 threads per CU), 64 or 8 active lanes.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_ops tests/micro/lds_ops.hip ; run: /tmp/lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
#define WORDS 12288u      /* 48 KB of LDS per workgroup: the aggregation table's size */
template <int OP> __global__ __launch_bounds__(512) void k(uint32_t rounds, uint32_t lanes, unsigned long long *cyc, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[WORDS];
    for (uint32_t i = threadIdx.x; i < WORDS; i += blockDim.x) tab[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u; const bool on = lane < lanes;
    uint32_t st = mix(blockIdx.x * 977u + threadIdx.x), acc = 0;
    uint32_t a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { st = mix(st + 0x9E3779B9u); a[u] = st % WORDS; }      // eight addresses per lane, reused every round: the loop is the LDS operation only
    const unsigned long long t0 = clock64();
    for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t s = a[u];
            if (OP == 0) { if (on) __hip_atomic_fetch_add(&tab[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (OP == 1) { if (on) { uint32_t e = 0xFFFFFFFFu; __hip_atomic_compare_exchange_strong(&tab[s], &e, r, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += e; } }
            if (OP == 2) { if (on) __hip_atomic_fetch_or(&tab[s], 1u << (r & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (OP == 3) { if (on) { const unsigned long long v = __hip_atomic_load((unsigned long long *)__builtin_assume_aligned(&tab[s & ~1u], 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += (uint32_t)v + (uint32_t)(v >> 32); } }
            if (OP == 4) { if (on) acc += ((volatile uint16_t *)tab)[s]; }
            if (OP == 5) { acc += (uint32_t)__builtin_amdgcn_ds_bpermute((int)((s & 63u) << 2), (int)acc + u); }
            if (OP == 6) { if (on) __hip_atomic_store(&tab[s], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (OP == 7) { if (on) acc += __hip_atomic_fetch_add(&tab[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }      // with return
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    if (acc == 0x1234567u) sink[0] = acc + tab[acc % WORDS];
}
template <int OP> void run(const char *name, uint32_t grid, uint32_t lanes, unsigned long long *d_cyc, uint32_t *sink)
{
    const uint32_t rounds = 4000;
    hipMemset(d_cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(512), 0, 0, 10u, lanes, d_cyc, sink);      // warm
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(512), 0, 0, rounds, lanes, d_cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave_instr = (double)c / (grid * 8.0) / (rounds * 8.0);              // cycles a wave spends per instruction
    const double wgs_per_cu = grid / 256.0, waves = 8.0 * (wgs_per_cu < 1 ? 1 : wgs_per_cu);
    printf("%-22s %2u lanes, %4u workgroups: %7.1f cycles per wave-instruction seen by a wave, %6.1f cycles of the CU per wave-instruction (%.0f waves per CU), kernel %.2f ms\n",
           name, lanes, grid, per_wave_instr, per_wave_instr / waves, waves, ms);
}
int main()
{
    unsigned long long *d_cyc; uint32_t *sink;
    hipMalloc(&d_cyc, 8); hipMalloc(&sink, 8);
    for (uint32_t grid : {256u, 512u}) for (uint32_t lanes : {64u, 8u}) {
        run<0>("ds_add_u32 (no return)", grid, lanes, d_cyc, sink);
        run<7>("ds_add_rtn_u32", grid, lanes, d_cyc, sink);
        run<1>("ds_cmpst_rtn_b32", grid, lanes, d_cyc, sink);
        run<2>("ds_or_b32", grid, lanes, d_cyc, sink);
        run<3>("ds_read_b64", grid, lanes, d_cyc, sink);
        run<4>("ds_read_u16", grid, lanes, d_cyc, sink);
        run<5>("ds_bpermute_b32", grid, lanes, d_cyc, sink);
        run<6>("ds_write_b32", grid, lanes, d_cyc, sink);
    }
    return 0;
}
