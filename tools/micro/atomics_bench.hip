// Microbenchmark (GPU box): throughput of N random atomicMax updates on a table of M 64-bit keys -- the RESOLVE feed at
// the end of a bid kernel (one update per bidder).  Variants: scope, width, the issuing XCD owning a slice of the table.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/atomics_bench.hip -o build_ab/atomics_bench && build_ab/atomics_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *tab, unsigned *tab32, int M, int per_wg, unsigned seed) {
    const int t = threadIdx.x;
    if (t >= per_wg) return;
    const unsigned id = blockIdx.x * per_wg + t;
    const unsigned h = hash32(id * 2654435761u + seed);
    unsigned idx = h % (unsigned)M;
    const unsigned long long key = ((unsigned long long)hash32(h) << 20) | id;
    if (MODE == 0) __hip_atomic_fetch_max(&tab[idx], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_fetch_max(&tab[idx], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 2) __hip_atomic_fetch_max(&tab32[idx], (unsigned)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) {  // the table in 8 slices, a workgroup touches the slice of "its" XCD (round-robin dispatch)
        const unsigned sl = M / 8;
        idx = (blockIdx.x % 8) * sl + h % sl;
        __hip_atomic_fetch_max(&tab[idx], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 4) tab[idx] = key;  // plain random stores
    if (MODE == 5) __hip_atomic_fetch_max(&tab[idx], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 6) __hip_atomic_fetch_add(&tab32[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 7) {  // one update per 128-byte line at most from a wavefront?  no: the same random pattern, keys 16x apart
        __hip_atomic_fetch_max(&tab[(size_t)idx * 16], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200000, M = argc > 2 ? atoi(argv[2]) : 200000;
    const int wgs = 256, per_wg = (N + wgs - 1) / wgs;
    unsigned long long *tab;
    unsigned *tab32;
    hipMalloc(&tab, sizeof(unsigned long long) * (size_t)M * 16);
    hipMalloc(&tab32, sizeof(unsigned) * (size_t)M);
    hipMemset(tab, 0, sizeof(unsigned long long) * (size_t)M * 16);
    hipMemset(tab32, 0, sizeof(unsigned) * (size_t)M);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"u64 max, agent scope", "u64 max, system scope", "u32 max, agent scope", "u64 max, agent, XCD-local slices",
                           "plain u64 stores", "u64 max, workgroup scope", "u32 add, agent scope", "u64 max, agent, one key per 128-B line"};
#define RUN(MODE)                                                                                               \
    {                                                                                                           \
        float best = 1e9f, sum = 0;                                                                             \
        for (int r = 0; r < 12; ++r) {                                                                          \
            hipMemsetAsync(tab, 0, sizeof(unsigned long long) * (size_t)M * (MODE == 7 ? 16 : 1), 0);           \
            hipMemsetAsync(tab32, 0, sizeof(unsigned) * (size_t)M, 0);                                          \
            hipExtLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), 0, 0, e0, e1, 0, tab, tab32, M, per_wg, 77u + r); \
            hipEventSynchronize(e1);                                                                            \
            float ms;                                                                                           \
            hipEventElapsedTime(&ms, e0, e1);                                                                   \
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }                                                \
        }                                                                                                       \
        printf("%-42s N=%d M=%d: avg %.1f us  best %.1f us  (%.1f G updates/s)\n", names[MODE], N, M, sum / 10 * 1e3, \
               best * 1e3, N / (sum / 10 * 1e-3) / 1e9);                                                        \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
