// Microbenchmark (GPU box): what shape of a streaming kernel reaches the HBM rate of an MI355X -- the shape
// misslap_measure_hbm then uses for the "measured peak" of bench.py.  1 GiB, read-only and copy; variants: how a
// workgroup walks the buffer (grid-stride over the whole buffer / one contiguous chunk per workgroup), workgroups per
// CU, loads in flight per lane, non-temporal loads.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream_bench.hip -o build_ab/stream_bench && build_ab/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int UNROLL, bool CHUNK, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4u *src, size_t n16, unsigned *sink) {
    size_t k, end, step;
    if (CHUNK) {  // workgroup b owns [b * per, (b + 1) * per)
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        k = (size_t)blockIdx.x * per + threadIdx.x;
        end = min(n16, (size_t)(blockIdx.x + 1) * per);
        step = 256;
    } else {
        k = (size_t)blockIdx.x * 256 + threadIdx.x;
        end = n16;
        step = (size_t)gridDim.x * 256;
    }
    unsigned acc = 0;
    for (; k + (UNROLL - 1) * step < end; k += UNROLL * step) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + k + u * step) : src[k + u * step];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; k < end; k += step) {
        const v4u a = src[k];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
template <int UNROLL, bool CHUNK, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4u *src, v4u *dst, size_t n16) {
    size_t k, end, step;
    if (CHUNK) {
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        k = (size_t)blockIdx.x * per + threadIdx.x;
        end = min(n16, (size_t)(blockIdx.x + 1) * per);
        step = 256;
    } else {
        k = (size_t)blockIdx.x * 256 + threadIdx.x;
        end = n16;
        step = (size_t)gridDim.x * 256;
    }
    for (; k + (UNROLL - 1) * step < end; k += UNROLL * step) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + k + u * step) : src[k + u * step];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + k + u * step);
            else dst[k + u * step] = v[u];
        }
    }
    for (; k < end; k += step) dst[k] = src[k];
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    v4u *src, *dst;
    unsigned *sink;
    CHECK(hipMalloc((void **)&src, bytes));
    CHECK(hipMalloc((void **)&dst, bytes));
    CHECK(hipMalloc((void **)&sink, 256));
    CHECK(hipMemset(src, 1, bytes));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    auto run = [&](const char *name, int wg_per_cu, auto launch, double moved) {
        const dim3 grid(wg_per_cu > 0 ? (unsigned)(cus * wg_per_cu) : (unsigned)(n16 / 256 / 4));
        for (int r = -3; r < reps; ++r) {
            if (r == 0) (void)hipEventRecord(e0, nullptr);
            launch(grid);
        }
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s wg/CU %3d  %8.1f GB/s\n", name, wg_per_cu, moved * reps / (ms * 1e-3) / 1e9);
    };
#define RD(U, C, N) [&](dim3 g) { hipLaunchKernelGGL((k_read<U, C, N>), g, dim3(256), 0, nullptr, src, n16, sink); }
#define CP(U, C, N) [&](dim3 g) { hipLaunchKernelGGL((k_copy<U, C, N>), g, dim3(256), 0, nullptr, src, dst, n16); }
    for (int w : {4, 8, 16, 32, 0}) {
        run("read  grid-stride x4", w, RD(4, false, false), (double)bytes);
        run("read  grid-stride x8", w, RD(8, false, false), (double)bytes);
        run("read  grid-stride x4 nt", w, RD(4, false, true), (double)bytes);
        run("read  chunk x4", w, RD(4, true, false), (double)bytes);
        run("read  chunk x8", w, RD(8, true, false), (double)bytes);
        run("read  chunk x4 nt", w, RD(4, true, true), (double)bytes);
        run("read  chunk x1", w, RD(1, true, false), (double)bytes);
        run("copy  grid-stride x4", w, CP(4, false, false), 2.0 * bytes);
        run("copy  chunk x4", w, CP(4, true, false), 2.0 * bytes);
        run("copy  chunk x4 nt", w, CP(4, true, true), 2.0 * bytes);
        run("copy  grid-stride x4 nt", w, CP(4, false, true), 2.0 * bytes);
    }
    return 0;
}
