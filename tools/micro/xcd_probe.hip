// Which CU-mask bits belong to which XCD: launches a kernel of many small workgroups on streams created with
// hipExtStreamCreateWithCUMask and records HW_REG_XCC_ID / HW_ID of every workgroup.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/xcd_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <set>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);          // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // stay a little so that the workgroups spread over the allowed CUs
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
}

static int run(hipStream_t s, const char *label, unsigned *d, int nwg) {
    std::vector<unsigned> h(2 * nwg);
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(64), 0, s, d, 2000);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;
    for (int i = 0; i < nwg; ++i) cus[h[2 * i]].insert((h[2 * i + 1] >> 8) & 0xf | ((h[2 * i + 1] >> 13) & 0x7) << 4 | ((h[2 * i + 1] >> 12) & 1) << 7);
    printf("%-34s:", label);
    for (auto &kv : cus) printf(" xcc%u:%zu", kv.first, kv.second.size());
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs %d\n", p.name, p.multiProcessorCount);
    unsigned *d;
    const int nwg = 4096;
    CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * nwg));
    hipStream_t s0;
    CHECK(hipStreamCreate(&s0));
    if (run(s0, "no mask", d, nwg)) return 1;
    const int words = (p.multiProcessorCount + 31) / 32;
    for (int variant = 0; variant < 3; ++variant)
        for (int k = 0; k < 8; k += (variant == 2 ? 8 : 1)) {
            std::vector<uint32_t> mask(words, 0);
            char label[64];
            if (variant == 0) {  // 32 contiguous bits
                for (int b = 32 * k; b < 32 * k + 32 && b < p.multiProcessorCount; ++b) mask[b / 32] |= 1u << (b % 32);
                snprintf(label, sizeof label, "bits [%d, %d)", 32 * k, 32 * k + 32);
            } else if (variant == 1) {  // every 8th bit
                for (int b = k; b < p.multiProcessorCount; b += 8) mask[b / 32] |= 1u << (b % 32);
                snprintf(label, sizeof label, "bits %d + 8 i", k);
            } else {
                for (int b = 0; b < 64; ++b) mask[b / 32] |= 1u << (b % 32);
                snprintf(label, sizeof label, "bits [0, 64)");
            }
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data());
            if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", label, hipGetErrorString(e)); continue; }
            if (run(s, label, d, nwg)) return 1;
            CHECK(hipStreamDestroy(s));
        }
    return 0;
}
