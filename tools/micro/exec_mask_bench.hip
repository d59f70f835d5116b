// Microbenchmark (GPU box): cycles per DEPENDENT VALU instruction of a single wavefront as a function of the EXEC mask
// (all 64 lanes, the low 32, the low 16) -- does CDNA4 skip the 16-lane passes of a wave64 instruction whose lanes are
// all masked off?  The tail kernels evaluate one candidate line (32 lanes) per wavefront: if the answer is yes, their
// serial instruction chains get cheaper by masking the idle half.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/exec_mask_bench.hip -o build_ab/exec_mask_bench && build_ab/exec_mask_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define REP256(X) REP16(REP16(X))

template <int OP>
__global__ __launch_bounds__(64) void k(unsigned long long mask, double *out, unsigned long long *cyc) {
    double x = 1.0 + threadIdx.x, y = 1.000001;
    float xf = 1.0f + threadIdx.x, yf = 1.000001f;
    int xi = threadIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_mov_b64 exec, %0" ::"s"(mask));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (OP == 0) { REP256(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (OP == 1) { REP256(asm volatile("v_add_f32 %0, %0, %1" : "+v"(xf) : "v"(yf));) }
    if (OP == 2) { REP256(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (OP == 3) { REP256(asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(xi));) }
    if (OP == 4) {  // two independent chains interleaved: issue rate rather than latency
        double x2 = 2.0 + threadIdx.x;
        REP256(asm volatile("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2" : "+v"(x), "+v"(x2) : "v"(y));)
        x += x2;
    }
    if (OP == 5) { REP256(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi) : "v"(xi));) }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_mov_b64 exec, -1");
    out[threadIdx.x] = x + xf + xi;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 64 * sizeof(double));
    hipMalloc(&cyc, sizeof(unsigned long long));
    const char *names[] = {"v_add_f64 (dependent)", "v_add_f32 (dependent)", "v_max_f64 (dependent)", "s_nop 1 + v_max_i32_dpp (dependent)",
                           "2 x v_add_f64 (independent pair)", "v_cndmask_b32 (dependent)"};
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x1ull};
#define RUN(OP)                                                                               \
    for (int m = 0; m < 4; ++m) {                                                             \
        unsigned long long best = ~0ull;                                                      \
        for (int r = 0; r < 5; ++r) {                                                         \
            hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, masks[m], out, cyc);           \
            hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);                             \
            if (h < best) best = h;                                                           \
        }                                                                                     \
        printf("%-40s exec=%016llx: %6llu ticks / 256 instr = %.2f per instr\n", names[OP], masks[m], best, best / 256.0); \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
