// VALU issue-rate probe (gfx950): cycles per wave-instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_exp_f16, v_rcp_f16 with 1 / 2 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_valu.hip -o tools/probes/probe_valu ; run: tools/probes/probe_valu
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 1) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 2) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3\n v_rcp_f16 %4, %4\n v_rcp_f16 %5, %5\n v_rcp_f16 %6, %6\n v_rcp_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 5) { REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
static void run(const char* name, float* out, unsigned long long* cyc) {
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {  // 256 threads = one wave per SIMD per workgroup
        const int iters = 200;
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD %d: %6.2f clocks per wave-instruction (%.2f per instruction and SIMD)\n", name, wg_per_cu, (double)c / (iters * 64.0), (double)c / (iters * 64.0) / wg_per_cu);
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_exp_f32", out, cyc);
    run<2>("v_rcp_f32", out, cyc);
    run<3>("v_exp_f16", out, cyc);
    run<4>("v_rcp_f16", out, cyc);
    run<5>("1 v_exp_f32 : 3 v_fma_f32", out, cyc);
    return 0;
}
