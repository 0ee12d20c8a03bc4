// Probe: empirical semantics of ds_read_b64_tr_b16 on gfx950 (which lane supplies which address, which lane gets what).
// LDS holds u16 values = element index. Each lane passes the address of element (lane * 4) [test A] so that
// value v in the result tells "came from lane v/4, element v%4".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + lane * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 probe: lane -> 4 x (src_lane.elem)\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  %2d.%d", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        printf("\n");
    }
    return 0;
}
