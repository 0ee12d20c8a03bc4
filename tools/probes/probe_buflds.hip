// Probe: buffer_load_dwordx4 ... lds on gfx950 -- LDS image layout (lane-linear 16 B?) and out-of-range behaviour (zeros written?).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint32_t* x, int bytes, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
    int voff = (63 - threadIdx.x) * 16;            // reversed gather so layout is visible
    if ((threadIdx.x & 7) == 3) voff = 0x7ffffff0;  // out of range
    if (threadIdx.x == 5) voff = bytes - 8;         // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 256), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    uint32_t *d, *o;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, 1024 /*bytes visible*/, o);
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    for (int l = 0; l < 72; ++l) {
        printf("slot %2d (byte %4d):", l, l * 16);
        for (int j = 0; j < 4; ++j) printf(" %08x", r[l * 4 + j]);
        printf("\n");
    }
    return 0;
}
