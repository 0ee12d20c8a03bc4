// LDS bank behaviour of ds_read_b64_tr_b16 on gfx950: clocks per wave-instruction for a set of per-lane address patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -o tr_bank_probe tr_bank_probe.hip ; run on the GPU box. One workgroup of NW waves on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int KIND>  // 0: tr_b16, 1: plain b64
__global__ __launch_bounds__(512) void probe(const int* addr, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned char* p = smem + addr[lane];
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            u32x2 v;
            if (KIND == 0) {
                s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + u * 4096));
                v = __builtin_bit_cast(u32x2, r);
            } else {
                v = *reinterpret_cast<const u32x2*>(p + u * 4096);
            }
            acc ^= v[0] ^ v[1];
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x12345678u) out[1] = acc;
}

int main() {
    struct Pat { const char* name; int (*f)(int); };
    static const Pat pats[] = {
        {"A  contiguous l*8 (16-lane groups 128 B apart)", [](int l) { return l * 8; }},
        {"B  (l&15)*8 + (l>>4)*512 (guide: 8x[32][16])", [](int l) { return (l & 15) * 8 + (l >> 4) * 512; }},
        {"B2 (l&15)*8 + (q&1)*512 + (q>>1)*2048", [](int l) { int q = l >> 4; return (l & 15) * 8 + (q & 1) * 512 + (q >> 1) * 2048; }},
        {"B3 (l&15)*8 + (q&1)*256 + (q>>1)*2048", [](int l) { int q = l >> 4; return (l & 15) * 8 + (q & 1) * 256 + (q >> 1) * 2048; }},
        {"B4 (l&15)*8 + (q&1)*1024 + (q>>1)*128", [](int l) { int q = l >> 4; return (l & 15) * 8 + (q & 1) * 1024 + (q >> 1) * 128; }},
        {"F  dY now: (q>>1)*3072 + (q&1)*128 + r4*32 + c4*8", [](int l) { int q = l >> 4; return (q >> 1) * 3072 + (q & 1) * 128 + (l & 15) * 8; }},
        {"C  X now: 64-B rows 4q+r4, unit swap on odd q", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 64 + ((q & 1) << 5) + c4 * 8; }},
        {"D  X plain: 64-B rows 4q+r4, no swap", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 64 + c4 * 8; }},
        {"D2 X rows 4q+r4 +1 row shift (odd start)", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4 + 1) * 64 + c4 * 8; }},
        {"D3 X rows 4q+r4 +2 rows", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4 + 2) * 64 + c4 * 8; }},
        {"D4 X rows 4q+r4+3, second unit", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4 + 3) * 64 + 32 + c4 * 8; }},
        {"G  X 32-B rows (one unit per row) 4q+r4", [](int l) { int q = l >> 4; return (l & 15) * 8 + q * 128; }},
        {"H  X 64-B rows, 8 keys of q at rows 8q.. (r4 rows)", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (8 * q + r4) * 64 + c4 * 8; }},
        {"I  X 64-B rows, q at rows 16q", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (16 * q + r4) * 64 + c4 * 8; }},
        {"J  X 64-B rows, q&1 -> rows +8, q>>1 -> rows +4", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (8 * (q & 1) + 4 * (q >> 1) + r4) * 64 + c4 * 8; }},
        {"K  X 64-B rows 4q+r4, unit = r4&1 swap", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 64 + ((r4 & 1) << 5) + c4 * 8; }},
        {"L  all lanes same 128 B (16-lane broadcast)", [](int l) { return (l & 15) * 8; }},
        {"M  X 128-B rows 4q+r4", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 128 + c4 * 8; }},
        {"N  X 96-B rows 4q+r4", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 96 + c4 * 8; }},
        {"O  X 80-B rows 4q+r4", [](int l) { int q = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3; return (4 * q + r4) * 80 + c4 * 8; }},
    };
    int* d_addr; unsigned long long* d_out;
    hipMalloc(&d_addr, 64 * sizeof(int)); hipMalloc(&d_out, 16);
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    const int iters = 2000;
    printf("clocks per wave-instruction (LDS-array bound when all 4 SIMDs issue): [tr_b16 4 waves, 8 waves | plain b64 4 waves, 8 waves]\n");
    for (const Pat& pt : pats) {
        int h[64];
        for (int l = 0; l < 64; ++l) h[l] = pt.f(l);
        hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
        double res[4];
        int k = 0;
        for (int kind = 0; kind < 2; ++kind)
            for (int nw : {4, 8}) {
                unsigned long long c = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(nw * 64), 72 * 1024, 0, d_addr, d_out, iters);
                    else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(nw * 64), 72 * 1024, 0, d_addr, d_out, iters);
                    hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                }
                res[k++] = (double)c / ((double)iters * 16 * nw);
            }
        printf("%-55s  %6.2f %6.2f | %6.2f %6.2f\n", pt.name, res[0], res[1], res[2], res[3]);
    }
    return 0;
}
