// The Detect head's eval-mode decode (DFL softmax-expectation + dist2bbox + sigmoid, writes the reference's [N,4+nc,A]) and the fold of
// a channel-padded weight gradient (the stem's forward lives in stem_mfma.hip, its weight gradient in stem_wgrad.hip).
#include "common.h"

namespace cdet {

// ------------------------------------------------------------------------------------------------
// Detect decode: one thread per (image, anchor)
// ------------------------------------------------------------------------------------------------
struct DecodeArgs {
    const void* f[3];
    int h[3], w[3];
    float stride[3];
    int a_off[4];  // anchor offsets of the levels, a_off[3] = A
};

__global__ __launch_bounds__(256) void detect_decode_kernel(DecodeArgs d, int N, int nc, int f_ld, int dtype, void* __restrict__ y, int out_dtype) {
    const int A = d.a_off[3];
    const int64_t total = (int64_t)N * A;
    const int no = f_ld;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / A);
        const int a = (int)(idx - (int64_t)n * A);
        const int lvl = a >= d.a_off[2] ? 2 : (a >= d.a_off[1] ? 1 : 0);
        const int la = a - d.a_off[lvl];
        const int gx = la % d.w[lvl], gy = la / d.w[lvl];
        const void* f = d.f[lvl];
        const int64_t base = ((int64_t)n * d.h[lvl] * d.w[lvl] + la) * no;
        float dist[4];
        const bool vec = dtype == CDET_F32 && (no & 3) == 0;  // the engine's head maps: fp32 rows of 64 + pad8(nc) floats -> 16-byte loads
        const f32x4* row4 = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(f) + base);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[16];
            float mx = -INFINITY;
            if (vec) {
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const f32x4 q = row4[s * 4 + b4];
                    v[b4 * 4 + 0] = q[0]; v[b4 * 4 + 1] = q[1]; v[b4 * 4 + 2] = q[2]; v[b4 * 4 + 3] = q[3];
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) mx = fmaxf(mx, v[b]);
            } else {
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    v[b] = load_elem(f, base + s * 16 + b, dtype);
                    mx = fmaxf(mx, v[b]);
                }
            }
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const float e = expf(v[b] - mx);
                den += e;
                num += e * (float)b;
            }
            dist[s] = num / den;
        }
        const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f, st = d.stride[lvl];
        const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
        const int64_t yo = (int64_t)n * (4 + nc) * A + a;
        store_elem(y, yo + 0 * (int64_t)A, (x1 + x2) * 0.5f * st, out_dtype);
        store_elem(y, yo + 1 * (int64_t)A, (y1 + y2) * 0.5f * st, out_dtype);
        store_elem(y, yo + 2 * (int64_t)A, (x2 - x1) * st, out_dtype);
        store_elem(y, yo + 3 * (int64_t)A, (y2 - y1) * st, out_dtype);
        if (vec) {
            for (int c4 = 0; c4 * 4 < nc; ++c4) {
                const f32x4 q = row4[16 + c4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c4 * 4 + i < nc) store_elem(y, yo + (int64_t)(4 + c4 * 4 + i) * A, 1.0f / (1.0f + expf(-q[i])), out_dtype);
            }
        } else {
            for (int c = 0; c < nc; ++c) {
                const float z = load_elem(f, base + 64 + c, dtype);
                store_elem(y, yo + (int64_t)(4 + c) * A, 1.0f / (1.0f + expf(-z)), out_dtype);
            }
        }
    }
}

// Round 5: FOUR lanes per (image, anchor) for the engine's head maps (fp32 rows of 64 + pad8(nc) floats). Lane s of a group owns side s of the box: its
// 16 DFL bins are one contiguous 64-byte run (four lanes = 256 contiguous bytes of the row, where the one-thread form had every lane 576 bytes from its
// neighbour), the softmax expectation keeps the per-side order of the one-thread form (same bits), the four distances meet through three DPP reads of the
// quad, every lane stores ONE of cx / cy / w / h and every fourth class. A quarter of the dependent chain per thread, four times the threads: the decode is
// the tail of the eval forward (48 -> see profiles/r05_fwd_timeline.txt).
__global__ __launch_bounds__(256) void detect_decode4_kernel(DecodeArgs d, int N, int nc, int f_ld, float* __restrict__ y) {
    const int A = d.a_off[3];
    const int64_t total = (int64_t)N * A * 4;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (whole quads stay together: the grid covers `total` rounded up to a multiple of 4 by construction, and out-of-range quads do their reads on anchor 0)
    for (int64_t t = first; t < (total + 255) / 256 * 256; t += (int64_t)gridDim.x * blockDim.x) {
        const bool live = t < total;
        const int64_t idx = live ? t >> 2 : 0;
        const int s = (int)(t & 3);
        const int n = (int)(idx / A);
        const int a = (int)(idx - (int64_t)n * A);
        const int lvl = a >= d.a_off[2] ? 2 : (a >= d.a_off[1] ? 1 : 0);
        const int la = a - d.a_off[lvl];
        const int gx = la % d.w[lvl], gy = la / d.w[lvl];
        const float* row = reinterpret_cast<const float*>(d.f[lvl]) + ((int64_t)n * d.h[lvl] * d.w[lvl] + la) * f_ld;
        const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
            const f32x4 q = row4[s * 4 + b4];
            v[b4 * 4 + 0] = q[0]; v[b4 * 4 + 1] = q[1]; v[b4 * 4 + 2] = q[2]; v[b4 * 4 + 3] = q[3];
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) mx = fmaxf(mx, v[b]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const float e = expf(v[b] - mx);
            den += e;
            num += e * (float)b;
        }
        const float dist = num / den;
        // the quad's four distances (quad_perm broadcasts of lanes 0..3 of the quad)
        const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dist), 0x00, 0xf, 0xf, true));
        const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dist), 0x55, 0xf, 0xf, true));
        const float d2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dist), 0xAA, 0xf, 0xf, true));
        const float d3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dist), 0xFF, 0xf, 0xf, true));
        const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f, st = d.stride[lvl];
        const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
        const float box = s == 0 ? (x1 + x2) * 0.5f * st : (s == 1 ? (y1 + y2) * 0.5f * st : (s == 2 ? (x2 - x1) * st : (y2 - y1) * st));
        if (live) {
            const int64_t yo = (int64_t)n * (4 + nc) * A + a;
            y[yo + (int64_t)s * A] = box;
            for (int c = s; c < nc; c += 4) y[yo + (int64_t)(4 + c) * A] = 1.0f / (1.0f + expf(-row[64 + c]));
        }
    }
}

}  // namespace cdet

using namespace cdet;

__global__ void fold_padded_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int Ip, int I, int taps, int accumulate) {
    const int n = O * I * taps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i / (I * taps), r = i - o * (I * taps);
        const float v = src[(int64_t)o * Ip * taps + r];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

extern "C" int cdet_fold_padded_wgrad(const float* dw_pad, float* dw, int32_t O, int32_t I_pad, int32_t I_real, int32_t taps, int32_t accumulate,
                                      void* stream) {
    CDET_CHECK_ARG(dw_pad && dw && O > 0 && I_real > 0 && I_real <= I_pad && taps > 0, "cdet_fold_padded_wgrad: bad arguments");
    const int n = O * I_real * taps;
    hipLaunchKernelGGL(fold_padded_wgrad_kernel, dim3(div_up(n, 256) < 64 ? div_up(n, 256) : 64), dim3(256), 0, (hipStream_t)stream, dw_pad, dw, O, I_pad,
                       I_real, taps, accumulate);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_detect_decode(const void* f0, const void* f1, const void* f2, const int32_t* hw6, const float* strides3, int32_t N, int32_t nc,
                                  int32_t f_ld, int32_t dtype, void* y, int32_t out_dtype, void* stream) {
    CDET_CHECK_ARG(f0 && f1 && f2 && hw6 && strides3 && y, "cdet_detect_decode: null pointer");
    CDET_CHECK_ARG(f_ld >= 64 + nc, "cdet_detect_decode: f_ld (%d) < 64 + nc (%d)", f_ld, 64 + nc);
    DecodeArgs d;
    d.f[0] = f0; d.f[1] = f1; d.f[2] = f2;
    int off = 0;
    for (int i = 0; i < 3; ++i) {
        d.h[i] = hw6[2 * i]; d.w[i] = hw6[2 * i + 1]; d.stride[i] = strides3[i];
        d.a_off[i] = off;
        off += d.h[i] * d.w[i];
    }
    d.a_off[3] = off;
    const int64_t total = (int64_t)N * off;
    if (dtype == CDET_F32 && out_dtype == CDET_F32 && (f_ld & 3) == 0 && tune_env("CDET_DECODE4", 1)) {  // the engine's head maps: four lanes per anchor
        const int64_t b4 = (total * 4 + 255) / 256;
        hipLaunchKernelGGL(detect_decode4_kernel, dim3((unsigned)(b4 > 16384 ? 16384 : b4)), dim3(256), 0, (hipStream_t)stream, d, N, nc, f_ld, (float*)y);
        CDET_LAUNCH_CHECK();
        return 0;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(detect_decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d, N, nc, f_ld, dtype, y, out_dtype);
    CDET_LAUNCH_CHECK();
    return 0;
}
