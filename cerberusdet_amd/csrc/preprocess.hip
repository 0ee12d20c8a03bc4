// Inference pre-processing on the GPU: letterbox (bilinear resize + 114-gray border) + BGR->RGB + HWC->CHW + /255 in ONE kernel,
// replacing the per-image cv2.resize / cv2.copyMakeBorder / numpy transpose / H2D of the float tensor of the reference
// (cerberusdet/cerberusdet_preprocessor.py:42-74, data/augmentations.py:59-89). The host computes the letterbox geometry
// (integers, augmentations.py:65-86) and ships the raw uint8 images; one thread produces one output pixel (3 channels).
// Resize arithmetic = OpenCV's 8-bit INTER_LINEAR (half-pixel centres, 11-bit fixed-point coefficients, two integer passes;
// exact 2x shrink = 2x2 area average) so that the uint8 image equals cv2's bit for bit -- restated from OpenCV's published
// algorithm, parity unpinned (oracle/preprocess.py). HBM-bound: reads every source pixel ~once, writes 3 x 2-4 B per output pixel.
#include "common.h"

namespace cdet {

__device__ __forceinline__ void lin_coef(int d, double scale, int src, int& s, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);  // OpenCV evaluates the source coordinate in double and keeps it as a float
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
        s = 0;
        f = 0.f;
    }
    if (s >= src - 1) {
        s = src - 1;
        f = 0.f;
    }
    a0 = (int)rintf((1.0f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
}

#pragma clang fp contract(off)  // the float32 accumulation of the area resize is compared bit for bit with numpy

// OpenCV computeResizeAreaTab for one destination index: first covered source cell, number of cells, and the weight of cell k
struct AreaSpan {
    int s0, n;
    float a_first, a_mid, a_last;
    bool has_first, has_last;
    int mid0, mid1;
};
__device__ __forceinline__ AreaSpan area_span(int d, int src, int dst) {
    const double scale = (double)src / dst;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, src - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, src - 1);
    s1 = min(s1, s2);
    AreaSpan sp;
    sp.has_first = s1 - f1 > 1e-3;
    sp.has_last = f2 - s2 > 1e-3;
    sp.a_first = (float)((s1 - f1) / cell);
    sp.a_mid = (float)(1.0 / cell);
    sp.a_last = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
    sp.mid0 = s1;
    sp.mid1 = s2;
    return sp;
}

// pixel (rx, ry) of cv2.resize(img, (new_w, new_h), INTER_AREA), both directions shrinking
__device__ __forceinline__ void area_pixel(const unsigned char* src, int pitch, int h, int w, int new_h, int new_w, int rx, int ry, int v[3]) {
    const double sx = (double)w / new_w, sy = (double)h / new_h;
    const int isx = (int)rint(sx), isy = (int)rint(sy);
    if (fabs(sx - isx) < 2.220446049250313e-16 && fabs(sy - isy) < 2.220446049250313e-16) {  // integer factors: block average
        int sum[3] = {0, 0, 0};
        for (int yy = 0; yy < isy; ++yy) {
            const unsigned char* p = src + (int64_t)(ry * isy + yy) * pitch + rx * isx * 3;
            for (int xx = 0; xx < isx; ++xx)
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] += p[xx * 3 + c];
        }
        if (isx == 2 && isy == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (sum[c] + 2) >> 2;
        } else {
            const float sc = (float)(1.0 / (isx * isy));
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = min(max((int)rintf((float)sum[c] * sc), 0), 255);
        }
        return;
    }
    const AreaSpan X = area_span(rx, w, new_w), Y = area_span(ry, h, new_h);
    float out[3] = {0.f, 0.f, 0.f};
    bool first_row = true;
    auto row = [&](int yy, float beta) {
        const unsigned char* p = src + (int64_t)yy * pitch;
        float acc[3] = {0.f, 0.f, 0.f};
        if (X.has_first)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + (float)p[(X.mid0 - 1) * 3 + c] * X.a_first;
        for (int xx = X.mid0; xx < X.mid1; ++xx)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + (float)p[xx * 3 + c] * X.a_mid;
        if (X.has_last)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + (float)p[X.mid1 * 3 + c] * X.a_last;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = first_row ? acc[c] * beta : out[c] + acc[c] * beta;
        first_row = false;
    };
    if (Y.has_first) row(Y.mid0 - 1, Y.a_first);
    for (int yy = Y.mid0; yy < Y.mid1; ++yy) row(yy, Y.a_mid);
    if (Y.has_last) row(Y.mid1, Y.a_last);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = min(max((int)rintf(out[c]), 0), 255);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const cdet_letterbox_item* __restrict__ items, void* __restrict__ out, int B, int H, int W,
                                                        int out_dtype, int pad_value) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    if (x >= W) return;
    const cdet_letterbox_item it = items[b];
    int v[3] = {pad_value, pad_value, pad_value};  // B, G, R
    const int ry = y - it.top, rx = x - it.left;
    if (ry >= 0 && ry < it.new_h && rx >= 0 && rx < it.new_w) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(it.img);
        if (it.new_h == it.h && it.new_w == it.w) {
            const unsigned char* p = src + (int64_t)ry * it.pitch + rx * 3;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else if (it.area && it.new_w <= it.w && it.new_h <= it.h) {
            area_pixel(src, it.pitch, it.h, it.w, it.new_h, it.new_w, rx, ry, v);
        } else if (it.w == 2 * it.new_w && it.h == 2 * it.new_h) {  // cv::resize: exact 2x shrink -> INTER_AREA
            const unsigned char* p0 = src + (int64_t)(2 * ry) * it.pitch + 2 * rx * 3;
            const unsigned char* p1 = p0 + it.pitch;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
        } else {
            int sx, xa0, xa1, sy, ya0, ya1;
            lin_coef(rx, (double)it.w / it.new_w, it.w, sx, xa0, xa1);
            lin_coef(ry, (double)it.h / it.new_h, it.h, sy, ya0, ya1);
            const int sx1 = min(sx + 1, it.w - 1), sy1 = min(sy + 1, it.h - 1);
            const unsigned char* r0 = src + (int64_t)sy * it.pitch;
            const unsigned char* r1 = src + (int64_t)sy1 * it.pitch;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[sx * 3 + c] * xa0 + r0[sx1 * 3 + c] * xa1;
                const int h1 = r1[sx * 3 + c] * xa0 + r1[sx1 * 3 + c] * xa1;
                const int o = (((ya0 * (h0 >> 4)) >> 16) + ((ya1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = min(max(o, 0), 255);
            }
        }
    }
    const int64_t plane = (int64_t)H * W;
    const int64_t o = ((int64_t)b * 3) * plane + (int64_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // output channel c = RGB[c] = BGR[2 - c]
        const int u = v[2 - c];
        if (out_dtype == CDET_U8) reinterpret_cast<unsigned char*>(out)[o + c * plane] = (unsigned char)u;
        else store_elem(out, o + c * plane, out_dtype == CDET_F16 ? (float)(_Float16)((float)u / 255.0f) : (float)u / 255.0f, out_dtype);
    }
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_letterbox_batch(const cdet_letterbox_item* items, int32_t B, void* out_nchw, int32_t H, int32_t W, int32_t out_dtype,
                                    int32_t pad_value, void* stream) {
    CDET_CHECK_ARG(items && out_nchw && B > 0 && H > 0 && W > 0, "cdet_letterbox_batch: bad arguments");
    CDET_CHECK_ARG(out_dtype == CDET_F32 || out_dtype == CDET_F16 || out_dtype == CDET_BF16 || out_dtype == CDET_U8, "cdet_letterbox_batch: bad out dtype");
    CDET_CHECK_ARG(B <= 65535 && H <= 65535, "cdet_letterbox_batch: batch / height exceed the grid limits");
    dim3 grid(div_up(W, 256), H, B);
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, (hipStream_t)stream, items, out_nchw, B, H, W, out_dtype, pad_value);
    CDET_LAUNCH_CHECK();
    return 0;
}
