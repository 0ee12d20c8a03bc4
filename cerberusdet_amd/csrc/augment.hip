// Training-time augmentation on the GPU: mosaic + random affine + mixup + HSV + flips + BGR->RGB + HWC->CHW in ONE kernel per batch,
// straight from the ORIGINAL decoded uint8 images. Replaces, per sample, the reference's chain of cv2 calls on a
// CPU worker: cv2.resize of four images (data/datasets.py:470-477 load_image), the 2s x 2s paste (483-527 load_mosaic), cv2.warpAffine
// (data/augmentations.py:151 random_perspective), the mixup blend (205-211), cv2.cvtColor / cv2.LUT (43-57 augment_hsv), np.flipud /
// np.fliplr and the final transpose (datasets.py:420-438). All random parameters and the label geometry are produced on the host
// (cerberusdet_amd/augment.py) in the reference's order of draws; this kernel only renders.
//
// One thread = one output pixel (3 channels):
//   1. undo the flips; 2. cv2.warpAffine's INTER_LINEAR for 8-bit images: inverse map in 10-bit fixed point (or cv2.warpPerspective's
//   per-pixel division in double when hyp['perspective'] != 0), coordinates with 5 fractional bits, four taps weighted by (32-fx)(32-fy)*32 ... / 2^15, taps outside the canvas = 114; 3. a tap on the canvas is the pixel of the tile
//   that covers it (else 114), i.e. cv2.resize's INTER_LINEAR sample of the original image (11-bit coefficients, two integer passes; exact
//   2x shrink = area average), evaluated on the fly -- the resized images and the canvas are never materialised; 4. mixup: a second mosaic
//   rendered the same way, (a*r + b*(1-r)) truncated; 5. BGR -> HSV (OpenCV's 8-bit integer form, H in [0,180)), three lookup tables,
//   HSV -> BGR (float32 sector formula, rounded).
// OpenCV's arithmetic is restated from its published algorithms (no cv2 in this image): parity unpinned; oracle/augment.py is the numpy
// restatement this kernel is bit-exact against.
#include "common.h"

namespace cdet {

#pragma clang fp contract(off)  // the float32 HSV->BGR formula is compared bit for bit with numpy: no fused multiply-adds

__device__ __forceinline__ void aug_lin_coef(int d, double scale, int src, int& s, int& a0, int& a1) {
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

// pixel (rx, ry) of cv2.resize(img, (w, h), INTER_LINEAR) without building it
__device__ __forceinline__ void aug_resized_pixel(const cdet_aug_tile& t, int rx, int ry, int v[3]) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(t.img);
    if (t.h == t.h0 && t.w == t.w0) {
        const unsigned char* p = src + (int64_t)ry * t.pitch + rx * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else if (t.w0 == 2 * t.w && t.h0 == 2 * t.h) {
        const unsigned char* p0 = src + (int64_t)(2 * ry) * t.pitch + 2 * rx * 3;
        const unsigned char* p1 = p0 + t.pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
    } else {
        int sx, xa0, xa1, sy, ya0, ya1;
        aug_lin_coef(rx, (double)t.w0 / t.w, t.w0, sx, xa0, xa1);
        aug_lin_coef(ry, (double)t.h0 / t.h, t.h0, sy, ya0, ya1);
        const int sx1 = min(sx + 1, t.w0 - 1), sy1 = min(sy + 1, t.h0 - 1);
        const unsigned char* r0 = src + (int64_t)sy * t.pitch;
        const unsigned char* r1 = src + (int64_t)sy1 * t.pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[sx * 3 + c] * xa0 + r0[sx1 * 3 + c] * xa1;
            const int h1 = r1[sx * 3 + c] * xa0 + r1[sx1 * 3 + c] * xa1;
            const int o = (((ya0 * (h0 >> 4)) >> 16) + ((ya1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v[c] = min(max(o, 0), 255);
        }
    }
}

// pixel (cx, cy) of the virtual canvas -- 2s x 2s for a mosaic, s x s for the single letterboxed image -- 114 where no tile was pasted
// (letterbox's border colour is the same 114) and outside (warpAffine's border value)
__device__ __forceinline__ void aug_canvas_pixel(const cdet_aug_tile* tiles, int s2, int cx, int cy, int v[3]) {
    v[0] = v[1] = v[2] = 114;
    if ((unsigned)cx >= (unsigned)s2 || (unsigned)cy >= (unsigned)s2) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cdet_aug_tile& t = tiles[i];
        if (cx >= t.x1a && cx < t.x2a && cy >= t.y1a && cy < t.y2a) {
            aug_resized_pixel(t, cx - t.x1a + t.x1b, cy - t.y1a + t.y1b, v);
            return;
        }
    }
}

__device__ __forceinline__ int aug_round_sat(double x) {  // saturate_cast<int>(double) = cvRound: nearest, ties to even
    x = rint(x);
    return x >= 2147483647.0 ? 2147483647 : (x <= -2147483648.0 ? (int)-2147483648ll : (int)x);
}

// remapBilinear (INTER_LINEAR, BORDER_CONSTANT 114) at the fixed-point coordinate (X, Y): 5 fractional bits each
__device__ __forceinline__ void aug_remap_pixel(const cdet_aug_tile* tiles, int canvas, int X, int Y, int v[3]) {
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767);
    sy = min(max(sy, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int p00[3], p01[3], p10[3], p11[3];
    aug_canvas_pixel(tiles, canvas, sx, sy, p00);
    aug_canvas_pixel(tiles, canvas, sx + 1, sy, p01);
    aug_canvas_pixel(tiles, canvas, sx, sy + 1, p10);
    aug_canvas_pixel(tiles, canvas, sx + 1, sy + 1, p11);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11 + (1 << 14)) >> 15;
}

// cv2.warpAffine(canvas, M, (s, s), borderValue=114) at (x, y): inverse map in 10-bit fixed point
__device__ __forceinline__ void aug_warp_pixel(const cdet_aug_tile* tiles, const double* m, int canvas, int x, int y, int v[3]) {
    const int adelta = aug_round_sat(m[0] * x * 1024.0), bdelta = aug_round_sat(m[3] * x * 1024.0);
    const int X0 = aug_round_sat((m[1] * y + m[2]) * 1024.0) + 16, Y0 = aug_round_sat((m[4] * y + m[5]) * 1024.0) + 16;
    aug_remap_pixel(tiles, canvas, (X0 + adelta) >> 5, (Y0 + bdelta) >> 5, v);
}

// cv2.warpPerspective(canvas, M, (s, s), borderValue=114) at (x, y) (data/augmentations.py:152-153, hyp['perspective'] != 0): OpenCV
// walks the output in blocks 64 wide, evaluates the homogeneous coordinate of the block's first column in double, adds the column
// offset, divides per pixel (W = 32 / w, 0 when w == 0) and rounds to 5 fractional bits -- the summation order is kept
__device__ __forceinline__ void aug_warp_persp_pixel(const cdet_aug_tile* tiles, const double* m, int canvas, int x, int y, int v[3]) {
    const int xb = x & ~63, x1 = x & 63;
    const double X0 = m[0] * xb + m[1] * y + m[2], Y0 = m[3] * xb + m[4] * y + m[5], W0 = m[6] * xb + m[7] * y + m[8];
    double W = W0 + m[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + m[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + m[3] * x1) * W));
    aug_remap_pixel(tiles, canvas, aug_round_sat(fX), aug_round_sat(fY), v);
}

// cv2.cvtColor(BGR2HSV) for 8-bit images: integer form with 12-bit reciprocal tables, H in [0, 180)
__device__ __forceinline__ void aug_bgr2hsv(int b, int g, int r, int& h, int& sat, int& val) {
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int sdiv = v ? aug_round_sat((255 << 12) / (double)v) : 0;
    const int hdiv = diff ? aug_round_sat((180 << 12) / (6.0 * diff)) : 0;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    sat = (diff * sdiv + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    if (hh < 0) hh += 180;
    h = hh;
    val = v;
}

// cv2.cvtColor(HSV2BGR) for 8-bit images: float32 sector formula, saturate_cast<uchar>(x * 255)
__device__ __forceinline__ void aug_hsv2bgr(int hi, int si, int vi, int out[3]) {
    const float s = (float)si * (1.0f / 255.0f), v = (float)vi * (1.0f / 255.0f);
    float b = v, g = v, r = v;
    if (si != 0) {
        float h = (float)hi * (6.0f / 180.0f);
        while (h < 0.f) h += 6.f;
        while (h >= 6.f) h -= 6.f;
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) {
            sector = 0;
            h = 0.f;
        }
        float tab[4];
        tab[0] = v;
        tab[1] = v * (1.f - s);
        tab[2] = v * (1.f - s * h);
        tab[3] = v * (1.f - s * (1.f - h));
        const int sb[6] = {1, 1, 3, 0, 0, 2}, sg[6] = {3, 0, 0, 2, 1, 1}, sr[6] = {0, 2, 1, 1, 3, 0};
        b = tab[sb[sector]];
        g = tab[sg[sector]];
        r = tab[sr[sector]];
    }
    out[0] = min(max((int)rintf(b * 255.f), 0), 255);
    out[1] = min(max((int)rintf(g * 255.f), 0), 255);
    out[2] = min(max((int)rintf(r * 255.f), 0), 255);
}

__global__ __launch_bounds__(256) void mosaic_augment_kernel(const cdet_aug_sample* __restrict__ samples, unsigned char* __restrict__ out, int B, int s) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    if (x >= s) return;
    const cdet_aug_sample& sm = samples[b];
    // output pixel (x, y) = pixel (xs, ys) of the image before np.flipud / np.fliplr
    const int xs = sm.fliplr ? s - 1 - x : x, ys = sm.flipud ? s - 1 - y : y;
    int v[3];
    if (sm.perspective) aug_warp_persp_pixel(sm.tiles, sm.minv, sm.canvas, xs, ys, v);
    else aug_warp_pixel(sm.tiles, sm.minv, sm.canvas, xs, ys, v);
    if (sm.n_mosaic > 1) {  // mixup: (im * r + im2 * (1 - r)).astype(np.uint8) in float64
        int v2[3];
        if (sm.perspective) aug_warp_persp_pixel(sm.tiles + 4, sm.minv + 9, sm.canvas, xs, ys, v2);
        else aug_warp_pixel(sm.tiles + 4, sm.minv + 9, sm.canvas, xs, ys, v2);
        const double r = sm.mix_ratio;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (int)((double)v[c] * r + (double)v2[c] * (1.0 - r));
    }
    if (sm.use_hsv) {
        int h, sa, va;
        aug_bgr2hsv(v[0], v[1], v[2], h, sa, va);
        aug_hsv2bgr(sm.lut[h], sm.lut[256 + sa], sm.lut[512 + va], v);
    }
    const int64_t plane = (int64_t)s * s;
    const int64_t o = (int64_t)b * 3 * plane + (int64_t)y * s + x;
    out[o] = (unsigned char)v[2];  // RGB planes from BGR
    out[o + plane] = (unsigned char)v[1];
    out[o + 2 * plane] = (unsigned char)v[0];
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_mosaic_augment_batch(const cdet_aug_sample* samples, int32_t B, void* out_u8_nchw, int32_t s, void* stream) {
    CDET_CHECK_ARG(samples && out_u8_nchw && B > 0 && s > 0, "cdet_mosaic_augment_batch: bad arguments");
    CDET_CHECK_ARG(B <= 65535 && s <= 16384, "cdet_mosaic_augment_batch: batch / size exceed the grid and fixed-point limits");
    dim3 grid(div_up(s, 256), s, B);
    hipLaunchKernelGGL(mosaic_augment_kernel, grid, dim3(256), 0, (hipStream_t)stream, samples, (unsigned char*)out_u8_nchw, B, s);
    CDET_LAUNCH_CHECK();
    return 0;
}
