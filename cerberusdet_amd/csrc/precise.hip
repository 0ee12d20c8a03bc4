// Support kernels of the fp32-ACCURATE eval path (cerberusdet_amd/precise.py; round 5).
//
// The reference evaluates a model whose parameters are fp32 in fp32 (cerberusdet/models/cerberus.py:804-882 on a `.float()` model); BASELINE.json asks
// for boxes within 1e-3 relative of it. The engine's 16-bit plans are a statistical match, so the library also carries a path at
// the reference's own precision that still runs every contraction on the MFMA kernels of the product: an fp32 map t is kept as THREE bf16 terms
// t = hi + mid + lo (8 + 8 + 8 mantissa bits, exact up to 2^-24 |t|), likewise every weight, and a convolution is the six term pairs above 2^-24
// accumulated in fp32 by conv_halo_kernel / conv_vt_kernel (fp32-destination epilogue, accumulate form). What sits between two convolutions --
// folded BatchNorm, SiLU, the Bottleneck shortcut, max-pool, nearest upsample, Concat -- is fp32 arithmetic in the kernels below, each of which writes
// the fp32 value AND its three terms, so the next convolution starts from operands that are already split.
//
// All maps are NHWC [N, H, W, ld] with a channel slice [coff, coff + C); the three term buffers of a map share its geometry.
#include "common.h"

namespace cdet {

__device__ __forceinline__ void split3(float t, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    hi = f32_to_bf16_bits(t);
    const float r = t - bf16_bits_to_f32(hi);   // exact (Sterbenz / 8-bit head removed from a 24-bit significand)
    mid = f32_to_bf16_bits(r);
    lo = f32_to_bf16_bits(r - bf16_bits_to_f32(mid));
}

__device__ __forceinline__ void emit(float v, int64_t off, float* y, uint16_t* hi, uint16_t* mid, uint16_t* lo) {
    if (y != nullptr) y[off] = v;
    if (hi != nullptr) {
        uint16_t a, b, c;
        split3(v, a, b, c);
        hi[off] = a;
        mid[off] = b;
        lo[off] = c;
    }
}

// dst pixel (n, y, x), channel c  <-  src pixel (n, y >> up, x >> up) (up = 1: nearest 2x upsample of an H/2 x W/2 source), NHWC or NCHW source
__global__ void split3_kernel(const void* __restrict__ src, int src_dtype, int src_ld, int src_coff, int nchw, int up, float* __restrict__ f32,
                              uint16_t* __restrict__ hi, uint16_t* __restrict__ mid, uint16_t* __restrict__ lo, int dst_ld, int dst_coff, int N, int H,
                              int W, int C) {
    const int64_t total = (int64_t)N * H * W * C;
    const int Hs = H >> up, Ws = W >> up;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int x = (int)(p % W);
        const int64_t q = p / W;
        const int y = (int)(q % H), n = (int)(q / H);
        const int ys = y >> up, xs = x >> up;
        const int64_t so = nchw ? (((int64_t)n * C + c) * Hs + ys) * Ws + xs : (((int64_t)n * Hs + ys) * Ws + xs) * src_ld + src_coff + c;
        emit(load_elem(src, so, src_dtype), p * dst_ld + dst_coff + c, f32, hi, mid, lo);
    }
}

// y = act(z * scale + bias) + res, all fp32 (scale / bias / res may be null)
__global__ void epilogue_f32_kernel(const float* __restrict__ z, int z_ld, int z_coff, const float* __restrict__ scale, const float* __restrict__ bias,
                                    int act, const float* __restrict__ res, int res_ld, int res_coff, float* __restrict__ y, uint16_t* __restrict__ hi,
                                    uint16_t* __restrict__ mid, uint16_t* __restrict__ lo, int y_ld, int y_coff, int64_t M, int C) {
    const int64_t total = M * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        float v = z[p * z_ld + z_coff + c];
        if (scale != nullptr) v *= scale[c];
        if (bias != nullptr) v += bias[c];
        if (act == CDET_ACT_SILU) v = v / (1.0f + expf(-v));  // (expf, not the fast intrinsic of the 16-bit plans)
        if (res != nullptr) v += res[p * res_ld + res_coff + c];
        emit(v, p * y_ld + y_coff + c, y, hi, mid, lo);
    }
}

// k x k max-pool, stride 1, padding k / 2 (SPPF, reference models/common.py:174-191): out-of-image taps do not take part
__global__ void pool_f32_kernel(const float* __restrict__ x, int x_ld, int x_coff, float* __restrict__ y, uint16_t* __restrict__ hi,
                                uint16_t* __restrict__ mid, uint16_t* __restrict__ lo, int y_ld, int y_coff, int N, int H, int W, int C, int k) {
    const int64_t total = (int64_t)N * H * W * C;
    const int r = k / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int px = (int)(p % W);
        const int64_t q = p / W;
        const int py = (int)(q % H), n = (int)(q / H);
        float m = -INFINITY;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = py + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = px + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                m = fmaxf(m, x[(((int64_t)n * H + yy) * W + xx) * x_ld + x_coff + c]);
            }
        }
        emit(m, p * y_ld + y_coff + c, y, hi, mid, lo);
    }
}

// Train-form BatchNorm of an fp32 map (nn.BatchNorm2d in train mode, reference models/common.py:57-62): batch statistics in double, two launches with a
// fixed summation order -- partial sums of BNS_NB row groups, then one thread per channel adds them ascending and writes the folded scale / bias the
// epilogue applies (scale = gamma / sqrt(var + eps), bias = beta - mean * scale) and the running statistics (momentum form, unbiased variance).
constexpr int BNS_NB = 128;

__global__ void bn_stats_f32_partial_kernel(const float* __restrict__ z, int z_ld, int z_coff, int64_t M, int C, double* __restrict__ ws) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (c < C) {
        for (int64_t p = (int64_t)blockIdx.y * 4 + r; p < M; p += (int64_t)BNS_NB * 4) {
            const double v = (double)z[p * z_ld + z_coff + c];
            s += v;
            q += v * v;
        }
    }
    red[0][r][threadIdx.x & 63] = s;
    red[1][r][threadIdx.x & 63] = q;
    __syncthreads();
    if (r == 0 && c < C) {
        const int l = threadIdx.x & 63;
        ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = ((red[0][0][l] + red[0][1][l]) + red[0][2][l]) + red[0][3][l];
        ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = ((red[1][0][l] + red[1][1][l]) + red[1][2][l]) + red[1][3][l];
    }
}

__global__ void bn_stats_f32_final_kernel(const double* __restrict__ ws, int64_t M, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          double eps, double momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                          float* __restrict__ scale, float* __restrict__ bias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < BNS_NB; ++b) {
        s += ws[((int64_t)b * 2 + 0) * C + c];
        q += ws[((int64_t)b * 2 + 1) * C + c];
    }
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sc = (double)gamma[c] / sqrt(var + eps);
    scale[c] = (float)sc;
    bias[c] = (float)((double)beta[c] - mean * sc);
    if (running_mean != nullptr) {
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

static inline int grid_for(int64_t total) {
    const int64_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_split3(const void* src, int src_dtype, int src_ld, int src_coff, int src_nchw, int upsample, float* dst_f32, void* dst_hi, void* dst_mid,
                           void* dst_lo, int dst_ld, int dst_coff, int N, int H, int W, int C, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(src != nullptr && dst_hi != nullptr && dst_mid != nullptr && dst_lo != nullptr, "cdet_split3: null pointer");
    CDET_CHECK_ARG(src_dtype == CDET_F32 || src_dtype == CDET_BF16 || src_dtype == CDET_F16, "cdet_split3: source dtype %d", src_dtype);
    CDET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && dst_coff >= 0 && dst_coff + C <= dst_ld, "cdet_split3: bad geometry");
    CDET_CHECK_ARG(src_nchw || (src_coff >= 0 && src_coff + C <= src_ld), "cdet_split3: source slice outside its pixel row");
    CDET_CHECK_ARG(!upsample || (H % 2 == 0 && W % 2 == 0), "cdet_split3: an upsampled destination has even sides");
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(split3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, src_dtype, src_ld, src_coff, src_nchw, upsample ? 1 : 0, dst_f32,
                       (uint16_t*)dst_hi, (uint16_t*)dst_mid, (uint16_t*)dst_lo, dst_ld, dst_coff, N, H, W, C);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_epilogue_f32(const float* z, int z_ld, int z_coff, const float* scale, const float* bias, int act, const float* res, int res_ld,
                                 int res_coff, float* y, void* y_hi, void* y_mid, void* y_lo, int y_ld, int y_coff, int64_t M, int C, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(z != nullptr && (y != nullptr || y_hi != nullptr), "cdet_epilogue_f32: null pointer");
    CDET_CHECK_ARG((y_hi == nullptr) == (y_mid == nullptr) && (y_hi == nullptr) == (y_lo == nullptr), "cdet_epilogue_f32: the three term buffers come together");
    CDET_CHECK_ARG(M > 0 && C > 0 && z_coff + C <= z_ld && y_coff + C <= y_ld && (res == nullptr || res_coff + C <= res_ld), "cdet_epilogue_f32: bad geometry");
    CDET_CHECK_ARG(act == CDET_ACT_NONE || act == CDET_ACT_SILU, "cdet_epilogue_f32: activation %d", act);
    hipLaunchKernelGGL(epilogue_f32_kernel, dim3(grid_for(M * C)), dim3(256), 0, stream, z, z_ld, z_coff, scale, bias, act, res, res_ld, res_coff, y,
                       (uint16_t*)y_hi, (uint16_t*)y_mid, (uint16_t*)y_lo, y_ld, y_coff, M, C);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_maxpool_f32(const float* x, int x_ld, int x_coff, float* y, void* y_hi, void* y_mid, void* y_lo, int y_ld, int y_coff, int N, int H,
                                int W, int C, int k, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(x != nullptr && (y != nullptr || y_hi != nullptr), "cdet_maxpool_f32: null pointer");
    CDET_CHECK_ARG((y_hi == nullptr) == (y_mid == nullptr) && (y_hi == nullptr) == (y_lo == nullptr), "cdet_maxpool_f32: the three term buffers come together");
    CDET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k >= 1 && (k & 1) && x_coff + C <= x_ld && y_coff + C <= y_ld, "cdet_maxpool_f32: bad geometry");
    hipLaunchKernelGGL(pool_f32_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, stream, x, x_ld, x_coff, y, (uint16_t*)y_hi, (uint16_t*)y_mid,
                       (uint16_t*)y_lo, y_ld, y_coff, N, H, W, C, k);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t cdet_bn_train_f32_ws_doubles(int C) { return (int64_t)BNS_NB * 2 * C; }

extern "C" int cdet_bn_train_f32(const float* z, int z_ld, int z_coff, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, double* ws, float* scale, float* bias, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(z != nullptr && gamma != nullptr && beta != nullptr && ws != nullptr && scale != nullptr && bias != nullptr, "cdet_bn_train_f32: null pointer");
    CDET_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cdet_bn_train_f32: running mean and variance come together");
    CDET_CHECK_ARG(M > 0 && C > 0 && z_coff >= 0 && z_coff + C <= z_ld, "cdet_bn_train_f32: bad geometry");
    hipLaunchKernelGGL(bn_stats_f32_partial_kernel, dim3((C + 63) / 64, BNS_NB), dim3(256), 0, stream, z, z_ld, z_coff, M, C, ws);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_f32_final_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, ws, M, C, gamma, beta, (double)eps, (double)momentum, running_mean,
                       running_var, scale, bias);
    CDET_LAUNCH_CHECK();
    return 0;
}
