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
        // (a uint8 image is scaled like the reference's preprocess_batch: img.float() / 255, one fp32 division)
        const float v = src_dtype == CDET_U8 ? (float)((const uint8_t*)src)[so] / 255.0f : load_elem(src, so, src_dtype);
        emit(v, p * dst_ld + dst_coff + c, f32, hi, mid, lo);
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
                                          float* __restrict__ scale, float* __restrict__ bias, float* __restrict__ mean_out,
                                          float* __restrict__ invstd_out) {
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
    if (mean_out != nullptr) {
        mean_out[c] = (float)mean;
        invstd_out[c] = (float)(1.0 / sqrt(var + eps));
    }
    if (running_mean != nullptr) {
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// ---- backward of y = SiLU(gamma * zhat + beta) (+ res) with zhat = (z - mean) * invstd, z the fp32 accumulator of the convolution (train form):
//   da = dy * SiLU'(a), a = z * scale + bias;  dbeta = sum da;  dgamma = sum da * zhat;  dz = scale * (da - dbeta / M - zhat * dgamma / M)
// three launches: partial sums in double (fixed order), per-channel totals (accumulated into the parameter gradients), the apply pass that writes dz
// as value + three terms -- the operand of the data- and weight-gradient convolutions.
__device__ __forceinline__ float silu_grad_f32(float a) {
    const float sg = 1.0f / (1.0f + expf(-a));
    return sg * (1.0f + a * (1.0f - sg));
}

__global__ void bn_silu_bwd_f32_partial_kernel(const float* __restrict__ dy, int dy_ld, int dy_coff, const float* __restrict__ z, int z_ld,
                                               const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ mean,
                                               const float* __restrict__ invstd, int64_t M, int C, double* __restrict__ ws) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const float sc = scale[c], bi = bias[c], mu = mean[c], is = invstd[c];
        for (int64_t p = (int64_t)blockIdx.y * 4 + r; p < M; p += (int64_t)BNS_NB * 4) {
            const float zv = z[p * z_ld + c];
            const float da = dy[p * dy_ld + dy_coff + c] * silu_grad_f32(zv * sc + bi);
            s += (double)da;
            q += (double)da * (double)((zv - mu) * is);
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

__global__ void bn_silu_bwd_f32_final_kernel(double* __restrict__ ws, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < BNS_NB; ++b) {
        s += ws[((int64_t)b * 2 + 0) * C + c];
        q += ws[((int64_t)b * 2 + 1) * C + c];
    }
    ws[c] = s;          // (row 0 of the scratch now holds the totals the apply pass reads: every partial of column c has been consumed by this thread)
    ws[C + c] = q;
    if (dbeta != nullptr) dbeta[c] += (float)s;
    if (dgamma != nullptr) dgamma[c] += (float)q;
}

__global__ void bn_silu_bwd_f32_apply_kernel(const float* __restrict__ dy, int dy_ld, int dy_coff, const float* __restrict__ z, int z_ld,
                                             const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ mean,
                                             const float* __restrict__ invstd, const double* __restrict__ tot, int64_t M, int C, float* __restrict__ dz,
                                             uint16_t* __restrict__ hi, uint16_t* __restrict__ mid, uint16_t* __restrict__ lo, int dz_ld) {
    const int64_t total = M * C;
    const double inv_m = 1.0 / (double)M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const float zv = z[p * z_ld + c], sc = scale[c];
        const float da = dy[p * dy_ld + dy_coff + c] * silu_grad_f32(zv * sc + bias[c]);
        const float zh = (zv - mean[c]) * invstd[c];
        const float v = sc * (float)((double)da - tot[c] * inv_m - (double)zh * tot[C + c] * inv_m);
        emit(v, p * dz_ld + c, dz, hi, mid, lo);
    }
}

// out[c] += sum over the M pixels of src[:, coff + c] (double, fixed order): the bias gradient of the head's projections
__global__ void colsum_f32_partial_kernel(const float* __restrict__ src, int ld, int coff, int64_t M, int C, double* __restrict__ ws) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    double s = 0.0;
    if (c < C)
        for (int64_t p = (int64_t)blockIdx.y * 4 + r; p < M; p += (int64_t)BNS_NB * 4) s += (double)src[p * ld + coff + c];
    red[r][threadIdx.x & 63] = s;
    __syncthreads();
    if (r == 0 && c < C) {
        const int l = threadIdx.x & 63;
        ws[(int64_t)blockIdx.y * C + c] = ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
    }
}
__global__ void colsum_f32_final_kernel(const double* __restrict__ ws, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < BNS_NB; ++b) s += ws[(int64_t)b * C + c];
    out[c] += (float)s;
}

// dst (+)= src, optionally through the 2x2 down-sum (the backward of the nearest 2x upsample): shortcut / Concat / Upsample gradients
__global__ void add_f32_kernel(const float* __restrict__ src, int src_ld, int src_coff, float* __restrict__ dst, int dst_ld, int dst_coff, int N, int H, int W,
                               int C, int down, int accumulate) {
    const int64_t total = (int64_t)N * H * W * C;  // destination geometry
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        float v;
        if (down) {
            const int x = (int)(p % W);
            const int64_t q = p / W;
            const int y = (int)(q % H), n = (int)(q / H);
            const int64_t b = (((int64_t)n * 2 * H + 2 * y) * 2 * W + 2 * x) * src_ld + src_coff + c;
            v = (src[b] + src[b + src_ld]) + (src[b + (int64_t)2 * W * src_ld] + src[b + (int64_t)2 * W * src_ld + src_ld]);
        } else {
            v = src[p * src_ld + src_coff + c];
        }
        float* d = dst + p * dst_ld + dst_coff + c;
        *d = accumulate ? *d + v : v;
    }
}

// backward of the k x k stride-1 max-pool: dx[n, y, x, c] += sum of dy over the windows whose FIRST maximum (row-major scan, as nn.MaxPool2d) sits at (y, x)
__global__ void pool_bwd_f32_kernel(const float* __restrict__ x, int x_ld, int x_coff, const float* __restrict__ dy, int dy_ld, int dy_coff,
                                    float* __restrict__ dx, int dx_ld, int dx_coff, int N, int H, int W, int C, int k) {
    const int64_t total = (int64_t)N * H * W * C;
    const int r = k / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int px = (int)(p % W);
        const int64_t q = p / W;
        const int py = (int)(q % H), n = (int)(q / H);
        const float* xb = x + (int64_t)n * H * W * x_ld + x_coff + c;
        const float mine = xb[((int64_t)py * W + px) * x_ld];
        float g = 0.f;
        for (int oy = py - r; oy <= py + r; ++oy) {        // output pixels whose window holds (py, px)
            if ((unsigned)oy >= (unsigned)H) continue;
            for (int ox = px - r; ox <= px + r; ++ox) {
                if ((unsigned)ox >= (unsigned)W) continue;
                bool first_max = true;                      // is (py, px) the first maximum of window (oy, ox)?
                for (int wy = oy - r; wy <= oy + r && first_max; ++wy) {
                    if ((unsigned)wy >= (unsigned)H) continue;
                    for (int wx = ox - r; wx <= ox + r; ++wx) {
                        if ((unsigned)wx >= (unsigned)W) continue;
                        const float v = xb[((int64_t)wy * W + wx) * x_ld];
                        const bool before = wy < py || (wy == py && wx < px);
                        if (before ? v >= mine : v > mine) {
                            first_max = false;
                            break;
                        }
                    }
                }
                if (first_max) g += dy[(((int64_t)n * H + oy) * W + ox) * dy_ld + dy_coff + c];
            }
        }
        dx[p * dx_ld + dx_coff + c] += g;
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
    CDET_CHECK_ARG(src_dtype == CDET_F32 || src_dtype == CDET_BF16 || src_dtype == CDET_F16 || src_dtype == CDET_U8, "cdet_split3: source dtype %d", src_dtype);
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
                                float* running_mean, float* running_var, double* ws, float* scale, float* bias, float* mean, float* invstd, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(z != nullptr && gamma != nullptr && beta != nullptr && ws != nullptr && scale != nullptr && bias != nullptr, "cdet_bn_train_f32: null pointer");
    CDET_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cdet_bn_train_f32: running mean and variance come together");
    CDET_CHECK_ARG((mean == nullptr) == (invstd == nullptr), "cdet_bn_train_f32: mean and invstd come together");
    CDET_CHECK_ARG(M > 0 && C > 0 && z_coff >= 0 && z_coff + C <= z_ld, "cdet_bn_train_f32: bad geometry");
    hipLaunchKernelGGL(bn_stats_f32_partial_kernel, dim3((C + 63) / 64, BNS_NB), dim3(256), 0, stream, z, z_ld, z_coff, M, C, ws);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_f32_final_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, ws, M, C, gamma, beta, (double)eps, (double)momentum, running_mean,
                       running_var, scale, bias, mean, invstd);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_bn_silu_bwd_f32(const float* dy, int dy_ld, int dy_coff, const float* z, int z_ld, const float* scale, const float* bias, const float* mean,
                                    const float* invstd, int64_t M, int C, double* ws, float* dgamma, float* dbeta, float* dz, void* dz_hi, void* dz_mid,
                                    void* dz_lo, int dz_ld, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(dy != nullptr && z != nullptr && scale != nullptr && bias != nullptr && mean != nullptr && invstd != nullptr && ws != nullptr,
                   "cdet_bn_silu_bwd_f32: null pointer");
    CDET_CHECK_ARG(dz != nullptr || dz_hi != nullptr, "cdet_bn_silu_bwd_f32: no destination");
    CDET_CHECK_ARG((dz_hi == nullptr) == (dz_mid == nullptr) && (dz_hi == nullptr) == (dz_lo == nullptr), "cdet_bn_silu_bwd_f32: the three term buffers come together");
    CDET_CHECK_ARG(M > 0 && C > 0 && dy_coff >= 0 && dy_coff + C <= dy_ld && C <= z_ld && C <= dz_ld, "cdet_bn_silu_bwd_f32: bad geometry");
    hipLaunchKernelGGL(bn_silu_bwd_f32_partial_kernel, dim3((C + 63) / 64, BNS_NB), dim3(256), 0, stream, dy, dy_ld, dy_coff, z, z_ld, scale, bias, mean, invstd,
                       M, C, ws);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_silu_bwd_f32_final_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, ws, C, dgamma, dbeta);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_silu_bwd_f32_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, stream, dy, dy_ld, dy_coff, z, z_ld, scale, bias, mean, invstd, ws, M, C,
                       dz, (uint16_t*)dz_hi, (uint16_t*)dz_mid, (uint16_t*)dz_lo, dz_ld);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_add_f32(const float* src, int src_ld, int src_coff, float* dst, int dst_ld, int dst_coff, int N, int H, int W, int C, int downsum,
                            int accumulate, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(src != nullptr && dst != nullptr, "cdet_add_f32: null pointer");
    CDET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && src_coff >= 0 && src_coff + C <= src_ld && dst_coff >= 0 && dst_coff + C <= dst_ld, "cdet_add_f32: bad geometry");
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, stream, src, src_ld, src_coff, dst, dst_ld, dst_coff, N, H, W, C,
                       downsum ? 1 : 0, accumulate ? 1 : 0);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_maxpool_bwd_f32(const float* x, int x_ld, int x_coff, const float* dy, int dy_ld, int dy_coff, float* dx, int dx_ld, int dx_coff, int N,
                                    int H, int W, int C, int k, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(x != nullptr && dy != nullptr && dx != nullptr, "cdet_maxpool_bwd_f32: null pointer");
    CDET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k >= 1 && (k & 1) && x_coff + C <= x_ld && dy_coff + C <= dy_ld && dx_coff + C <= dx_ld,
                   "cdet_maxpool_bwd_f32: bad geometry");
    hipLaunchKernelGGL(pool_bwd_f32_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, stream, x, x_ld, x_coff, dy, dy_ld, dy_coff, dx, dx_ld, dx_coff,
                       N, H, W, C, k);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_colsum_f32(const float* src, int ld, int coff, int64_t M, int C, double* ws, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CDET_CHECK_ARG(src != nullptr && ws != nullptr && out != nullptr, "cdet_colsum_f32: null pointer");
    CDET_CHECK_ARG(M > 0 && C > 0 && coff >= 0 && coff + C <= ld, "cdet_colsum_f32: bad geometry");
    hipLaunchKernelGGL(colsum_f32_partial_kernel, dim3((C + 63) / 64, BNS_NB), dim3(256), 0, stream, src, ld, coff, M, C, ws);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_f32_final_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, ws, C, out);
    CDET_LAUNCH_CHECK();
    return 0;
}
