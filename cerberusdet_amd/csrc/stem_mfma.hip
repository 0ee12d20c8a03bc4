// Stem convolution on MFMA (gfx950): models/common.py:57 for the first backbone row -- Cin = 3, 3x3, stride 2, pad 1 -- reading the
// reference's NCHW image directly (uint8 scaled by 1/255: trainers/base_trainer.py:61-63, or float / half).
//
// K = 27 is too short for the implicit-GEMM kernels (padding the image to 8 channels makes K = 72 -> 128: 0.5 ms for 0.44 GFLOP/img,
// 5x off the HBM roofline of the 524 MB bf16 output at batch 32). Here one workgroup owns an 8 x 32 patch of output pixels:
//   * its 17 x 65 x 3 input patch is converted once (/255, 16-bit) into LDS (6.7 KB);
//   * each wave gathers the im2col fragment of its 2 x 32 pixels from LDS (16 ds_read_u16 per fragment half, K padded 27 -> 32) and
//     multiplies it with the 96 x 32 weight operand held in registers: 12 x v_mfma_f32_32x32x16 per wave;
//   * epilogue as in conv_halo.hip: BN partial sums from the fp32 accumulators (train), scale / bias / SiLU (eval), permlane32 swap to
//     8 couts per lane, LDS-staged whole-row stores.
// Bound: HBM (output write); the arithmetic is 1 % of the MFMA roof.
#include "common.h"

namespace cdet {

typedef __attribute__((ext_vector_type(16))) float sf32x16;
typedef __attribute__((ext_vector_type(2))) float sf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 sbf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 sf16x2;

struct StemArgs {
    const void* img;
    const float* w;
    const float* scale;
    const float* bias;
    void* y;
    float* stats;
    int img_dtype, N, H, W, Ho, Wo, Cout, act;
    int tiles_x, tiles_y;
};

constexpr int ST_PH = 8, ST_PW = 32;              // output patch
constexpr int ST_IH = 2 * ST_PH + 1, ST_IW = 2 * ST_PW + 1, ST_IWP = ST_IW + 1;  // input patch 17 x 65 (+1 pad)
constexpr int ST_HC = 96, ST_NF = 3;
constexpr int ST_RS = ST_HC * 2 + 16;

template <int DT>
__device__ __forceinline__ uint32_t spack2(float a, float b) {
    if (DT == CDET_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(sf32x2{a, b}, sbf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(sf32x2{a, b}, sf16x2));
}

template <int DT>
__device__ __forceinline__ void smfma32(const u32x4& a, const u32x4& b, sf32x16& c) {
    if (DT == CDET_BF16) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float shalf_sum32(float v) {
#define CDET_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CDET_DPP_ADD(0xB1);
    CDET_DPP_ADD(0x4E);
    CDET_DPP_ADD(0x141);
    CDET_DPP_ADD(0x140);
#undef CDET_DPP_ADD
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    return v;
}

template <int DT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemArgs a) {
    __shared__ uint16_t tile[3 * ST_IH * ST_IWP];
    __shared__ __attribute__((aligned(16))) float red[4 * 2 * ST_HC];
    __shared__ __attribute__((aligned(16))) float sb[2 * ST_HC];
    __shared__ __attribute__((aligned(16))) unsigned char stg[4 * 32 * ST_RS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    if (t < ST_HC) {
        const int c = t < a.Cout ? t : a.Cout - 1;
        sb[t] = a.scale ? a.scale[c] : 1.f;
        sb[ST_HC + t] = a.bias ? a.bias[c] : 0.f;
    }
    // ---- weight operand (A): row co = f*32 + l31, k = s*16 + h*8 + j over the OIHW row [27], zero beyond
    u32x4 af[ST_NF][2];
#pragma unroll
    for (int f = 0; f < ST_NF; ++f) {
        const int co = f * 32 + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + h * 8 + j;
                wv[j] = (co < a.Cout && k < 27) ? a.w[co * 27 + k] : 0.f;
            }
            af[f][s] = u32x4{spack2<DT>(wv[0], wv[1]), spack2<DT>(wv[2], wv[3]), spack2<DT>(wv[4], wv[5]), spack2<DT>(wv[6], wv[7])};
        }
    }
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    unsigned char* const st = stg + wave * (32 * ST_RS);
    constexpr int CH = ST_HC / 8;
    // persistent over patches: the weight operand and scale / bias are set up once per workgroup
    const int n_patches = a.N * a.tiles_y * a.tiles_x;
    for (int patch = blockIdx.x; patch < n_patches; patch += gridDim.x) {
    int b = patch;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int oy0 = ty * ST_PH, ox0 = tx * ST_PW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    __syncthreads();  // the previous patch's fragment gathers are done with `tile`, its statistics with `red`
    // ---- input patch -> LDS (16-bit, uint8 scaled by 1/255); outside the image = the zero padding. 5 threads per (channel, row):
    //      13 consecutive columns each
    if (t < 3 * ST_IH * 5) {
        const int row = t / 5, part = t - row * 5;
        const int c = row / ST_IH, r = row - c * ST_IH;
        const int iy = iy0 + r;
        const bool rok = (unsigned)iy < (unsigned)a.H;
        const int64_t rbase = (((int64_t)n * 3 + c) * a.H + (rok ? iy : 0)) * a.W;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const int col = part * 13 + j;
            const int ix = ix0 + col;
            float v = 0.f;
            if (rok && (unsigned)ix < (unsigned)a.W) {
                const int64_t i = rbase + ix;
                v = a.img_dtype == CDET_U8 ? (float)((const uint8_t*)a.img)[i] * (1.0f / 255.0f) : load_elem(a.img, i, a.img_dtype);
            }
            tile[row * ST_IWP + col] = Elem<DT>::from_f32(v);
        }
    }
    __syncthreads();
    // ---- im2col fragments (B) of this wave's 2 x 32 pixels and the 12 MFMAs
    sf32x16 acc[ST_NF][2];
#pragma unroll
    for (int f = 0; f < ST_NF; ++f)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int base = (2 * (2 * wave + g)) * ST_IWP + 2 * l31;  // tile row 2*oy_l (+kh), column 2*ox_l (+kw)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // k = s*16 + h*8 + j: both candidates are compile-time, the lane's half picks one
                const int k0 = s * 16 + j, k1 = s * 16 + 8 + j;
                const int i0 = k0 < 27 ? ((k0 / 9) * ST_IH + (k0 % 9) / 3) * ST_IWP + k0 % 3 : -1;
                const int i1 = k1 < 27 ? ((k1 / 9) * ST_IH + (k1 % 9) / 3) * ST_IWP + k1 % 3 : -1;
                uint32_t v0 = 0u, v1 = 0u;
                if (i0 >= 0) v0 = tile[base + i0];
                if (i1 >= 0) v1 = tile[base + i1];
                e[j] = h ? v1 : v0;
            }
            const u32x4 bf = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
#pragma unroll
            for (int f = 0; f < ST_NF; ++f) smfma32<DT>(af[f][s], bf, acc[f][g]);
        }
    }
    // ---- BN partial sums (train): pixels outside the image contribute zeros only if their accumulators are zero: mask them
    const int oyw = oy0 + 2 * wave;
    bool pv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) pv[g] = (oyw + g < a.Ho) && (ox0 + l31 < a.Wo);
    if (a.stats != nullptr) {
#pragma unroll
        for (int f = 0; f < ST_NF; ++f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float v0 = pv[g] ? acc[f][g][r] : 0.f;
                    s_ += v0;
                    q_ += v0 * v0;
                }
                const float sv = shalf_sum32(s_), qv = shalf_sum32(q_);
                if (l31 == 0) {
                    const int cl = f * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    red[(wave * 2 + 0) * ST_HC + cl] = sv;
                    red[(wave * 2 + 1) * ST_HC + cl] = qv;
                }
            }
        }
        __syncthreads();
        if (t < a.Cout) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sv += red[(m * 2 + 0) * ST_HC + t];
                qv += red[(m * 2 + 1) * ST_HC + t];
            }
            a.stats[((int64_t)patch * 2 + 0) * a.Cout + t] = sv;
            a.stats[((int64_t)patch * 2 + 1) * a.Cout + t] = qv;
        }
    }
    // ---- epilogue: 8 consecutive couts per lane (permlane32 swap), scale / bias / SiLU, LDS-staged whole-row stores
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int f = 0; f < ST_NF; ++f) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lo = acc[f][g][4 * q + r], hi = acc[f][g][4 * q + 4 + r];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    v[r] = __uint_as_float(s0);
                    v[4 + r] = __uint_as_float(s1);
                }
                const int cl = f * 32 + 8 * (q + h);
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sb + cl), s1 = *reinterpret_cast<const f32x4*>(sb + cl + 4);
                const f32x4 b0v = *reinterpret_cast<const f32x4*>(sb + ST_HC + cl), b1v = *reinterpret_cast<const f32x4*>(sb + ST_HC + cl + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = v[r] * s0[r] + b0v[r];
                    v[4 + r] = v[4 + r] * s1[r] + b1v[r];
                }
                if (a.act == CDET_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                }
                u32x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = spack2<DT>(v[2 * r], v[2 * r + 1]);
                *reinterpret_cast<u32x4*>(st + l31 * ST_RS + cl * 2) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int oy = oyw + g;
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
            const int id = it * 64 + lane;
            const int px = id / CH, c = id - px * CH;
            if (id < 32 * CH) {
                const u32x4 pk = *reinterpret_cast<const u32x4*>(st + px * ST_RS + c * 16);
                const int ox = ox0 + px;
                if (oy < a.Ho && ox < a.Wo && 8 * c < a.Cout)
                    *reinterpret_cast<u32x4*>(yp + (((int64_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + 8 * c) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    }  // patches
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_stem_conv_stat_blocks(int32_t N, int32_t H, int32_t W) {
    return N * div_up(H / 2, ST_PH) * div_up(W / 2, ST_PW);
}

extern "C" int cdet_stem_conv(const void* img, int32_t img_dtype, const float* w, const float* scale, const float* bias, void* y, int32_t N,
                              int32_t H, int32_t W, int32_t Cout, int32_t out_dtype, int32_t act, float* stats, void* stream) {
    CDET_CHECK_ARG(img && w && y, "cdet_stem_conv: null pointer");
    CDET_CHECK_ARG(Cout % 8 == 0 && Cout <= ST_HC, "cdet_stem_conv: Cout must be a multiple of 8 and <= %d (got %d)", ST_HC, Cout);
    CDET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && N > 0, "cdet_stem_conv: H and W must be even");
    CDET_CHECK_ARG(out_dtype == CDET_BF16 || out_dtype == CDET_F16, "cdet_stem_conv: 16-bit output only (the MFMA operands are 16-bit)");
    CDET_CHECK_ARG(img_dtype == CDET_U8 || img_dtype == CDET_F32 || img_dtype == CDET_F16 || img_dtype == CDET_BF16, "cdet_stem_conv: bad image dtype");
    StemArgs a;
    a.img = img; a.w = w; a.scale = scale; a.bias = bias; a.y = y; a.stats = stats;
    a.img_dtype = img_dtype; a.N = N; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Cout = Cout; a.act = act;
    a.tiles_x = div_up(a.Wo, ST_PW);
    a.tiles_y = div_up(a.Ho, ST_PH);
    const int n_patches = N * a.tiles_x * a.tiles_y;
    const int blocks = n_patches < 1024 ? n_patches : 1024;  // persistent: 4 workgroups per CU
    if (out_dtype == CDET_BF16) hipLaunchKernelGGL(stem_mfma_kernel<CDET_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_mfma_kernel<CDET_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    CDET_LAUNCH_CHECK();
    return 0;
}
