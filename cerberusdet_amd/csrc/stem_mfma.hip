// Stem convolution on MFMA (gfx950): models/common.py:57 for the first backbone row -- Cin = 3, 3x3, stride 2, pad 1 -- reading the
// reference's NCHW image directly (uint8 scaled by 1/255: trainers/base_trainer.py:61-63, or float / half).
//
// K = 27 is too short for the implicit-GEMM kernels (padding the image to 8 channels makes K = 72 -> 128: 0.5 ms for 0.44 GFLOP/img,
// 5x off the HBM roofline of the 524 MB bf16 output at batch 32). Here one persistent workgroup walks 8 x 32 patches of output pixels:
//   * the 17 x 65 x 3 input patch of the NEXT patch is fetched with aligned dword loads into registers (4 uint8 / 2 halfs / 1 float
//     per load) while the current patch computes, then converted (/255, 16-bit) into the other half of a double-buffered LDS tile;
//   * the GEMM's K axis is laid out as k = 4 * (c * 3 + kh) + kw with a fourth, zero-weight column, K = 36 -> 48: the eight k values a
//     lane feeds to one v_mfma_f32_32x32x16 are then two runs of four CONSECUTIVE image columns, i.e. two ds_read2_b32 straight from
//     the tile -- no per-element gather, no packing arithmetic (the round-2 kernel spent 16 ds_read_u16 + 8 shifts per fragment half);
//   * 18 MFMAs per wave and patch against the 96 x 48 weight operand held in registers;
//   * epilogue as in conv_halo.hip: BN partial sums from the fp32 accumulators (train), scale / bias / SiLU (eval), permlane32 swap to
//     8 couts per lane, LDS-staged whole-row stores.
// Bound: HBM (output write); the arithmetic is 1.5 % of the MFMA roof.
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
constexpr int ST_IH = 2 * ST_PH + 1;              // input rows per channel (17); columns: entries c' = 0 .. 65 <-> image column 64*tx - 1 + c'
constexpr int ST_ROWS = 3 * ST_IH;                // LDS tile rows (channel-major)
constexpr int ST_PITCH = 136;                     // bytes per tile row: 66 entries of 2 bytes, padded (row -> row shifts 2 banks)
constexpr int ST_TILE = (ST_ROWS * ST_PITCH + 15) / 16 * 16;
constexpr int ST_HC = 96, ST_NF = 3, ST_KS = 3;   // couts per block, 32-cout fragments, k16 steps (K = 48)
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

// BatchNorm partial sums of one 32 x 32 accumulator tile over its 32 pixels (same scheme as halo_common.h tile_stats32: v_permlane16_swap pairs
// registers r / r + 8, four DPP levels inside the 16-lane rows, lanes 0 / 16 / 32 / 48 store the totals)
__device__ __forceinline__ float srow_sum16(float v) {
#define CDET_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CDET_DPP_ADD(0xB1);
    CDET_DPP_ADD(0x4E);
    CDET_DPP_ADD(0x141);
    CDET_DPP_ADD(0x140);
#undef CDET_DPP_ADD
    return v;
}

__device__ __forceinline__ void stile_stats32(const float (&s)[16], const float (&q)[16], float* ds, float* dq, int lane) {
    float ts[8], tq[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[r]), __float_as_uint(s[r + 8]), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[r]), __float_as_uint(q[r + 8]), false, false);
        const unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        ts[r] = srow_sum16(__uint_as_float(a0) + __uint_as_float(a1));
        tq[r] = srow_sum16(__uint_as_float(b0) + __uint_as_float(b1));
    }
    if ((lane & 15) == 0) {
        const int base = 4 * (lane >> 5) + 16 * ((lane >> 4) & 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ds[base + 8 * (r >> 2) + (r & 3)] = ts[r];
            dq[base + 8 * (r >> 2) + (r & 3)] = tq[r];
        }
    }
}

// A: image elements per aligned dword load (4 = uint8, 2 = half / bfloat16, 1 = float)
template <int DT, int A>
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const StemArgs a) {
    constexpr int ND = (A + 64) / A;                    // dwords per tile row: columns 64*tx - A .. 64*tx + 63
    constexpr int NITEM = ST_ROWS * ND;
    constexpr int NPT = (NITEM + 255) / 256;            // prefetch registers per thread: 4 / 7 / 13
    __shared__ __attribute__((aligned(16))) unsigned char tiles[2 * ST_TILE];
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
    // both tile buffers start as zeros: entries that are never written (c' = 65) must hold finite values (their weights are zero)
    for (int i = t; i < 2 * ST_TILE / 4; i += 256) reinterpret_cast<uint32_t*>(tiles)[i] = 0u;
    // ---- weight operand (A): row co = f*32 + l31; k = 16*s + 8*h + j  <->  (c*3 + kh) = k / 4, kw = k % 4 (kw = 3 and rows >= 9: zero)
    u32x4 af[ST_NF][ST_KS];
#pragma unroll
    for (int f = 0; f < ST_NF; ++f) {
        const int co = f * 32 + l31;
#pragma unroll
        for (int s = 0; s < ST_KS; ++s) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + h * 8 + j;
                const int r = k >> 2, kw = k & 3;
                wv[j] = (co < a.Cout && r < 9 && kw < 3) ? a.w[co * 27 + r * 3 + kw] : 0.f;
            }
            af[f][s] = u32x4{spack2<DT>(wv[0], wv[1]), spack2<DT>(wv[2], wv[3]), spack2<DT>(wv[4], wv[5]), spack2<DT>(wv[6], wv[7])};
        }
    }
    // ---- B-fragment addresses (bytes inside a tile buffer): for (g, s) the lane reads 4 entries of tile row R(4s + 2h) and 4 of
    //      R(4s + 2h + 1), R(r) = (r / 3) * 17 + r % 3 + 2 * oy_l, from entry 2 * ox_l on (rows beyond 8 re-read row 8: zero weights)
    int boff[2][ST_KS][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < ST_KS; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int r = 4 * s + 2 * h + q;
                r = r > 8 ? 8 : r;
                boff[g][s][q] = ((r / 3) * ST_IH + r % 3 + 2 * (2 * wave + g)) * ST_PITCH + 4 * l31;
            }
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    unsigned char* const st = stg + wave * (32 * ST_RS);
    constexpr int CH = ST_HC / 8;
    const int n_patches = a.N * a.tiles_y * a.tiles_x;
    const int es = A == 4 ? 1 : (A == 2 ? 2 : 4);

    uint32_t pre[NPT];
    // raw dwords of patch `pt` -> registers (zeros outside the image)
    auto fetch = [&](int pt) {
        int b = pt;
        const int tx = b % a.tiles_x;
        b /= a.tiles_x;
        const int ty = b % a.tiles_y;
        const int n = b / a.tiles_y;
        const int iy0 = 2 * ty * ST_PH - 1, col00 = 64 * tx - A;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int idx = i * 256 + t;
            uint32_t v = 0u;
            if (idx < NITEM) {
                const int row = idx / ND, d = idx - row * ND;
                const int c = row / ST_IH, r = row - c * ST_IH;
                const int iy = iy0 + r, col0 = col00 + d * A;
                if ((unsigned)iy < (unsigned)a.H && col0 >= 0 && col0 < a.W) {
                    const int64_t e0 = (((int64_t)n * 3 + c) * a.H + iy) * a.W + col0;
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.img) + e0 * es;
                    if (col0 + A <= a.W) {
                        v = *reinterpret_cast<const uint32_t*>(src);
                    } else {  // W not a multiple of the load width: the row's last, partial dword element by element
#pragma unroll
                        for (int e = 0; e < A; ++e)
                            if (col0 + e < a.W) {
                                if (A == 4) v |= (uint32_t)src[e] << (8 * e);
                                else v |= (uint32_t)reinterpret_cast<const uint16_t*>(src)[e] << (16 * e);
                            }
                    }
                }
            }
            pre[i] = v;
        }
    };
    // registers -> 16-bit tile entries of buffer `buf` (entry c' = 1 - A + d*A + e)
    auto stash = [&](int buf) {
        uint16_t* const tl = reinterpret_cast<uint16_t*>(tiles + buf * ST_TILE);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int idx = i * 256 + t;
            if (idx < NITEM) {
                const int row = idx / ND, d = idx - row * ND;
                const uint32_t v = pre[i];
#pragma unroll
                for (int e = 0; e < A; ++e) {
                    const int cp = 1 - A + d * A + e;
                    float f;
                    if (A == 4) f = (float)((v >> (8 * e)) & 0xffu) * (1.0f / 255.0f);
                    else if (A == 2) f = a.img_dtype == CDET_F16 ? f16_bits_to_f32((uint16_t)(v >> (16 * e))) : bf16_bits_to_f32((uint16_t)(v >> (16 * e)));
                    else f = __uint_as_float(v);
                    if (cp >= 0) tl[row * (ST_PITCH / 2) + cp] = Elem<DT>::from_f32(f);
                }
            }
        }
    };

    int patch = blockIdx.x;
    if (patch < n_patches) fetch(patch);
    __syncthreads();  // the zero fill (and sb) before the first tile entries land
    stash(0);
    __syncthreads();
    int cur = 0;
    for (; patch < n_patches; patch += gridDim.x) {
    int b = patch;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int oy0 = ty * ST_PH, ox0 = tx * ST_PW;
    const bool more = patch + (int)gridDim.x < n_patches;
    if (more) fetch(patch + gridDim.x);  // in flight under this patch's MFMAs and epilogue
    // ---- B fragments straight from the tile (two ds_read2_b32 each) and the 18 MFMAs
    sf32x16 acc[ST_NF][2];
#pragma unroll
    for (int f = 0; f < ST_NF; ++f)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;
    const unsigned char* const tb = tiles + cur * ST_TILE;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int s = 0; s < ST_KS; ++s) {
            const uint32_t* p0 = reinterpret_cast<const uint32_t*>(tb + boff[g][s][0]);
            const uint32_t* p1 = reinterpret_cast<const uint32_t*>(tb + boff[g][s][1]);
            const u32x4 bf = u32x4{p0[0], p0[1], p1[0], p1[1]};
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
            float s16[16], q16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float v0 = pv[g] ? acc[f][g][r] : 0.f;
                    s_ += v0;
                    q_ += v0 * v0;
                }
                s16[r] = s_;
                q16[r] = q_;
            }
            stile_stats32(s16, q16, red + (wave * 2 + 0) * ST_HC + f * 32, red + (wave * 2 + 1) * ST_HC + f * 32, lane);
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
    if (more) stash(cur ^ 1);  // (waits for the prefetched dwords) the other buffer was last read before the previous barrier
    __syncthreads();
    cur ^= 1;
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
    const int blocks = n_patches < 512 ? n_patches : 512;  // persistent: 2 workgroups per CU (the register budget of __launch_bounds__(256, 2))
    const int per = img_dtype == CDET_U8 ? 4 : (img_dtype == CDET_F32 ? 1 : 2);  // image elements per dword load
    hipStream_t s = (hipStream_t)stream;
#define CDET_STEM_LAUNCH(DT_, A_) hipLaunchKernelGGL((stem_mfma_kernel<DT_, A_>), dim3(blocks), dim3(256), 0, s, a)
    if (out_dtype == CDET_BF16) {
        if (per == 4) CDET_STEM_LAUNCH(CDET_BF16, 4);
        else if (per == 2) CDET_STEM_LAUNCH(CDET_BF16, 2);
        else CDET_STEM_LAUNCH(CDET_BF16, 1);
    } else {
        if (per == 4) CDET_STEM_LAUNCH(CDET_F16, 4);
        else if (per == 2) CDET_STEM_LAUNCH(CDET_F16, 2);
        else CDET_STEM_LAUNCH(CDET_F16, 1);
    }
#undef CDET_STEM_LAUNCH
    CDET_LAUNCH_CHECK();
    return 0;
}
