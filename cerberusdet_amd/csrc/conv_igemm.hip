// Implicit-GEMM convolution on MFMA for gfx950 (bf16 / f16 in, fp32 accumulate).
//
// GEMM view (FWD):  D[co][p] = sum_k W[co][k] * X[p][k],  k = (kh, kw, ci) flattened, p = (n, oy, ox).
// Both operands are K-contiguous: NHWC activations give 8-channel (16 B) vectors per (pixel, tap),
// weights are pre-packed [Cout][Kpad]. The MFMA "A" operand is the weight tile and "B" the pixel
// tile, so each lane's 4 accumulator registers are 4 CONSECUTIVE output channels of one pixel ->
// 8-byte NHWC stores and lane-local per-channel BN statistics.
//
// Tiling: wave tile = 80 couts x 64 pixels (5 x 4 MFMA 16x16x32 tiles, 80 fp32 accumulators/lane);
// YOLOv8x widths are multiples of 80 (80/160/320/640), so 80-wide wave tiles waste nothing.
//   <2,2>: block = 160 couts x 128 pixels (Cout >= 160)      <4,1>: block = 80 couts x 256 pixels (Cout <= 80)
// K step 64 (128-B LDS rows, XOR-swizzled 16-B slots -> conflict-free ds_read_b128 fragments),
// single LDS stage + register prefetch of the next K step (global loads fly under the MFMAs).
// Workgroup ids are remapped so that the cout-blocks of one pixel-block run back to back on ONE XCD
// (their im2col tile is served by that XCD's L2 instead of HBM).
#include <stdlib.h>

#include "common.h"

#ifndef CDET_SETPRIO
#define CDET_SETPRIO 1  // s_setprio(1) around the MFMA cluster: +7 % on the dominant layer (2 workgroups per CU arbitrate)
#endif

namespace cdet {

struct ConvArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* bias;
    const uint16_t* res;
    void* y;
    float* stats;
    int N, Hs, Ws, Cs, Hd, Wd, Cd;
    int KH, KW, stride, pad;
    int src_ld, src_coff, dst_ld, dst_coff, res_ld, res_coff;
    int Kpad, nk;       // packed K (multiple of 64), number of 64-wide K steps
    int M;              // N*Hd*Wd
    int n_pblk, n_cblk;
    int act, out_dtype, accumulate;
    // stride-2 data gradient, decomposed by output-pixel parity (glds family, MODE 2): class q owns the dX pixels with
    // (y & 1, x & 1) == (cy, cx); only the taps kh = kh0 + 2*ih, kw = kw0 + 2*iw reach them (2.25 of 9 on average for 3x3)
    struct ParClass {
        int cy, cx, kh0, kw0, nh, nw, Hc, Wc, M, nk, blk0, nblk;
    } pc[4];
    int par_interleave;  // all four classes have the same tile count: class = L & 3, so the four classes of one pixel tile run together
    unsigned x_bytes, w_bytes;  // extents of the source / packed-weight buffers (buffer-resource bounds of the pipelined kernel)
};

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // 128

__device__ __forceinline__ int lds_slot(int row, int kvec) { return row * ROW_BYTES + ((kvec ^ ((row >> 1) & 7)) << 4); }

template <int DT> struct Mfma;
template <> struct Mfma<CDET_BF16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<CDET_F16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// in-place MFMA (accumulator tied to an AGPR tuple): the pipelined kernel touches every accumulator twice per loop iteration
// from two differently scheduled phases, and the register allocator otherwise ping-pongs them through ~116 v_accvgpr copies
template <int DT>
__device__ __forceinline__ void mfma_inplace(const u32x4& a, const u32x4& b, f32x4& c) {
    if (DT == CDET_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int DT, int WAVES_M, int WAVES_N, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BP = 64 * WAVES_M;          // pixels per block
    constexpr int BC = 80 * WAVES_N;          // couts per block
    constexpr int XR = BP / 32;               // X rows per thread
    constexpr int WR = (BC + 31) / 32;        // W rows per thread (last may be partial)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Wsm = smem + BP * ROW_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // ---- XCD-aware block remap (bijective): logical id L walks cout-blocks fastest --------------
    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int cblk = L % a.n_cblk;
    const int pblk = L / a.n_cblk;
    const int p0 = pblk * BP;
    const int c0 = cblk * BC;

    // ---- per-thread load bookkeeping --------------------------------------------------------------
    const int kvec = t & 7;
    const int lrow = t >> 3;  // 0..31
    int ybase[XR], xbase[XR], nbase[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int p = p0 + lrow + 32 * i;
        if (p < a.M) {
            const int hw = a.Hd * a.Wd;
            const int n = p / hw;
            const int rem = p - n * hw;
            const int py = rem / a.Wd;
            const int px = rem - py * a.Wd;
            nbase[i] = n * a.Hs;
            if (DGRAD) {
                ybase[i] = py + a.pad;
                xbase[i] = px + a.pad;
            } else {
                ybase[i] = py * a.stride - a.pad;
                xbase[i] = px * a.stride - a.pad;
            }
        } else {
            nbase[i] = 0;
            ybase[i] = -(1 << 20);  // every tap fails the bounds test
            xbase[i] = -(1 << 20);
        }
    }
    // position of this thread's 8-element K vector: channel c within tap (kh, kw)
    int kc, kkh, kkw;
    {
        const int k = kvec * 8;
        const int tap = k / a.Cs;
        kc = k - tap * a.Cs;
        kkh = tap / a.KW;
        kkw = tap - kkh * a.KW;
    }
    const uint16_t* wrow[WR];
    bool wok[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int r = lrow + 32 * i;
        const int co = c0 + r;
        wok[i] = (r < BC) && (co < a.Cd);
        wrow[i] = a.w + (int64_t)(wok[i] ? co : 0) * a.Kpad + kvec * 8;
    }

    u32x4 xreg[XR], wreg[WR];
    auto load_global = [&](int ks) {
        const bool ktap_ok = kkh < a.KH;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            int sy, sx;
            bool ok = ktap_ok;
            if (DGRAD) {
                const int ty = ybase[i] - kkh, tx = xbase[i] - kkw;
                if (a.stride == 2) {
                    ok = ok && (((ty | tx) & 1) == 0);
                    sy = ty >> 1;
                    sx = tx >> 1;
                } else {
                    sy = ty;
                    sx = tx;
                }
            } else {
                sy = ybase[i] + kkh;
                sx = xbase[i] + kkw;
            }
            ok = ok && ((unsigned)sy < (unsigned)a.Hs) && ((unsigned)sx < (unsigned)a.Ws);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const int64_t off = ((int64_t)(nbase[i] + sy) * a.Ws + sx) * a.src_ld + a.src_coff + kc;
                v = *reinterpret_cast<const u32x4*>(a.x + off);
            }
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (wok[i]) v = *reinterpret_cast<const u32x4*>(wrow[i] + (int64_t)ks * BK);
            wreg[i] = v;
        }
        // advance the K cursor by one step (64 elements)
        kc += BK;
        while (kc >= a.Cs) {
            kc -= a.Cs;
            if (++kkw == a.KW) {
                kkw = 0;
                ++kkh;
            }
        }
    };

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = lane >> 4;  // 0..3

    load_global(0);
    for (int ks = 0; ks < a.nk; ++ks) {
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<u32x4*>(Xs + lds_slot(lrow + 32 * i, kvec)) = xreg[i];
#pragma unroll
        for (int i = 0; i < WR; ++i)
            if (lrow + 32 * i < BC) *reinterpret_cast<u32x4*>(Wsm + lds_slot(lrow + 32 * i, kvec)) = wreg[i];
        __syncthreads();
        if (ks + 1 < a.nk) load_global(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[5], bf[4];
#pragma unroll
            for (int i = 0; i < 5; ++i) af[i] = *reinterpret_cast<const u32x4*>(Wsm + lds_slot(wn * 80 + i * 16 + frow, kk * 4 + fk));
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(Xs + lds_slot(wm * 64 + j * 16 + frow, kk * 4 + fk));
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<DT>::run(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- BN statistics of the raw convolution (train mode) --------------------------------------
    if (a.stats != nullptr) {
        float* st = reinterpret_cast<float*>(smem);  // [WAVES_M][2][BC]
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s += acc[i][j];
                q += acc[i][j] * acc[i][j];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv = s[r], qv = q[r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    sv += __shfl_xor(sv, m);
                    qv += __shfl_xor(qv, m);
                }
                if (frow == 0) {
                    const int cl = wn * 80 + i * 16 + fk * 4 + r;
                    st[(wm * 2 + 0) * BC + cl] = sv;
                    st[(wm * 2 + 1) * BC + cl] = qv;
                }
            }
        }
        __syncthreads();
        if (t < BC && c0 + t < a.Cd) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < WAVES_M; ++m) {
                sv += st[(m * 2 + 0) * BC + t];
                qv += st[(m * 2 + 1) * BC + t];
            }
            a.stats[((int64_t)pblk * 2 + 0) * a.Cd + c0 + t] = sv;
            a.stats[((int64_t)pblk * 2 + 1) * a.Cd + c0 + t] = qv;
        }
    }

    // ---- epilogue: scale/bias, activation, residual, store -------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int co = c0 + wn * 80 + i * 16 + fk * 4;
        if (co >= a.Cd) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + co);
        if (a.bias) bi = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + wm * 64 + j * 16 + frow;
            if (p >= a.M) continue;
            f32x4 v = acc[i][j] * sc + bi;
            if (a.act == CDET_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
            }
            if (a.res) {
                const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + (int64_t)p * a.res_ld + a.res_coff + co);
                v[0] += Elem<DT>::to_f32((uint16_t)(rv[0] & 0xffff));
                v[1] += Elem<DT>::to_f32((uint16_t)(rv[0] >> 16));
                v[2] += Elem<DT>::to_f32((uint16_t)(rv[1] & 0xffff));
                v[3] += Elem<DT>::to_f32((uint16_t)(rv[1] >> 16));
            }
            const int64_t o = (int64_t)p * a.dst_ld + a.dst_coff + co;
            if (a.out_dtype == CDET_F32) {
                float* yp = reinterpret_cast<float*>(a.y) + o;
                if (a.accumulate) v += *reinterpret_cast<const f32x4*>(yp);
                *reinterpret_cast<f32x4*>(yp) = v;
            } else {
                u32x2 pk;
                if (a.out_dtype == CDET_BF16) {
                    pk[0] = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
                    pk[1] = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
                } else {
                    pk[0] = (uint32_t)f32_to_f16_bits(v[0]) | ((uint32_t)f32_to_f16_bits(v[1]) << 16);
                    pk[1] = (uint32_t)f32_to_f16_bits(v[2]) | ((uint32_t)f32_to_f16_bits(v[3]) << 16);
                }
                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + o) = pk;
            }
        }
    }
}

// ================================================================================================
// v2 (default): same tiling / fragments as above, re-engineered around what the ablations on the MI355X showed
// (profiles/r01_conv_ablation.txt): the K loop was NOT the problem -- ~6000 instructions of prologue + epilogue per wave
// cost ~20 us per workgroup (as much as the 45 K steps of the dominant layer).
//   * operand staging: HBM -> LDS directly with global_load_lds (16 B per lane, no VGPR staging / ds_write pass). The LDS
//     image of a wave-instruction is lane-linear (8 rows x 128 B), so the XOR swizzle is applied to the SOURCE address:
//     the lane that owns physical slot s of row r fetches logical k-vector s ^ ((r>>1)&7); padding taps fetch a zero page;
//   * two LDS stages, ONE barrier per K step: the DMA of step t+1 is issued before the MFMAs of step t and waited for
//     (vmcnt(0), placed by the compiler in front of the barrier) after them;
//   * prologue: one integer division per thread (further rows advance incrementally), tap validity as a bit mask built
//     from separable row / column bits, 32-bit element offsets;
//   * epilogue: compile-time specialised (raw | scale-bias-act-residual) x (16-bit | fp32 out), v_cvt_pk_bf16_f32, rcp
//     instead of IEEE division in SiLU, row base addresses hoisted.
// ================================================================================================
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[16];

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int DT>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    if (DT == CDET_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}

// bit k set <=> tap k of one axis reads a valid source coordinate
template <bool DGRAD>
__device__ __forceinline__ unsigned axis_bits(int base, int K, int stride, int limit) {
    unsigned bits = 0u;
    for (int k = 0; k < K; ++k) {
        int s;
        bool ok = true;
        if (DGRAD) {
            const int t = base - k;
            if (stride == 2) {
                ok = (t & 1) == 0;
                s = t >> 1;
            } else {
                s = t;
            }
        } else {
            s = base + k;
        }
        if (ok && (unsigned)s < (unsigned)limit) bits |= 1u << k;
    }
    return bits;
}

constexpr int EPI_RAW = 0, EPI_FULL = 1;

// sum over the 16 lanes of a DPP row; every lane ends up with the total
__device__ __forceinline__ float dpp_sum16(float v) {
#define CDET_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CDET_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    CDET_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    CDET_DPP_ADD(0x141);  // row_half_mirror
    CDET_DPP_ADD(0x140);  // row_mirror
#undef CDET_DPP_ADD
    return v;
}

// BN partial statistics + store of one block's accumulators (shared by the glds kernels). PAR: the block's pixels are the
// class-local pixels of one dX parity class and are scattered back to (2*py + cy, 2*px + cx).
template <int DT, int WAVES_M, int WAVES_N, bool PAR, int EPI, bool OUT_F32>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[5][4], unsigned char* smem, int t, int wm, int wn, int frow, int fk,
                                              int p0, int c0, int pblk, int Mloc, int gH, int gW, int cy, int cx) {
    constexpr int BC = 80 * WAVES_N;
    // ---- BN statistics of the raw convolution (train mode) --------------------------------------------------------------
    if (a.stats != nullptr) {
        float* st = reinterpret_cast<float*>(smem);  // [WAVES_M][2][BC]
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            f32x4 s = acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
            f32x4 q = acc[i][0] * acc[i][0] + acc[i][1] * acc[i][1] + acc[i][2] * acc[i][2] + acc[i][3] * acc[i][3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // sum over the 16 pixel lanes of the MFMA tile row (one DPP row): VALU only, no LDS crossbar
                s[r] = dpp_sum16(s[r]);
                q[r] = dpp_sum16(q[r]);
            }
            if (frow == 0) {
                const int cl = wn * 80 + i * 16 + fk * 4;
                *reinterpret_cast<f32x4*>(st + (wm * 2 + 0) * BC + cl) = s;
                *reinterpret_cast<f32x4*>(st + (wm * 2 + 1) * BC + cl) = q;
            }
        }
        __syncthreads();
        if (t < BC && c0 + t < a.Cd) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < WAVES_M; ++m) {
                sv += st[(m * 2 + 0) * BC + t];
                qv += st[(m * 2 + 1) * BC + t];
            }
            a.stats[((int64_t)pblk * 2 + 0) * a.Cd + c0 + t] = sv;
            a.stats[((int64_t)pblk * 2 + 1) * a.Cd + c0 + t] = qv;
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    int64_t obase[4];
    int64_t rbase[4];
    bool pvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p = p0 + wm * 64 + j * 16 + frow;
        pvalid[j] = p < Mloc;
        if (PAR) {  // class pixel -> dX pixel
            const int hw = gH * gW;
            const int n = p / hw;
            const int rem = p - n * hw;
            const int py = rem / gW;
            const int px = rem - py * gW;
            p = (n * a.Hd + 2 * py + cy) * a.Wd + 2 * px + cx;
        }
        obase[j] = (int64_t)p * a.dst_ld + a.dst_coff;
        rbase[j] = (int64_t)p * a.res_ld + a.res_coff;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int co = c0 + wn * 80 + i * 16 + fk * 4;
        if (co >= a.Cd) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_FULL) {
            if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + co);
            if (a.bias) bi = *reinterpret_cast<const f32x4*>(a.bias + co);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!pvalid[j]) continue;
            f32x4 v = acc[i][j];
            if (EPI == EPI_FULL) {
                v = v * sc + bi;
                if (a.act == CDET_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                }
                if (a.res) {
                    const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + rbase[j] + co);
                    v[0] += Elem<DT>::to_f32((uint16_t)(rv[0] & 0xffff));
                    v[1] += Elem<DT>::to_f32((uint16_t)(rv[0] >> 16));
                    v[2] += Elem<DT>::to_f32((uint16_t)(rv[1] & 0xffff));
                    v[3] += Elem<DT>::to_f32((uint16_t)(rv[1] >> 16));
                }
            }
            if (OUT_F32) {
                float* yp = reinterpret_cast<float*>(a.y) + obase[j] + co;
                if (a.accumulate) v += *reinterpret_cast<const f32x4*>(yp);
                *reinterpret_cast<f32x4*>(yp) = v;
            } else {
                u32x2 pk;
                pk[0] = pack2<DT>(v[0], v[1]);
                pk[1] = pack2<DT>(v[2], v[3]);
                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + obase[j] + co) = pk;
            }
        }
    }
}

// 8-wave instantiation (<4,2>: 256 pixels x 160 couts, 512 threads, ONE workgroup per CU): three LDS stages (3 x 52 KiB = 156 KiB of
// the CU's 160 KiB) and a counted-vmcnt pipeline -- the DMA of K step t+2 is issued while step t is computed and step t+1 is
// still in flight; a wave only waits for ITS OWN pieces of step t (`s_waitcnt vmcnt(pieces of one step)`), then one raw
// s_barrier per step publishes the tile. __syncthreads() is avoided inside the loop because it would drain the DMA (vmcnt(0)).
template <int DT, int WAVES_M, int WAVES_N, int MODE, int EPI, bool OUT_F32>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_glds_kernel(const ConvArgs a) {
    constexpr bool DGRAD = MODE != 0;
    constexpr bool PAR = MODE == 2;  // stride-2 dgrad split into the four output-parity classes
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr bool DEEP = NW == 8;              // 3-stage counted pipeline
    constexpr bool SETPRIO = CDET_SETPRIO != 0;
    constexpr int BP = 64 * WAVES_M;
    constexpr int BC = 80 * WAVES_N;
    constexpr int XI = BP / (8 * NW);           // X wave-instructions per wave per K step (8 rows each)
    constexpr int WI_TOTAL = BC / 8;            // W wave-instructions per K step in the block
    constexpr int WI = (WI_TOTAL + NW - 1) / NW;  // per wave (last waves may have one fewer)
    constexpr int STAGE = (BP + BC) * ROW_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // parity mode: the linear block index selects the class and the tile inside it. With equal tile counts the classes interleave
    // (the four classes of a pixel tile write the four pixels of every 2x2 cell: run together on one XCD their partial lines merge
    // in L2); otherwise class by class, heaviest first
    int cls = 0;
    if (PAR) {
        if (a.par_interleave) {
            cls = L & 3;
            L >>= 2;
        } else {
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (L >= a.pc[q].blk0) cls = q;
            L -= a.pc[cls].blk0;
        }
    }
    const int cy = PAR ? a.pc[cls].cy : 0, cx = PAR ? a.pc[cls].cx : 0;
    const int kh0 = PAR ? a.pc[cls].kh0 : 0, kw0 = PAR ? a.pc[cls].kw0 : 0;
    const int tapsW = PAR ? a.pc[cls].nw : a.KW;                       // taps per kernel row walked by the K cursor
    const int ntap = PAR ? a.pc[cls].nh * a.pc[cls].nw : a.KH * a.KW;  // taps walked by the K loop
    const int gH = PAR ? a.pc[cls].Hc : a.Hd, gW = PAR ? a.pc[cls].Wc : a.Wd;  // pixel grid the M index enumerates
    const int Mloc = PAR ? a.pc[cls].M : a.M;
    const int nk = PAR ? a.pc[cls].nk : a.nk;
    const int cblk = L % a.n_cblk;
    const int pblk = L / a.n_cblk;
    const int p0 = pblk * BP;
    const int c0 = cblk * BC;

    const int slot = lane & 7;
    const int lr = lane >> 3;  // row within a wave-instruction
    // ---- X rows owned by this thread: row(j) = wave*8*XI + 8*j + lr ---------------------------------------------------
    int rowoff[XI];
    unsigned tapmask[XI];
    {
        int p = p0 + wave * (8 * XI) + lr;
        const int hw = gH * gW;
        int n = p / hw;
        int rem = p - n * hw;
        int py = rem / gW;
        int px = rem - py * gW;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            rowoff[j] = 0;
            tapmask[j] = 0u;
            if (PAR) {
                if (p < Mloc) {
                    // source coordinate of class tap (ih, iw): (oy - ih, ox - iw), oy = (y + pad - kh0) / 2 (exact)
                    const int oy = (2 * py + cy + a.pad - kh0) >> 1, ox = (2 * px + cx + a.pad - kw0) >> 1;
                    unsigned rb = 0u, cb = 0u;
                    for (int ih = 0; ih < a.pc[cls].nh; ++ih)
                        if ((unsigned)(oy - ih) < (unsigned)a.Hs) rb |= 1u << ih;
                    for (int iw = 0; iw < tapsW; ++iw)
                        if ((unsigned)(ox - iw) < (unsigned)a.Ws) cb |= 1u << iw;
                    unsigned m = 0u;
                    for (int ih = 0; ih < a.pc[cls].nh; ++ih)
                        if ((rb >> ih) & 1u) m |= cb << (ih * tapsW);
                    tapmask[j] = m;
                    rowoff[j] = ((n * a.Hs + oy) * a.Ws + ox) * a.src_ld + a.src_coff;
                }
            } else if (p < a.M) {
                int by, bx;  // base source coordinates of tap (0,0)
                if (DGRAD) {
                    by = py + a.pad;
                    bx = px + a.pad;
                } else {
                    by = py * a.stride - a.pad;
                    bx = px * a.stride - a.pad;
                }
                const unsigned rb = axis_bits<DGRAD>(by, a.KH, a.stride, a.Hs);
                const unsigned cb = axis_bits<DGRAD>(bx, a.KW, a.stride, a.Ws);
                unsigned m = 0u;
                for (int kh = 0; kh < a.KH; ++kh)
                    if ((rb >> kh) & 1u) m |= cb << (kh * a.KW);
                tapmask[j] = m;
                const int oy = (DGRAD && a.stride == 2) ? by >> 1 : by;
                const int ox = (DGRAD && a.stride == 2) ? bx >> 1 : bx;
                rowoff[j] = ((n * a.Hs + oy) * a.Ws + ox) * a.src_ld + a.src_coff;
            }
            p += 8;
            px += 8;
            while (px >= gW) {
                px -= gW;
                if (++py == gH) {
                    py = 0;
                    ++n;
                }
            }
        }
    }
    // two K cursors: even j use k-vector (slot ^ sw0), odd j use that ^ 4 (rows 8 apart flip bit 2 of (row>>1)&7)
    const int sw0 = ((wave * (8 * XI) + lr) >> 1) & 7;
    int kc[2], ktap[2], kkh[2], kkw[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int kv = slot ^ sw0 ^ (c ? 4 : 0);
        const int k = kv * 8;
        ktap[c] = k / a.Cs;
        kc[c] = k - ktap[c] * a.Cs;
        kkh[c] = ktap[c] / tapsW;
        kkw[c] = ktap[c] - kkh[c] * tapsW;
    }
    // ---- W rows: wave-instruction id wi = wave + NW*j covers rows 8*wi + lr --------------------------------------------
    // (parity mode: the class's taps are not contiguous in the packed [kh][kw][c] rows, so the W pieces follow the X cursors:
    //  W row-group wi uses cursor (wi ^ wave*XI) & 1 -- the same swizzle phase -- and only its row base is kept here)
    const uint16_t* wptr[WI];
    bool wadv[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int wi = wave + NW * j;
        const int r = 8 * wi + lr;
        const int co = c0 + r;
        const int kv = slot ^ ((r >> 1) & 7);
        wadv[j] = wi < WI_TOTAL && co < a.Cd;  // rows beyond Cd stream zeros from the zero page (pointer not advanced)
        wptr[j] = wadv[j] ? a.w + (int64_t)co * a.Kpad + (PAR ? 0 : kv * 8) : reinterpret_cast<const uint16_t*>(g_zero_page);
    }

    auto stage = [&](int buf) {
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* wsm = xs + BP * ROW_BYTES;
        int toff[2];
        bool tin[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            tin[c] = ktap[c] < ntap;
            int o;
            if (PAR) o = -((kkh[c] * a.Ws + kkw[c]) * a.src_ld);
            else if (DGRAD) o = a.stride == 2 ? -(((kkh[c] >> 1) * a.Ws + (kkw[c] >> 1)) * a.src_ld) : -((kkh[c] * a.Ws + kkw[c]) * a.src_ld);
            else o = (kkh[c] * a.Ws + kkw[c]) * a.src_ld;
            toff[c] = o + kc[c];
        }
        int wk[2];  // parity mode: K offset inside a packed weight row of each cursor's piece
        if (PAR) {
#pragma unroll
            for (int c = 0; c < 2; ++c) wk[c] = ((kh0 + 2 * kkh[c]) * a.KW + kw0 + 2 * kkw[c]) * a.Cs + kc[c];
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int c = j & 1;
            const bool ok = tin[c] && ((tapmask[j] >> ktap[c]) & 1u);
            const uint16_t* src = ok ? a.x + (rowoff[j] + toff[c]) : reinterpret_cast<const uint16_t*>(g_zero_page);
            glds16(src, xs + (wave * (8 * XI) + 8 * j) * ROW_BYTES);
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            if (wave + NW * j < WI_TOTAL) {
                if (PAR) {
                    const int c = ((wave + NW * j) ^ (wave * XI)) & 1;
                    const uint16_t* src = (wadv[j] && tin[c]) ? wptr[j] + wk[c] : reinterpret_cast<const uint16_t*>(g_zero_page);
                    glds16(src, wsm + (8 * (wave + NW * j)) * ROW_BYTES);
                } else {
                    glds16(wptr[j], wsm + (8 * (wave + NW * j)) * ROW_BYTES);
                    if (wadv[j]) wptr[j] += BK;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            kc[c] += BK;
            while (kc[c] >= a.Cs) {
                kc[c] -= a.Cs;
                ++ktap[c];
                if (++kkw[c] == tapsW) {
                    kkw[c] = 0;
                    ++kkh[c];
                }
            }
        }
    };

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = lane >> 4;

    auto compute = [&](int buf) {
        const unsigned char* Xs = smem + buf * STAGE;
        const unsigned char* Wsm = Xs + BP * ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[5], bf[4];
#pragma unroll
            for (int i = 0; i < 5; ++i) af[i] = *reinterpret_cast<const u32x4*>(Wsm + lds_slot(wn * 80 + i * 16 + frow, kk * 4 + fk));
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(Xs + lds_slot(wm * 64 + j * 16 + frow, kk * 4 + fk));
            if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<DT>::run(af[i], bf[j], acc[i][j]);
            if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    if (DEEP) {
        // pieces (glds instructions) this wave issues per K step: XI + (its share of the W chunks)
        const bool w_extra = wave + NW * (WI - 1) < WI_TOTAL;  // wave-uniform
        stage(0);
        if (nk > 1) stage(1);
        int buf = 0, nbuf = 2;
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) {  // step ks+1 may stay in flight
                if (w_extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XI + WI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XI + WI - 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // tile ks complete for every wave; everyone is done reading the buffer of tile ks-1
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < nk) stage(nbuf);
            compute(buf);
            buf = buf == 2 ? 0 : buf + 1;
            nbuf = nbuf == 2 ? 0 : nbuf + 1;
        }
        __syncthreads();
    } else {
        stage(0);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < nk) stage(cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    }

    conv_epilogue<DT, WAVES_M, WAVES_N, PAR, EPI, OUT_F32>(a, acc, smem, t, wm, wn, frow, fk, p0, c0, pblk, Mloc, gH, gW, cy, cx);
}

// ================================================================================================
// v3: software-pipelined variant of the glds kernel (same tiles, same LDS image, same epilogue).
// What the ISA of v2 showed (rocprofv3 PMC: MFMA busy 32 % of SIMD cycles on the dominant layer; L2 latency 290 cycles, TLB
// misses nil): per K step a wave ran [~150 branchy address instructions + 9 DMA] -> [9 ds_read, wait] -> [20 MFMA] ->
// [9 ds_read, wait] -> [20 MFMA] -> [vmcnt(0), barrier], i.e. nothing of its own overlapped the MFMAs.  Here:
//   * operands are fetched with buffer_load_dwordx4 ... lds through buffer resources: one 32-bit byte offset per piece,
//     out-of-range offsets (padding taps, rows beyond Cout, steps beyond K) return zeros -> no zero page, no 64-bit address
//     math, no branches: the whole K step is ONE basic block;
//   * K cursors advance branch-free (needs Cs >= 64: at most one tap boundary per 64-wide step);
//   * the step is split in two phases around the single barrier, with the fragments double-buffered in registers:
//       A: ds_read fragments (ks, k32 #1) || MFMA on fragments (ks, k32 #0)
//          s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier     -- tile ks+1 complete, nobody reads buffer `cur` any more
//       B: DMA of tile ks+2 into `cur`  ||  ds_read fragments (ks+1, k32 #0)  ||  MFMA on fragments (ks, k32 #1)
//     sched_group_barrier pins the interleave (1 MFMA : 1 DS read : 1 DMA) so the memory instructions issue in the
//     shadow of the 16-cycle MFMAs.
// ================================================================================================
__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t r, int voff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// ABL (timing experiments only, results are wrong): 1 = no DMA in the K loop, 2 = no fragment reads, 4 = no MFMA (bits combine),
// 8 = no X (pixel) DMA, 16 = no W (weight) DMA
template <int DT, int WAVES_M, int WAVES_N, int MODE, int EPI, bool OUT_F32, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_pipe_kernel(const ConvArgs a) {
    constexpr bool DGRAD = MODE != 0;
    constexpr bool PAR = MODE == 2;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BP = 64 * WAVES_M;
    constexpr int BC = 80 * WAVES_N;
    constexpr int XI = BP / (8 * NW);
    constexpr int WI_TOTAL = BC / 8;
    constexpr int WI = (WI_TOTAL + NW - 1) / NW;
    constexpr int NV = XI + WI;  // DMA instructions per wave per K step
    constexpr int NS = NW == 8 ? 3 : 2;  // LDS stages: the 8-wave tile owns the CU (3 x 52 KiB) and prefetches two K steps ahead
    constexpr int STAGE = (BP + BC) * ROW_BYTES;
    constexpr int SENT = (int)0xfffffff0u;  // byte offset beyond every buffer -> the load returns zeros
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int cls = 0;
    if (PAR) {
        if (a.par_interleave) {  // see conv_igemm_glds_kernel
            cls = L & 3;
            L >>= 2;
        } else {
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (L >= a.pc[q].blk0) cls = q;
            L -= a.pc[cls].blk0;
        }
    }
    const int cy = PAR ? a.pc[cls].cy : 0, cx = PAR ? a.pc[cls].cx : 0;
    const int kh0 = PAR ? a.pc[cls].kh0 : 0, kw0 = PAR ? a.pc[cls].kw0 : 0;
    const int tapsH = PAR ? a.pc[cls].nh : a.KH;
    const int tapsW = PAR ? a.pc[cls].nw : a.KW;
    const int ntap = tapsH * tapsW;
    const int gH = PAR ? a.pc[cls].Hc : a.Hd, gW = PAR ? a.pc[cls].Wc : a.Wd;
    const int Mloc = PAR ? a.pc[cls].M : a.M;
    const int nk = PAR ? a.pc[cls].nk : a.nk;
    const int cblk = L % a.n_cblk;
    const int pblk = L / a.n_cblk;
    const int p0 = pblk * BP;
    const int c0 = cblk * BC;

    const int slot = lane & 7;
    const int lr = lane >> 3;
    const int ldB = a.src_ld * 2;
    // ---- X rows owned by this thread (byte offsets of tap (0,0), validity bit per tap) -----------------------------------
    int rowoffB[XI];
    unsigned tapmask[XI];
    {
        int p = p0 + wave * (8 * XI) + lr;
        const int hw = gH * gW;
        int n = p / hw;
        int rem = p - n * hw;
        int py = rem / gW;
        int px = rem - py * gW;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            rowoffB[j] = 0;
            tapmask[j] = 0u;
            if (p < Mloc) {
                int oy, ox;
                unsigned rb = 0u, cb = 0u;
                if (PAR) {
                    oy = (2 * py + cy + a.pad - kh0) >> 1;
                    ox = (2 * px + cx + a.pad - kw0) >> 1;
                } else if (DGRAD) {  // stride 1
                    oy = py + a.pad;
                    ox = px + a.pad;
                } else {
                    oy = py * a.stride - a.pad;
                    ox = px * a.stride - a.pad;
                }
                for (int k = 0; k < tapsH; ++k)
                    if ((unsigned)(DGRAD ? oy - k : oy + k) < (unsigned)a.Hs) rb |= 1u << k;
                for (int k = 0; k < tapsW; ++k)
                    if ((unsigned)(DGRAD ? ox - k : ox + k) < (unsigned)a.Ws) cb |= 1u << k;
                unsigned m = 0u;
                for (int k = 0; k < tapsH; ++k)
                    if ((rb >> k) & 1u) m |= cb << (k * tapsW);
                tapmask[j] = m;
                rowoffB[j] = (((n * a.Hs + oy) * a.Ws + ox) * a.src_ld + a.src_coff) * 2;
            }
            p += 8;
            px += 8;
            while (px >= gW) {
                px -= gW;
                if (++py == gH) {
                    py = 0;
                    ++n;
                }
            }
        }
    }
    // ---- two K cursors (even / odd row groups differ in bit 2 of the swizzle); Cs >= 64 -> they start inside tap 0 ------
    const int sw0 = ((wave * (8 * XI) + lr) >> 1) & 7;
    const int CsB = a.Cs * 2;
    const int sgn = DGRAD ? -1 : 1;
    const int dTap = sgn * ldB - CsB;                               // cursor crosses into the next tap of the same kernel row
    const int dRow = sgn * (a.Ws - (tapsW - 1)) * ldB - CsB;        // ... into the first tap of the next kernel row
    const int wTap = PAR ? CsB : 0;                                 // parity mode: packed-row offset skips one tap
    const int wRow = PAR ? ((2 * a.KW - 2 * (tapsW - 1)) * a.Cs - a.Cs) * 2 : 0;
    int kc[2], ktap[2], kkw[2], toffB[2], wkB[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int kv = slot ^ sw0 ^ (c ? 4 : 0);
        kc[c] = kv * 8;
        ktap[c] = 0;
        kkw[c] = 0;
        toffB[c] = kv * 16;
        wkB[c] = ((kh0 * a.KW + kw0) * a.Cs + kv * 8) * 2;
    }
    // ---- W pieces: row group wi = wave + NW*j (a wave without a last group repeats its previous one: same data, same place)
    int woffB[WI];
    int wdst[WI];
    int wcur[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        int wi = wave + NW * j;
        if (wi >= WI_TOTAL) wi -= NW;
        const int r = 8 * wi + lr;
        const int co = c0 + r;
        const int kv = slot ^ ((r >> 1) & 7);
        woffB[j] = co * a.Kpad * 2 + (PAR ? 0 : kv * 16);  // rows >= Cd lie beyond w_bytes -> zeros
        wdst[j] = 8 * wi * ROW_BYTES;
        wcur[j] = (wi ^ (wave * XI)) & 1;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);  // empty: every load is zero
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0, 0x00020000);

    // one DMA instruction of the K step the cursors point at: pieces 0..XI-1 are X row groups, XI..NV-1 W row groups
    auto dma_piece = [&](int idx, int buf, bool live) {
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* wsm = xs + BP * ROW_BYTES;
        if (idx < XI) {
            const int j = idx, c = j & 1;
            const bool ok = (tapmask[j] >> ktap[c]) & 1u;
            bufload_lds16(live ? rs_x : rs_x0, ok ? rowoffB[j] + toffB[c] : SENT, xs + (wave * (8 * XI) + 8 * j) * ROW_BYTES);
        } else {
            const int j = idx - XI;
            int voff;
            if (PAR) {
                const int c = wcur[j];
                voff = ktap[c] < ntap ? woffB[j] + wkB[c] : SENT;
            } else {
                voff = woffB[j];
                woffB[j] += 2 * BK;
            }
            bufload_lds16(live ? rs_w : rs_w0, voff, wsm + wdst[j]);
        }
    };
    auto advance = [&](int c) {
        kc[c] += BK;
        const bool wrap = kc[c] >= a.Cs;
        kc[c] -= wrap ? a.Cs : 0;
        ktap[c] += wrap ? 1 : 0;
        kkw[c] += wrap ? 1 : 0;
        const bool roww = kkw[c] == tapsW;
        kkw[c] = roww ? 0 : kkw[c];
        toffB[c] += 2 * BK + (wrap ? (roww ? dRow : dTap) : 0);
        if (PAR) wkB[c] += 2 * BK + (wrap ? (roww ? wRow : wTap) : 0);
    };
    auto stage = [&](int buf, bool live) {
#pragma unroll
        for (int i = 0; i < NV; ++i) dma_piece(i, buf, live);
        advance(0);
        advance(1);
    };

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = lane >> 4;

    // fragment r of a k32 half: r < 5 -> weight rows (A operand), else pixel rows (B operand)
    auto frag1 = [&](int buf, int kk, int r, u32x4 (&af)[5], u32x4 (&bf)[4]) {
        const unsigned char* Xs = smem + buf * STAGE;
        const unsigned char* Wsm = Xs + BP * ROW_BYTES;
        if (r < 5) af[r] = *reinterpret_cast<const u32x4*>(Wsm + lds_slot(wn * 80 + r * 16 + frow, kk * 4 + fk));
        else bf[r - 5] = *reinterpret_cast<const u32x4*>(Xs + lds_slot(wm * 64 + (r - 5) * 16 + frow, kk * 4 + fk));
    };
    // read order: the first MFMAs of the next phase need A row 0 and all four B rows
    constexpr int RORD[9] = {0, 5, 6, 7, 8, 1, 2, 3, 4};

    u32x4 a0[5], b0[4], a1[5], b1[4];
    // all NS stages are requested before the first wait: only tile 0's latency is exposed
    stage(0, true);
    stage(1, nk > 1);
    if (NS == 3) stage(2, nk > 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * NV) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 9; ++r) frag1(0, 0, r, a0, b0);
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const int nxt = cur + 1 == NS ? 0 : cur + 1;
        const bool live = ks + NS < nk;
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase A: MFMA on (ks, k32 #0) with the fragment reads of (ks, k32 #1) in their shadow ----
#pragma unroll
        for (int idx = 0; idx < 20; ++idx) {
            if (!(ABL & 4)) mfma_inplace<DT>(a0[idx >> 2], b0[idx & 3], acc[idx >> 2][idx & 3]);
            if (!(ABL & 2) && (idx & 1) == 0 && (idx >> 1) < 9) frag1(cur, 1, RORD[idx >> 1], a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile ks+1 landed for every wave (with 3 stages the NV pieces of tile ks+2 may stay in flight); buffer `cur` is free
        if (NS == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NV) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: MFMA on (ks, k32 #1); DMA of tile ks+NS and fragment reads of (ks+1, k32 #0) in their shadow ----
#pragma unroll
        for (int idx = 0; idx < 20; ++idx) {
            if (!(ABL & 4)) mfma_inplace<DT>(a1[idx >> 2], b1[idx & 3], acc[idx >> 2][idx & 3]);
            if (!(ABL & 1) && idx < NV && !((ABL & 8) && idx < XI) && !((ABL & 16) && idx >= XI)) dma_piece(idx, cur, live);
            if (idx == NV) advance(0);
            if (idx == NV + 1) advance(1);
            if (!(ABL & 2) && idx >= 20 - 9 - 1 && idx < 20 - 1) frag1(nxt, 0, RORD[idx - (20 - 9 - 1)], a0, b0);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs are opaque to the hazard recogniser: let the last ones retire before acc is read
    __syncthreads();  // drains the (empty) trailing DMA before LDS is reused below

    conv_epilogue<DT, WAVES_M, WAVES_N, PAR, EPI, OUT_F32>(a, acc, smem, t, wm, wn, frow, fk, p0, c0, pblk, Mloc, gH, gW, cy, cx);
}

// ------------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> [rows][Kpad] (bf16/f16), K = (kh, kw, c)
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int O, int O_pad, int I, int KH, int KW,
                                   int transpose, const float* __restrict__ row_scale, int dtype, int rows, int Kpad) {
    const int64_t total = (int64_t)rows * Kpad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / Kpad);
        const int k = (int)(idx - (int64_t)row * Kpad);
        const int C = transpose ? O_pad : I;  // reduction channels
        float v = 0.f;
        if (k < KH * KW * C) {
            const int tap = k / C, c = k - tap * C;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int o = transpose ? c : row, i = transpose ? row : c;
            if (o < O) {
                v = w[(((int64_t)o * I + i) * KH + kh) * KW + kw];
                if (row_scale) v *= row_scale[o];
            }
        }
        out[idx] = dtype == CDET_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
    }
}

static inline int kpad_of(int K) { return (K + BK - 1) / BK * BK; }

// every pack of a model in one launch: workgroup -> (item, 32 x 32 tile of (o, i)); the OIHW tile is read contiguously
// (32 rows of 32*taps floats), transposed through LDS and written in both operand layouts in 32-element runs
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const cdet_pack_item* __restrict__ items, int n, int dtype) {
    __shared__ int s_it;
    __shared__ float tile[32][32 * 9 + 1];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (items[mid].first_block <= (int)blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        s_it = lo;
    }
    __syncthreads();
    const cdet_pack_item p = items[s_it];
    const int taps = p.kh * p.kw;
    const int tiles_i = (p.I + 31) / 32;
    const int lb = (int)blockIdx.x - p.first_block;
    const int o0 = (lb / tiles_i) * 32, i0 = (lb % tiles_i) * 32;
    const int ni = min(32, p.I - i0), no = min(32, p.O - o0);
    const int run = ni * taps;  // contiguous floats per output channel of this tile
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int o = e / run, r = e - o * run;
        if (o < no) tile[o][r] = p.w_oihw[((int64_t)(o0 + o) * p.I + i0) * taps + r];
    }
    __syncthreads();
    const int Kf = (taps * p.I + BK - 1) / BK * BK, Kt = (taps * p.O_pad + BK - 1) / BK * BK;
    uint16_t* wf = reinterpret_cast<uint16_t*>(p.w_fwd);
    uint16_t* wt = reinterpret_cast<uint16_t*>(p.w_dgrad);
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5;  // 32 lanes along the contiguous axis, 8 groups over (row, tap)
    for (int rt = g; rt < 32 * taps; rt += 8) {
        const int row = rt / taps, tap = rt - row * taps;
        // forward operand: row = o, run over i
        if (row < no && l < ni) {
            const float v = tile[row][l * taps + tap];
            wf[(int64_t)(o0 + row) * Kf + tap * p.I + i0 + l] = dtype == CDET_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
        }
        // DGRAD operand: row = i, run over o
        if (wt != nullptr && row < ni && l < no) {
            const float v = tile[l][row * taps + tap];
            wt[(int64_t)(i0 + row) * Kt + tap * p.O_pad + o0 + l] = dtype == CDET_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
        }
    }
}

static int conv_impl() {
    static int impl = -1;
    if (impl < 0) {
        impl = tune_env("CDET_CONV_IMPL", 3);  // 1 = register-staged v1, 2 = global_load_lds v2, 3 = v2 tiles + software-pipelined K loop (v3)
    }
    return impl;
}

template <int DT, int WM, int WN, int DG, int EPI, bool F32>
static void launch_glds(const ConvArgs& a, hipStream_t s) {
    constexpr int BP = 64 * WM, BC = 80 * WN, NW = WM * WN;
    const size_t lds = (size_t)(NW == 8 ? 3 : 2) * (BP + BC) * ROW_BYTES;
    const int nblocks = DG == 2 ? a.pc[3].blk0 + a.pc[3].nblk : a.n_pblk * a.n_cblk;
    // pipelined variant: one tap boundary per K step at most (Cs >= 64), buffer-resource bounds fit 32 bits; stride-2 dgrad only as parity classes
    const bool pipe = conv_impl() >= 3 && a.Cs >= BK && a.x_bytes != 0 && !(DG == 1 && a.stride != 1);
    if (pipe) {
        static bool attr3 = false;
        if (!attr3) {
            (void)hipFuncSetAttribute((const void*)conv_igemm_pipe_kernel<DT, WM, WN, DG, EPI, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr3 = true;
        }
        if constexpr (DT == CDET_BF16 && WM == 2 && WN == 2 && DG == 0 && EPI == EPI_RAW && !F32) {
            static int abl = -1;
            if (abl < 0) {
                abl = tune_env("CDET_CONV_ABLATE", 0);
            }
            if (abl) {
#define CDET_ABL_CASE(N)                                                                                                                       \
    case N:                                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)conv_igemm_pipe_kernel<DT, WM, WN, DG, EPI, F32, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                                                   \
        hipLaunchKernelGGL((conv_igemm_pipe_kernel<DT, WM, WN, DG, EPI, F32, N>), dim3(nblocks), dim3(64 * NW), lds, s, a);                      \
        return;
                switch (abl) {
                    CDET_ABL_CASE(1) CDET_ABL_CASE(2) CDET_ABL_CASE(3) CDET_ABL_CASE(4) CDET_ABL_CASE(5) CDET_ABL_CASE(6) CDET_ABL_CASE(7)
                    CDET_ABL_CASE(8) CDET_ABL_CASE(16)
                    default: break;
                }
#undef CDET_ABL_CASE
            }
        }
        hipLaunchKernelGGL((conv_igemm_pipe_kernel<DT, WM, WN, DG, EPI, F32>), dim3(nblocks), dim3(64 * NW), lds, s, a);
        return;
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<DT, WM, WN, DG, EPI, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((conv_igemm_glds_kernel<DT, WM, WN, DG, EPI, F32>), dim3(nblocks), dim3(64 * NW), lds, s, a);
}

template <int DT, int WM, int WN>
static void launch_glds_variant(const ConvArgs& a, bool dgrad, hipStream_t s) {
    constexpr int NW = WM * WN;
    const bool f32out = a.out_dtype == CDET_F32;
    const bool full = a.scale || a.bias || a.res || a.act != CDET_ACT_NONE;
    if (dgrad && a.stride == 2) {  // parity-decomposed stride-2 data gradient
        if (full) { if (f32out) launch_glds<DT, WM, WN, 2, EPI_FULL, true>(a, s); else launch_glds<DT, WM, WN, 2, EPI_FULL, false>(a, s); }
        else      { if (f32out) launch_glds<DT, WM, WN, 2, EPI_RAW, true>(a, s);  else launch_glds<DT, WM, WN, 2, EPI_RAW, false>(a, s); }
    } else if (dgrad) {
        if (full) { if (f32out) launch_glds<DT, WM, WN, 1, EPI_FULL, true>(a, s); else launch_glds<DT, WM, WN, 1, EPI_FULL, false>(a, s); }
        else      { if (f32out) launch_glds<DT, WM, WN, 1, EPI_RAW, true>(a, s);  else launch_glds<DT, WM, WN, 1, EPI_RAW, false>(a, s); }
    } else {
        if (full) { if (f32out) launch_glds<DT, WM, WN, 0, EPI_FULL, true>(a, s); else launch_glds<DT, WM, WN, 0, EPI_FULL, false>(a, s); }
        else      { if (f32out) launch_glds<DT, WM, WN, 0, EPI_RAW, true>(a, s);  else launch_glds<DT, WM, WN, 0, EPI_RAW, false>(a, s); }
    }
}

// tile configuration: 0 = <4,1> 256 px x 80 couts (Cout <= 80), 1 = <2,2> 128 x 160, 2 = <4,2> 256 x 160 with the deep pipeline
static int pick_cfg(int Cd, int64_t M) {
    if (Cd <= 80) return 0;
    static int deep = -1;
    if (deep < 0) {
        deep = tune_env("CDET_CONV_DEEP", 0);  // measured on MI355X: 0.129 vs 0.127 ms (40x40 320->320), 0.163 vs 0.144 ms (80x80 160->160): off by default
    }
    return (deep && conv_impl() >= 2 && M >= 4096) ? 2 : 1;
}
static inline int cfg_bp(int cfg) { return cfg == 1 ? 128 : (cfg == 0 ? 192 : 256); }  // cfg 0: 3 waves x 64 px (2 x 69 KB LDS -> 2 WGs per CU)
static inline int cfg_bc(int cfg) { return (cfg == 0 || cfg == 3) ? 80 : 160; }

template <int DT>
static int launch_conv(const ConvArgs& a, int cfg, bool dgrad, hipStream_t s) {
    // v2 uses 32-bit element offsets for the im2col gather
    const bool fits32 = (int64_t)a.N * a.Hs * a.Ws * a.src_ld < (1ll << 31);
    const bool f32out = a.out_dtype == CDET_F32;
    const bool same16 = (a.out_dtype == CDET_BF16 && DT == CDET_BF16) || (a.out_dtype == CDET_F16 && DT == CDET_F16);
    if (cfg != 3 && conv_impl() >= 2 && fits32 && (f32out || same16)) {
        if (cfg == 0) launch_glds_variant<DT, 3, 1>(a, dgrad, s);
        else if (cfg == 1) launch_glds_variant<DT, 2, 2>(a, dgrad, s);
        else launch_glds_variant<DT, 4, 2>(a, dgrad, s);
    } else {
        dim3 grid(a.n_pblk * a.n_cblk), block(256);
        if (cfg == 0 || cfg == 3) {  // legacy kernels use 256-pixel tiles (cfg_of maps to cfg 3 when the glds family cannot run)
            const size_t lds = (size_t)(256 + 80) * ROW_BYTES;
            if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<DT, 4, 1, true>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<DT, 4, 1, false>), grid, block, lds, s, a);
        } else {
            const size_t lds = (size_t)(128 + 160) * ROW_BYTES;
            if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<DT, 2, 2, true>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<DT, 2, 2, false>), grid, block, lds, s, a);
        }
    }
    CDET_LAUNCH_CHECK();
    return 0;
}

static inline int cfg_of(const cdet_conv_desc* d) {
    int cfg = pick_cfg(d->Cd, (int64_t)d->N * d->Hd * d->Wd);
    const bool fits32 = (int64_t)d->N * d->Hs * d->Ws * d->src_ld < (1ll << 31);
    const bool f32out = d->out_dtype == CDET_F32;
    const bool same16 = d->out_dtype == d->dtype;
    if (cfg == 2 && !(fits32 && (f32out || same16))) cfg = 1;  // the deep variant only exists in the glds family
    if (cfg == 0 && !(conv_impl() >= 2 && fits32 && (f32out || same16))) cfg = 3;  // legacy register-staged <4,1> kernel: 256-px tiles
    return cfg;
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_conv2d_stat_blocks(const cdet_conv_desc* d) {
    const int64_t M = (int64_t)d->N * d->Hd * d->Wd;
    return div_up(M, cfg_bp(cfg_of(d)));
}

extern "C" int64_t cdet_packed_weight_elems(int32_t O, int32_t I, int32_t kh, int32_t kw, int32_t transpose) {
    // O is the PADDED output-channel count here
    const int rows = transpose ? I : O, C = transpose ? O : I;
    return (int64_t)rows * kpad_of(kh * kw * C);
}

extern "C" int cdet_pack_weight(const float* w, void* out, int32_t O, int32_t O_pad, int32_t I, int32_t kh, int32_t kw, int32_t transpose,
                                const float* row_scale, int32_t dtype, void* stream) {
    CDET_CHECK_ARG(w && out && O > 0 && O_pad >= O && I > 0 && kh > 0 && kw > 0, "cdet_pack_weight: bad arguments");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_pack_weight: dtype must be bf16/f16");
    const int rows = transpose ? I : O_pad, C = transpose ? O_pad : I;
    const int Kpad = kpad_of(kh * kw * C);
    const int64_t total = (int64_t)rows * Kpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)out, O, O_pad, I, kh, kw,
                       transpose, row_scale, dtype, rows, Kpad);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_pack_weights_batched(const cdet_pack_item* items, int32_t n_items, int32_t n_blocks_total, int32_t dtype, void* stream) {
    CDET_CHECK_ARG(items && n_items > 0 && n_blocks_total >= n_items, "cdet_pack_weights_batched: bad arguments (kh*kw <= 9 is the caller's duty)");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_pack_weights_batched: dtype must be bf16/f16");
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(n_blocks_total), dim3(256), 0, (hipStream_t)stream, items, n_items, dtype);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_conv2d(const cdet_conv_desc* d, const void* x, const void* w, const float* scale, const float* bias,
                           const void* residual, void* y, float* stats, void* stream) {
    CDET_CHECK_ARG(d && x && w && y, "cdet_conv2d: null pointer");
    CDET_CHECK_ARG(d->dtype == CDET_BF16 || d->dtype == CDET_F16, "cdet_conv2d: dtype must be bf16/f16 (got %d)", d->dtype);
    CDET_CHECK_ARG(d->Cs % 8 == 0 && d->src_ld % 8 == 0 && d->src_coff % 8 == 0,
                   "cdet_conv2d: source channels/ld/coff must be multiples of 8 (Cs=%d ld=%d coff=%d)", d->Cs, d->src_ld, d->src_coff);
    CDET_CHECK_ARG(d->Cd % 4 == 0 && d->dst_ld % 4 == 0 && d->dst_coff % 4 == 0,
                   "cdet_conv2d: destination channels/ld/coff must be multiples of 4 (Cd=%d ld=%d coff=%d)", d->Cd, d->dst_ld, d->dst_coff);
    CDET_CHECK_ARG(!residual || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0), "cdet_conv2d: residual ld/coff must be multiples of 4");
    CDET_CHECK_ARG(d->stride == 1 || d->stride == 2, "cdet_conv2d: stride must be 1 or 2");
    CDET_CHECK_ARG(!d->accumulate || d->out_dtype == CDET_F32, "cdet_conv2d: accumulate needs an fp32 destination");
    CDET_CHECK_ARG((int64_t)d->N * d->Hd * d->Wd < (1ll << 31) && (int64_t)d->N * d->Hs * d->Ws < (1ll << 31), "cdet_conv2d: too many pixels");
    ConvArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w; a.scale = scale; a.bias = bias; a.res = (const uint16_t*)residual;
    a.y = y; a.stats = stats;
    {  // timing experiment only: CDET_CONV_NOSTATS=1 drops the BN partial statistics (results of the BN that follows are wrong)
        static int nostats = -1;
        if (nostats < 0) nostats = tune_env("CDET_CONV_NOSTATS", 0);
        if (nostats) a.stats = nullptr;
    }
    a.N = d->N; a.Hs = d->Hs; a.Ws = d->Ws; a.Cs = d->Cs; a.Hd = d->Hd; a.Wd = d->Wd; a.Cd = d->Cd;
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dst_ld = d->dst_ld; a.dst_coff = d->dst_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.Kpad = kpad_of(d->kh * d->kw * d->Cs); a.nk = a.Kpad / BK;
    a.M = d->N * d->Hd * d->Wd;
    a.act = d->act; a.out_dtype = d->out_dtype; a.accumulate = d->accumulate;
    const int cfg = cfg_of(d);
    a.n_pblk = div_up(a.M, cfg_bp(cfg));
    a.n_cblk = div_up(d->Cd, cfg_bc(cfg));
    const bool dg = d->mode == CDET_CONV_DGRAD;
    {
        const int64_t xb = (int64_t)d->N * d->Hs * d->Ws * d->src_ld * 2, wb = (int64_t)d->Cd * a.Kpad * 2;
        const bool ok = xb < 0xfffffff0ll - 256 && wb < 0xfffffff0ll - 65536;
        a.x_bytes = ok ? (unsigned)xb : 0u;  // 0 -> the pipelined kernel is not used
        a.w_bytes = ok ? (unsigned)wb : 0u;
    }
    for (int q = 0; q < 4; ++q) a.pc[q] = ConvArgs::ParClass{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    a.par_interleave = 0;
    if (dg && d->stride == 2) {
        // parity classes of the dX pixels, heaviest (most taps) first so the short ones fill the tail
        int order[4] = {0, 1, 2, 3}, taps[4];
        ConvArgs::ParClass c4[4];
        for (int q = 0; q < 4; ++q) {
            ConvArgs::ParClass& c = c4[q];
            c.cy = q >> 1; c.cx = q & 1;
            c.kh0 = (c.cy + d->pad) & 1; c.kw0 = (c.cx + d->pad) & 1;
            c.nh = c.kh0 < d->kh ? (d->kh - c.kh0 + 1) / 2 : 0;
            c.nw = c.kw0 < d->kw ? (d->kw - c.kw0 + 1) / 2 : 0;
            if (c.nh == 0 || c.nw == 0) c.nh = c.nw = 0;
            c.Hc = (d->Hd - c.cy + 1) / 2; c.Wc = (d->Wd - c.cx + 1) / 2;
            c.M = d->N * c.Hc * c.Wc;
            c.nk = div_up(c.nh * c.nw * d->Cs, BK);
            taps[q] = c.nh * c.nw;
        }
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
                if (taps[order[j]] > taps[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
        int blk = 0;
        for (int i = 0; i < 4; ++i) {
            a.pc[i] = c4[order[i]];
            a.pc[i].blk0 = blk;
            a.pc[i].nblk = div_up(a.pc[i].M, cfg_bp(cfg)) * a.n_cblk;
            blk += a.pc[i].nblk;
        }
        a.par_interleave = tune_env("CDET_PAR_INTERLEAVE", 1) && a.pc[0].nblk == a.pc[1].nblk && a.pc[1].nblk == a.pc[2].nblk &&
                           a.pc[2].nblk == a.pc[3].nblk && a.pc[3].nblk > 0;
    }
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == CDET_BF16) return launch_conv<CDET_BF16>(a, cfg, dg, s);
    return launch_conv<CDET_F16>(a, cfg, dg, s);
}
