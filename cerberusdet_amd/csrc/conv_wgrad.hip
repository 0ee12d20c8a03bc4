// Weight gradient of the NHWC convolution on MFMA (gfx950).
//
//   dW[co][k] = sum_p dY[p][co] * Xg[p][k],   k = (kh, kw, ci) flattened, p = (n, oy, ox), Xg = im2col gather of X.
//
// The reduction runs over PIXELS, but NHWC keeps channels contiguous, so both operand tiles arrive in LDS as
// [pixel][channel] rows -- the transpose of what an MFMA fragment wants (8 consecutive reduction indices per lane).
// gfx950's LDS transpose read (ds_read_b64_tr_b16) does that transposition for free on the way to the registers:
// a 16-lane group reads a [4 pixels][16 channels] block and each lane receives one channel's 4 pixels.
// Two such reads give a lane 8 reduction indices {4q..4q+3, 16+4q..16+4q+3} (q = lane>>4); A and B fragments use the
// same index set, so the dot products are consistent.
//
// Tiling: A = X tile (rows = k columns, 64 per wave), B = dY tile (cols = couts, 80 per wave); block = 128 k-cols x
// 160 couts (<2,2> waves) or 256 x 80 (<4,1>); 64 pixels per step; the pixel range is split over `S` blocks (split-K),
// partials go to a workspace [S][Cout_pad][Kpad] and a second kernel reduces them in fixed order into fp32 OIHW.
#include <stdlib.h>

#include "common.h"

namespace cdet {

struct WgradArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* ws;
    int N, Hs, Ws, Cs, Hd, Wd, Cd;
    int KH, KW, stride, pad;
    int src_ld, src_coff, dy_ld, dy_coff;
    int Ktot, Kp;         // taps*Cs and its padding to the k-tile
    int Cd_pad;
    int P;                // N*Hd*Wd
    int chunk;            // pixels per split (multiple of 64)
    int n_kblk, n_cblk, S;
};

constexpr int WKP = 64;  // pixels per step

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, r);
}

template <int DT> struct Mfma2;
template <> struct Mfma2<CDET_BF16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma2<CDET_F16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ABL (profiling only, CDET_WGRAD_ABLATE): 1 = no global loads in the loop, 2 = no LDS fragment reads, 3 = no MFMA
template <int DT, int WAVES_K, int WAVES_C, int ABL = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int BKC = 64 * WAVES_K;          // k columns per block
    constexpr int BCO = 80 * WAVES_C;          // couts per block
    constexpr int XROW = BKC * 2 + 32;         // LDS row bytes (+32 B: rows land on different banks)
    constexpr int YROW = BCO * 2 + ((BCO * 2) % 256 == 0 ? 32 : 0);
    constexpr int XV = BKC / 8;                // vectors per X row
    constexpr int YV = BCO / 8;
    constexpr int XR = WKP * XV / 256;         // X vectors per thread
    constexpr int YTOT = WKP * YV;             // Y vectors per step
    constexpr int YR = (YTOT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Ys = smem + WKP * XROW;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / WAVES_C, wc = wave % WAVES_C;
    // Workgroup -> (pixel split s, tile) so that all tiles of one pixel chunk run on ONE XCD at the same time: the chunk's X and
    // dY rows (a few MB) are then served by that XCD's 4 MiB L2 instead of being re-fetched once per tile (the first mapping,
    // s fastest, made this kernel HBM-bound: 23 k-blocks x 2 cout-blocks re-read every chunk).
    int s, tile;
    {
        const int ntiles = a.n_kblk * a.n_cblk;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        if (a.S >= 8) {  // any S >= 8: the grid is padded to a multiple of 8 splits, surplus workgroups leave at once
            s = xcd + 8 * (j / ntiles);
            tile = j % ntiles;
            if (s >= a.S) return;
        } else {         // S in {1, 2, 4}: 8/S XCDs share a chunk
            const int g = 8 / a.S;
            s = xcd % a.S;
            tile = xcd / a.S + g * j;
            if (tile >= ntiles) return;
        }
    }
    const int cblk = tile % a.n_cblk;
    const int kblk = tile / a.n_cblk;
    const int k0 = kblk * BKC, c0 = cblk * BCO;
    const int p_begin = s * a.chunk;
    const int p_end = min(p_begin + a.chunk, a.P);
    if (p_begin >= p_end) {  // empty split (rounding): still owes zeros to its workspace slab
        float* wz = a.ws + (int64_t)s * a.Cd_pad * a.Kp;
        for (int e = threadIdx.x; e < BCO * BKC / 4; e += 256) {
            const int co = c0 + e / (BKC / 4), k = k0 + (e % (BKC / 4)) * 4;
            if (co < a.Cd_pad && k < a.Kp) *reinterpret_cast<f32x4*>(wz + (int64_t)co * a.Kp + k) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // ---- X gather bookkeeping: this thread always loads k-vector `xkv` of rows xrow0 + 16*i (XV == 16 for BKC 128) ----
    const int xkv = t % XV;
    const int xrow0 = t / XV;
    constexpr int XRSTEP = 256 / XV;
    const int kcol = k0 + xkv * 8;
    const bool k_ok = kcol < a.Ktot;
    int tap = 0, ci = 0, kh = 0, kw = 0;
    if (k_ok) {
        tap = kcol / a.Cs;
        ci = kcol - tap * a.Cs;
        kh = tap / a.KW;
        kw = tap - kh * a.KW;
    }
    // running (n, oy, ox) of each of this thread's rows; advanced by 64 pixels per step
    int rn[XR], roy[XR], rox[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int p = p_begin + xrow0 + XRSTEP * i;
        const int hw = a.Hd * a.Wd;
        const int n = p / hw, rem = p - n * hw;
        rn[i] = n;
        roy[i] = rem / a.Wd;
        rox[i] = rem - roy[i] * a.Wd;
    }

    // dY vectors of this thread: fixed (row, channel-vector); the pointer just advances by 64 pixels per step
    const uint16_t* yptr[YR];
    int yrow[YR];
    bool ycok[YR];
#pragma unroll
    for (int i = 0; i < YR; ++i) {
        const int v = t + 256 * i;
        const int row = v / YV, cv = v - row * YV;
        yrow[i] = row;
        ycok[i] = v < YTOT && (c0 + cv * 8) < a.Cd;
        yptr[i] = a.dy + (int64_t)(p_begin + row) * a.dy_ld + a.dy_coff + c0 + cv * 8;
    }
    const int64_t ystep = (int64_t)WKP * a.dy_ld;
    const int xconst = a.src_coff + ci;  // 32-bit element offsets (host checks N*Hs*Ws*src_ld < 2^31)
    const int sy_off = kh - a.pad, sx_off = kw - a.pad;

    u32x4 xreg[XR], yreg[YR];
    auto load_global = [&](int pbase) {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int sy = roy[i] * a.stride + sy_off, sx = rox[i] * a.stride + sx_off;
            const bool ok = k_ok && (pbase + xrow0 + XRSTEP * i) < p_end && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(a.x + (((rn[i] * a.Hs + sy) * a.Ws + sx) * a.src_ld + xconst));
            xreg[i] = v;
            rox[i] += WKP;  // advance by WKP pixels
            while (rox[i] >= a.Wd) {
                rox[i] -= a.Wd;
                if (++roy[i] == a.Hd) {
                    roy[i] = 0;
                    ++rn[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < YR; ++i) {
            u32x4 val = {0u, 0u, 0u, 0u};
            if (ycok[i] && (pbase + yrow[i]) < p_end) val = *reinterpret_cast<const u32x4*>(yptr[i]);
            yreg[i] = val;
            yptr[i] += ystep;
        }
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, li = lane & 15;
    // byte offset of this lane's 8-byte chunk inside a [4 rows][16 ch] block: row (li>>2), channels (li&3)*4..+3
    const int x_lane_off = (4 * q + (li >> 2)) * XROW + (wk * 64 + (li & 3) * 4) * 2;
    const int y_lane_off = (4 * q + (li >> 2)) * YROW + (wc * 80 + (li & 3) * 4) * 2;

    load_global(p_begin);
    for (int pb = p_begin; pb < p_end; pb += WKP) {
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<u32x4*>(Xs + (xrow0 + XRSTEP * i) * XROW + xkv * 16) = xreg[i];
#pragma unroll
        for (int i = 0; i < YR; ++i) {
            const int v = t + 256 * i;
            const int row = v / YV, cv = v - row * YV;
            if (v < YTOT) *reinterpret_cast<u32x4*>(Ys + row * YROW + cv * 16) = yreg[i];
        }
        __syncthreads();
        if (ABL != 1 && pb + WKP < p_end) load_global(pb + WKP);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {  // two 32-pixel MFMA steps
            u32x4 af[4], bf[5];
            if (ABL == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = u32x4{(unsigned)pb, 1u, 2u, (unsigned)i};
#pragma unroll
                for (int j = 0; j < 5; ++j) bf[j] = u32x4{(unsigned)pb, 1u, 2u, (unsigned)j};
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char* base = Xs + x_lane_off + kk * 32 * XROW + i * 32;  // 16 channels = 32 B per tile
                const u32x2 lo = tr_read(base), hi = tr_read(base + 16 * XROW);
                af[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const unsigned char* base = Ys + y_lane_off + kk * 32 * YROW + j * 32;
                const u32x2 lo = tr_read(base), hi = tr_read(base + 16 * YROW);
                bf[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            }
            if (ABL == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][0][0] += __uint_as_float(af[i][0] ^ af[i][3]);
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[0][j][1] += __uint_as_float(bf[j][0] ^ bf[j][3]);
            } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = Mfma2<DT>::run(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            }
        }
        __syncthreads();
    }

    // ---- store partial tile: ws[s][co][k], lane holds 4 consecutive k for one co -------------------------------------
    float* wsp = a.ws + (int64_t)s * a.Cd_pad * a.Kp;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int co = c0 + wc * 80 + j * 16 + li;
        if (co >= a.Cd_pad) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + wk * 64 + i * 16 + q * 4;
            if (k < a.Kp) *reinterpret_cast<f32x4*>(wsp + (int64_t)co * a.Kp + k) = acc[i][j];
        }
    }
}

// ================================================================================================
// v3: software-pipelined weight-gradient kernel (same tiles / splits / workspace as above).
//   * operands go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (no VGPR staging, no ds_write pass); out-of-range byte
//     offsets (padding taps, pixels beyond P, channels beyond Cout) return zeros, so the K step has no branches;
//   * the LDS image of a DMA instruction is lane-linear, rows cannot be padded: bank conflicts of the transpose reads are
//     removed by XOR-swizzling 32-byte units with the pixel-row index on the SOURCE side (X: unit ^ (row & 7); dY rows of
//     320 B: unit ^ ((row >> 2) & 1); 160-B dY rows are conflict-free as they are);
//   * two LDS stages, one barrier per 64-pixel step, fragments double-buffered in registers:
//       A: MFMA on (step, pixels 0..31)  || transpose reads of (step, pixels 32..63)
//          s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
//       B: MFMA on (step, pixels 32..63) || DMA of step+2 || transpose reads of (step+1, pixels 0..31)
//   * in-place (asm) MFMAs: the accumulators are touched from two phases per iteration.
// ================================================================================================
template <int DT>
__device__ __forceinline__ void mfma_inplace_w(const u32x4& a, const u32x4& b, f32x4& c) {
    if (DT == CDET_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ void bufload_lds16_w(__amdgpu_buffer_rsrc_t r, int voff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <int DT, int WAVES_K, int WAVES_C>
__global__ __launch_bounds__(256) void conv_wgrad_pipe_kernel(const WgradArgs a, unsigned x_bytes, unsigned y_bytes) {
    constexpr int BKC = 64 * WAVES_K;
    constexpr int BCO = 80 * WAVES_C;
    constexpr int XROW = BKC * 2;              // bytes per pixel row of the X tile (256 / 512)
    constexpr int YROW = BCO * 2;              // 320 / 160
    constexpr int XV = BKC / 8, YV = BCO / 8;  // 16-byte vectors per row
    constexpr int XI = XV / 4;                 // X DMA instructions per wave per step (XV in the block)
    constexpr int RPP = 64 / XV;               // pixel rows per X instruction (4 / 2)
    constexpr int NVAR = 8 / RPP;              // distinct (row & 7) phases among a wave's X instructions
    constexpr int YI = (YV + 3) / 4;           // dY DMA instructions per wave per step (YV in the block)
    constexpr int NV = XI + YI;
    constexpr int STAGE = WKP * (XROW + YROW);
    constexpr bool YSW = BCO == 160;           // dY swizzle needed only for 320-byte rows
    constexpr int SENT = (int)0xfffffff0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wk = wave / WAVES_C, wc = wave % WAVES_C;
    int s, tile;
    {
        const int ntiles = a.n_kblk * a.n_cblk;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        if (a.S >= 8) {
            s = xcd + 8 * (j / ntiles);
            tile = j % ntiles;
            if (s >= a.S) return;
        } else {
            const int g = 8 / a.S;
            s = xcd % a.S;
            tile = xcd / a.S + g * j;
            if (tile >= ntiles) return;
        }
    }
    const int cblk = tile % a.n_cblk;
    const int kblk = tile / a.n_cblk;
    const int k0 = kblk * BKC, c0 = cblk * BCO;
    const int p_begin = s * a.chunk;
    const int p_end = min(p_begin + a.chunk, a.P);
    const int q = lane >> 4, li = lane & 15;
    float* wsp = a.ws + (int64_t)s * a.Cd_pad * a.Kp;
    if (p_begin >= p_end) {  // empty split (rounding): still owes zeros to its workspace slab
        for (int e = threadIdx.x; e < BCO * BKC / 4; e += 256) {
            const int co = c0 + e / (BKC / 4), k = k0 + (e % (BKC / 4)) * 4;
            if (co < a.Cd_pad && k < a.Kp) *reinterpret_cast<f32x4*>(wsp + (int64_t)co * a.Kp + k) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int nsteps = (p_end - p_begin + WKP - 1) / WKP;

    // ---- X instructions of this wave: instruction i covers pixel rows (wave*XI + i)*RPP + lane/XV, physical slot lane%XV ----
    const int xslot = lane % XV, xrl = lane / XV;
    // per swizzle phase: tap offsets (rows / columns / source pixels) and the constant byte offset; vconst < 0 -> k column beyond K
    int vsy[NVAR], vsx[NVAR], vtap[NVAR], vconst[NVAR];
#pragma unroll
    for (int c = 0; c < NVAR; ++c) {
        const int rowbits = ((wave * XI + c) * RPP + xrl) & 7;  // (row & 7) of every instruction i with i % NVAR == c
        const int kv = (((xslot >> 1) ^ rowbits) << 1) | (xslot & 1);
        const int kcol = k0 + kv * 8;
        int tap = 0, ci = 0;
        if (kcol < a.Ktot) {
            tap = kcol / a.Cs;
            ci = kcol - tap * a.Cs;
        }
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        vsy[c] = kh - a.pad;
        vsx[c] = kw - a.pad;
        vtap[c] = (kh - a.pad) * a.Ws + kw - a.pad;
        vconst[c] = kcol < a.Ktot ? (a.src_coff + ci) * 2 : -1;
    }
    // running state of each instruction's pixel: linear output pixel rp, source row / column of tap (pad, pad) and the linear
    // source pixel spix of that tap; all advance by 64 output pixels per step with at most one carry per axis
    int rp[XI], rsy[XI], rsx[XI], spix[XI];
    {
        const int hw = a.Hd * a.Wd;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int p = p_begin + (wave * XI + i) * RPP + xrl;
            const int n = p / hw, rem = p - n * hw;
            const int oy = rem / a.Wd, ox = rem - oy * a.Wd;
            rp[i] = p;
            rsy[i] = oy * a.stride;
            rsx[i] = ox * a.stride;
            spix[i] = (n * a.Hs + rsy[i]) * a.Ws + rsx[i];
        }
    }
    // 64 pixels = an images + ay rows + ax pixels
    const int an = WKP / (a.Hd * a.Wd);
    const int ay = (WKP - an * a.Hd * a.Wd) / a.Wd;
    const int ax = WKP - an * a.Hd * a.Wd - ay * a.Wd;
    const int ldB = a.src_ld * 2;
    const int dsx = ax * a.stride, dsy = ay * a.stride;
    const int limx = a.Wd * a.stride, limy = a.Hd * a.stride;           // wrap limits of rsx / rsy
    const int dS = (an * a.Hs + dsy) * a.Ws + dsx;                       // source-pixel advance without carries
    const int dCx = a.stride * a.Ws - limx;                              // column carry: next output row, column - Wd
    const int dCy = a.Hs * a.Ws - limy * a.Ws;                           // row carry: next image, row - Hd

    // ---- dY instructions: vector v = (wave*YI + i)*64 + lane of the step's 64 x YV vectors (a wave without a last one repeats)
    int yoff[YI], ystep[YI], ydst[YI];
#pragma unroll
    for (int i = 0; i < YI; ++i) {
        int py = wave * YI + i;
        if (py >= YV) py = YV - 1;  // surplus slot: repeat the last instruction (same data, same place)
        const int v = py * 64 + lane;
        const int row = v / YV, slot = v - row * YV;
        const int cv = YSW ? ((((slot >> 1) ^ ((row >> 2) & 1)) << 1) | (slot & 1)) : slot;
        const bool ok = c0 + cv * 8 < a.Cd;
        yoff[i] = ok ? ((p_begin + row) * a.dy_ld + a.dy_coff + c0 + cv * 8) * 2 : SENT;
        ystep[i] = ok ? WKP * a.dy_ld * 2 : 0;
        ydst[i] = py * 1024;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0, 0x00020000);

    auto dma_piece = [&](int idx, int buf, bool live) {
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* ys = xs + WKP * XROW;
        if (idx < XI) {
            const int i = idx, c = i % NVAR;
            const int sy = rsy[i] + vsy[c], sx = rsx[i] + vsx[c];
            const bool ok = (vconst[c] >= 0) & (rp[i] < a.P) & ((unsigned)sy < (unsigned)a.Hs) & ((unsigned)sx < (unsigned)a.Ws);
            const int off = (int)(__umul24((unsigned)(spix[i] + vtap[c]), (unsigned)ldB) + (unsigned)vconst[c]);
            bufload_lds16_w(live ? rs_x : rs_x0, ok ? off : SENT, xs + (wave * XI + i) * 1024);
            // advance this instruction's pixel by 64
            rp[i] += WKP;
            rsx[i] += dsx;
            const bool cx = rsx[i] >= limx;
            rsx[i] -= cx ? limx : 0;
            rsy[i] += dsy + (cx ? a.stride : 0);
            const bool cyw = rsy[i] >= limy;
            rsy[i] -= cyw ? limy : 0;
            spix[i] += dS + (cx ? dCx : 0) + (cyw ? dCy : 0);
        } else {
            const int i = idx - XI;
            bufload_lds16_w(live ? rs_y : rs_y0, yoff[i], ys + ydst[i]);
            yoff[i] += ystep[i];
        }
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transpose-read offsets of this lane inside a stage: row (4q + li>>2) of a 32-pixel half, 8-byte chunk (li & 3) of the 32-byte unit
    const int rbits = (4 * q + (li >> 2)) & 7;
    int xo[4], yo[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) xo[i] = (4 * q + (li >> 2)) * XROW + (((wk * 4 + i) ^ rbits) * 32) + (li & 3) * 8;
#pragma unroll
    for (int j = 0; j < 5; ++j) yo[j] = WKP * XROW + (4 * q + (li >> 2)) * YROW + (((wc * 5 + j) ^ (YSW ? (q & 1) : 0)) * 32) + (li & 3) * 8;

    // fragment r of a 32-pixel half: r < 4 -> X (A operand), else dY (B operand); each is two transposing 8-byte reads
    auto frag1 = [&](int buf, int kk, int r, u32x4 (&af)[4], u32x4 (&bf)[5]) {
        const unsigned char* base = smem + buf * STAGE + (r < 4 ? xo[r] + kk * 32 * XROW : yo[r - 4] + kk * 32 * YROW);
        const u32x2 lo = tr_read(base), hi = tr_read(base + 16 * (r < 4 ? XROW : YROW));
        const u32x4 v = u32x4{lo[0], lo[1], hi[0], hi[1]};
        if (r < 4) af[r] = v;
        else bf[r - 4] = v;
    };
    constexpr int RORD[9] = {0, 4, 5, 6, 7, 8, 1, 2, 3};  // the first MFMAs of a phase need A row 0 and all five B columns

    u32x4 a0[4], b0[5], a1[4], b1[5];
    // both stages are requested before the first wait: only step 0's latency is exposed
#pragma unroll
    for (int i = 0; i < NV; ++i) dma_piece(i, 0, true);
#pragma unroll
    for (int i = 0; i < NV; ++i) dma_piece(i, 1, nsteps > 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 9; ++r) frag1(0, 0, r, a0, b0);
    for (int ks = 0; ks < nsteps; ++ks) {
        const int cur = ks & 1;
        const bool live = ks + 2 < nsteps;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 20; ++idx) {
            mfma_inplace_w<DT>(a0[idx / 5], b0[idx % 5], acc[idx / 5][idx % 5]);
            if ((idx & 1) == 0 && (idx >> 1) < 9) frag1(cur, 1, RORD[idx >> 1], a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 20; ++idx) {
            mfma_inplace_w<DT>(a1[idx / 5], b1[idx % 5], acc[idx / 5][idx % 5]);
            if (idx < NV) dma_piece(idx, cur, live);
            if (idx >= 20 - 9 - 1 && idx < 20 - 1) frag1(cur ^ 1, 0, RORD[idx - (20 - 9 - 1)], a0, b0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // let the last (asm, opaque to the hazard recogniser) MFMAs retire
    __syncthreads();

#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int co = c0 + wc * 80 + j * 16 + li;
        if (co >= a.Cd_pad) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + wk * 64 + i * 16 + q * 4;
            if (k < a.Kp) *reinterpret_cast<f32x4*>(wsp + (int64_t)co * a.Kp + k) = acc[i][j];
        }
    }
}

// per-element variant (small layers: O*I threads would not fill the chip)
__global__ __launch_bounds__(256) void wgrad_reduce_elem_kernel(const float* __restrict__ ws, float* __restrict__ dw, int S, int O, int Cd_pad, int I,
                                                               int KH, int KW, int Kp, int accumulate) {
    const int64_t total = (int64_t)O * I * KH * KW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        // iterate in packed order (o, kh, kw, i) so that workspace reads are coalesced
        const int i = (int)(idx % I);
        int64_t r = idx / I;
        const int kw = (int)(r % KW);
        r /= KW;
        const int kh = (int)(r % KH);
        const int o = (int)(r / KH);
        const int k = (kh * KW + kw) * I + i;
        // eight interleaved partial sums keep eight slab loads in flight (narrow layers have up to ~170 splits); the association
        // is fixed, so the result stays run-to-run identical
        const float* src = ws + (int64_t)o * Kp + k;
        const int64_t slab = (int64_t)Cd_pad * Kp;
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= S; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += src[(int64_t)(s + u) * slab];
        }
        for (int u = 0; s < S; ++s, ++u) part[u] += src[(int64_t)s * slab];
        const float sum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        float* d = dw + (((int64_t)o * I + i) * KH + kh) * KW + kw;
        *d = accumulate ? *d + sum : sum;
    }
}

// dw[o][i][kh][kw] (+)= sum_s ws[s][o][(kh,kw,i)], fixed order over s. One thread owns one kernel ROW of one (o, i) pair
// (kh fastest across threads): its KW taps are read with i running across the lanes of equal kh (coalesced slab reads) and
// written as KW consecutive floats, adjacent threads writing adjacent runs (the per-element version writes 4-byte elements
// KH*KW floats apart: 4x write amplification in the PMC counters).
template <int KH, int KW>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int S, int O, int Cd_pad, int I,
                                                          int Kp, int accumulate, const cdet_wgrad_item* __restrict__ items) {
    const int64_t total = (int64_t)O * I * KH;
    const int64_t slab = (int64_t)Cd_pad * Kp;
    if (items != nullptr) {  // grouped launch: blockIdx.y = layer, its S slabs are consecutive
        ws += (int64_t)blockIdx.y * S * slab;
        dw = items[blockIdx.y].dw;
    }
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int kh = (int)(idx % KH);
        const int64_t oi = idx / KH;
        const int i = (int)(oi % I);
        const int o = (int)(oi / I);
        const float* src = ws + (int64_t)o * Kp + kh * KW * I + i;
        float* d = dw + oi * (KH * KW) + kh * KW;
        float sum[KW], sum2[KW];  // two interleaved chains (fixed association): twice the loads in flight
#pragma unroll
        for (int tp = 0; tp < KW; ++tp) sum[tp] = sum2[tp] = 0.f;
        int s = 0;
        for (; s + 2 <= S; s += 2) {
#pragma unroll
            for (int tp = 0; tp < KW; ++tp) {
                sum[tp] += src[(int64_t)s * slab + tp * I];
                sum2[tp] += src[(int64_t)(s + 1) * slab + tp * I];
            }
        }
        if (s < S) {
#pragma unroll
            for (int tp = 0; tp < KW; ++tp) sum[tp] += src[(int64_t)s * slab + tp * I];
        }
#pragma unroll
        for (int tp = 0; tp < KW; ++tp) sum[tp] += sum2[tp];
#pragma unroll
        for (int tp = 0; tp < KW; ++tp) d[tp] = accumulate ? d[tp] + sum[tp] : sum[tp];
    }
}

static int wgrad_impl() {
    static int impl = -1;
    if (impl < 0) {
        impl = tune_env("CDET_WGRAD_IMPL", 3);  // 2 = register-staged single LDS stage, 3 = LDS-DMA + software-pipelined (default)
    }
    return impl;
}

struct WgradPlan {
    bool wide;
    int BKC, BCO, n_kblk, n_cblk, Kp, Cd_pad, S, chunk;
};

static WgradPlan plan_wgrad(const cdet_conv_desc* d) {
    WgradPlan p;
    p.wide = d->Cd > 80;
    p.BKC = p.wide ? 128 : 256;
    p.BCO = p.wide ? 160 : 80;
    const int Ktot = d->kh * d->kw * d->Cs;
    p.n_kblk = div_up(Ktot, p.BKC);
    p.n_cblk = div_up(d->Cd, p.BCO);
    p.Kp = p.n_kblk * p.BKC;
    p.Cd_pad = p.n_cblk * p.BCO;
    const int64_t P = (int64_t)d->N * d->Hd * d->Wd;
    const int tiles = p.n_kblk * p.n_cblk;
    // 2 workgroups fit a CU -> 512 resident slots. Every split costs a full fp32 slab write + re-read and a partly filled second
    // round costs a whole round, so: the largest split count (multiple of 8 for the XCD mapping, or 4 / 2 / 1) whose grid still fits
    // ONE round. Measured per task pass with the pipelined kernel: target 768 workgroups 15.9 ms, 576 16.7 (spills into a second
    // round on the 12-tile layers), 512 14.4, 448 14.0, 256 20.9.
    static int target = -1;
    if (target < 0) {
        target = tune_env("CDET_WGRAD_TARGET", 512);
    }
    int S = target / tiles;
    const int maxS = (int)((P + 511) / 512);  // at least 512 pixels per split
    if (S > maxS) S = maxS;
    const int64_t slab = (int64_t)p.Cd_pad * p.Kp * 4;
    const int capS = (int)((128ll << 20) / slab);  // at most 128 MB of partial slabs
    if (S > capS) S = capS;
    if (S >= 8) S = S / 8 * 8;
    else S = S >= 4 ? 4 : (S >= 2 ? 2 : 1);
    int chunk = (int)((P + S - 1) / S);
    chunk = (chunk + WKP - 1) / WKP * WKP;
    p.S = S;
    p.chunk = chunk;
    return p;
}

template <int DT, int WK, int WC>
static int launch_wgrad(const WgradArgs& a, hipStream_t s) {
    constexpr int BKC = 64 * WK, BCO = 80 * WC;
    constexpr int XROW = BKC * 2 + 32;
    constexpr int YROW = BCO * 2 + ((BCO * 2) % 256 == 0 ? 32 : 0);
    const size_t lds = (size_t)WKP * (XROW + YROW);
    static int abl = -1;
    if (abl < 0) abl = tune_env("CDET_WGRAD_ABLATE", 0);
    const int ntiles = a.n_kblk * a.n_cblk;
    const dim3 grid(a.S >= 8 ? ntiles * ((a.S + 7) / 8 * 8) : 8 * ((ntiles + 8 / a.S - 1) / (8 / a.S)));
    const int64_t xb = (int64_t)a.N * a.Hs * a.Ws * a.src_ld * 2, yb = (int64_t)a.P * a.dy_ld * 2;
    // (the 256 x 80 tile measured slower with the pipelined kernel: 0.342 vs 0.269 ms on 160x160 80->80 -- twice the X DMA instructions
    //  per wave; it keeps the register-staged kernel)
    const bool pix24 = (int64_t)a.N * a.Hs * a.Ws < (1 << 24) - 65536;  // v_mad_u32_u24 addressing of the source pixels
    if (wgrad_impl() >= 3 && WC == 2 && abl == 0 && pix24 && xb < 0xfffffff0ll - 256 && yb < 0xfffffff0ll - 256) {
        constexpr int lds3 = 2 * WKP * (BKC * 2 + BCO * 2);
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<DT, WK, WC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
            attr = true;
        }
        hipLaunchKernelGGL((conv_wgrad_pipe_kernel<DT, WK, WC>), grid, dim3(256), lds3, s, a, (unsigned)xb, (unsigned)yb);
        CDET_LAUNCH_CHECK();
        return 0;
    }
    if (abl == 1) hipLaunchKernelGGL((conv_wgrad_kernel<DT, WK, WC, 1>), grid, dim3(256), lds, s, a);
    else if (abl == 2) hipLaunchKernelGGL((conv_wgrad_kernel<DT, WK, WC, 2>), grid, dim3(256), lds, s, a);
    else if (abl == 3) hipLaunchKernelGGL((conv_wgrad_kernel<DT, WK, WC, 3>), grid, dim3(256), lds, s, a);
    else
    hipLaunchKernelGGL((conv_wgrad_kernel<DT, WK, WC>), grid, dim3(256), lds, s, a);
    CDET_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdet

using namespace cdet;

extern "C" int64_t cdet_conv2d_wgrad_ws_elems(const cdet_conv_desc* d) {
    if (!d) return -1;
    WgradHaloPlan hp;
    if (wgrad_halo_plan(d, &hp) || wgrad_gemm_plan(d, &hp) || wgrad_s2_plan(d, &hp)) return (int64_t)hp.S * hp.Cd_pad * hp.Kp;
    const WgradPlan p = plan_wgrad(d);
    return (int64_t)p.S * p.Cd_pad * p.Kp;
}

extern "C" int cdet_conv2d_wgrad(const cdet_conv_desc* d, const void* x, const void* dy, float* dw, float* ws, int32_t accumulate, void* stream) {
    CDET_CHECK_ARG(d && x && dy && dw && ws, "cdet_conv2d_wgrad: null pointer");
    CDET_CHECK_ARG(d->dtype == CDET_BF16 || d->dtype == CDET_F16, "cdet_conv2d_wgrad: dtype must be bf16/f16");
    CDET_CHECK_ARG(d->Cs % 8 == 0 && d->src_ld % 8 == 0 && d->src_coff % 8 == 0, "cdet_conv2d_wgrad: x channels/ld/coff must be multiples of 8");
    CDET_CHECK_ARG(d->dst_ld % 8 == 0 && d->dst_coff % 8 == 0, "cdet_conv2d_wgrad: dy ld/coff must be multiples of 8");
    CDET_CHECK_ARG(d->mode == CDET_CONV_FWD, "cdet_conv2d_wgrad: descriptor must describe the forward convolution");
    WgradPlan p = plan_wgrad(d);
    WgradHaloPlan hp;
    bool halo = wgrad_halo_plan(d, &hp);
    const bool gemm = !halo && wgrad_gemm_plan(d, &hp);
    const bool s2 = !halo && !gemm && wgrad_s2_plan(d, &hp);
    if (halo || gemm || s2) {  // stride-1 3x3 / 1x1 / stride-2 3x3: transpose-read kernels (conv_wgrad_halo.hip, conv_wgrad_s2.hip), same slab layout / reduction as below
        const int e = s2 ? wgrad_s2_launch(d, hp, x, dy, ws, (hipStream_t)stream)
                         : (gemm ? wgrad_gemm_launch(d, hp, x, dy, ws, (hipStream_t)stream) : wgrad_halo_launch(d, hp, x, dy, ws, (hipStream_t)stream));
        if (e) return e;
        halo = true;
        p.S = hp.S;
        p.Cd_pad = hp.Cd_pad;
        p.Kp = hp.Kp;
    }
    WgradArgs a;
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.ws = ws;
    a.N = d->N; a.Hs = d->Hs; a.Ws = d->Ws; a.Cs = d->Cs; a.Hd = d->Hd; a.Wd = d->Wd;
    a.Cd = (d->Cd + 7) / 8 * 8;  // dy carries the padded channel count; pad channels hold zeros
    a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dy_ld = d->dst_ld; a.dy_coff = d->dst_coff;
    a.Ktot = d->kh * d->kw * d->Cs; a.Kp = p.Kp; a.Cd_pad = p.Cd_pad;
    a.P = d->N * d->Hd * d->Wd; a.chunk = p.chunk; a.n_kblk = p.n_kblk; a.n_cblk = p.n_cblk; a.S = p.S;
    hipStream_t s = (hipStream_t)stream;
    int e;
    const bool fits32 = (int64_t)d->N * d->Hs * d->Ws * d->src_ld < (1ll << 31);
    CDET_CHECK_ARG(fits32, "cdet_conv2d_wgrad: tensor too large for 32-bit gather offsets");
    if (halo) e = 0;
    else if (d->dtype == CDET_BF16) e = p.wide ? launch_wgrad<CDET_BF16, 2, 2>(a, s) : launch_wgrad<CDET_BF16, 4, 1>(a, s);
    else e = p.wide ? launch_wgrad<CDET_F16, 2, 2>(a, s) : launch_wgrad<CDET_F16, 4, 1>(a, s);
    if (e) return e;
    const int taps = d->kh * d->kw;
    const int64_t rows = (int64_t)d->Cd * d->Cs * d->kh;  // threads of the row-wise reduction
    const bool rowwise = ((d->kh == 3 && d->kw == 3) || taps == 1) && rows >= 32768;
    if (!rowwise) {  // small layers / other kernel sizes: one thread per element keeps the chip busy
        const int64_t tot_e = (int64_t)d->Cd * d->Cs * taps;
        int be = (int)((tot_e + 255) / 256);
        if (be > 4096) be = 4096;
        hipLaunchKernelGGL(wgrad_reduce_elem_kernel, dim3(be), dim3(256), 0, s, ws, dw, p.S, d->Cd, p.Cd_pad, d->Cs, d->kh, d->kw, p.Kp, accumulate);
    } else {
        int blocks = (int)((rows + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (taps == 9) hipLaunchKernelGGL((wgrad_reduce_kernel<3, 3>), dim3(blocks), dim3(256), 0, s, ws, dw, p.S, d->Cd, p.Cd_pad, d->Cs, p.Kp, accumulate,
                                          (const cdet_wgrad_item*)nullptr);
        else hipLaunchKernelGGL((wgrad_reduce_kernel<1, 1>), dim3(blocks), dim3(256), 0, s, ws, dw, p.S, d->Cd, p.Cd_pad, d->Cs, p.Kp, accumulate,
                                (const cdet_wgrad_item*)nullptr);
    }
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t cdet_conv2d_wgrad_groupable(const cdet_conv_desc* d) {
    WgradHaloPlan hp;
    return d && d->mode == CDET_CONV_FWD && wgrad_halo_plan(d, &hp) ? 1 : 0;
}

extern "C" int64_t cdet_conv2d_wgrad_grouped_ws_elems(const cdet_conv_desc* d, int32_t n_items) {
    WgradHaloPlan hp;
    if (!d || n_items < 1 || !wgrad_halo_plan(d, &hp, n_items)) return -1;
    return (int64_t)n_items * hp.S * hp.Cd_pad * hp.Kp;
}

extern "C" int cdet_conv2d_wgrad_grouped(const cdet_conv_desc* d, const cdet_wgrad_item* items_dev, int32_t n_items, float* ws, int32_t accumulate,
                                         void* stream) {
    CDET_CHECK_ARG(d && items_dev && ws && n_items >= 1, "cdet_conv2d_wgrad_grouped: bad arguments");
    CDET_CHECK_ARG(d->mode == CDET_CONV_FWD, "cdet_conv2d_wgrad_grouped: descriptor must describe the forward convolution");
    CDET_CHECK_ARG(d->dst_ld % 8 == 0 && d->dst_coff % 8 == 0, "cdet_conv2d_wgrad_grouped: dy ld/coff must be multiples of 8");
    WgradHaloPlan hp;
    CDET_CHECK_ARG(wgrad_halo_plan(d, &hp, n_items), "cdet_conv2d_wgrad_grouped: geometry not groupable (see cdet_conv2d_wgrad_groupable)");
    hipStream_t s = (hipStream_t)stream;
    const int e = wgrad_halo_launch(d, hp, nullptr, nullptr, ws, s, items_dev, n_items);
    if (e) return e;
    const int64_t rows = (int64_t)d->Cd * d->Cs * 3;
    int blocks = (int)((rows + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((wgrad_reduce_kernel<3, 3>), dim3(blocks, n_items), dim3(256), 0, s, ws, (float*)nullptr, hp.S, d->Cd, hp.Cd_pad, d->Cs, hp.Kp, accumulate,
                       items_dev);
    CDET_LAUNCH_CHECK();
    return 0;
}
