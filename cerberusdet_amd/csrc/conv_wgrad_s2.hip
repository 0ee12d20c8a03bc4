// Tap-resident weight gradient of the stride-2 3x3 convolution (pad 1, even H / W; gfx950, bf16 / f16 in, fp32 accumulate):
//
//   dW[co][ky][kx][ci] = sum_{n,oh,ow} dY[n,oh,ow,co] * X[n, 2*oh + ky - 1, 2*ow + kx - 1, ci]      (autograd's convolution_backward(weight)
//                                                                                                    of the downsampling Convs, models/common.py:57)
//
// The im2col kernel (conv_wgrad.hip) gathers the nine taps of every output pixel separately. Here the machinery of conv_wgrad_halo.hip
// (one workgroup of 8 waves per CU, 160 couts x 32 cins x 9 taps = 45 accumulator tiles of v_mfma_f32_16x16x32 per wave, both operands
// reduction-major in LDS and transposed by ds_read_b64_tr_b16, every read "8 bytes at base + 8*lane" + an instruction immediate) runs on
// the input's four PARITY PLANES, as conv_vt.hip does for the forward: tap (ky, kx) reads input row 2*oh + ky - 1 -- an odd row for
// ky = 0 / 2, an even one for ky = 1 -- and likewise for the columns, so with
//      oo[i][j] = X[2i-1][2j-1]   (taps (0,0) (0,2) (2,0) (2,2))        oe[i][j] = X[2i-1][2j]   (taps (0,1) (2,1))
//      eo[i][j] = X[2i  ][2j-1]   (taps (1,0) (1,2))                    ee[i][j] = X[2i  ][2j]   (tap  (1,1))
// consecutive output pixels of a row read consecutive plane entries. A stage is a PATCH of 8 x 16 output pixels (128 pixels = the 3x3
// kernel's stage): dY [128][160] (40 KB) + per 16-cin plane the four parity planes with their one-entry halo (oo 9 x 17, oe 9 x 16,
// eo 8 x 17, ee 8 x 16 entries of 32 B; 19 KB), double buffered = 156 KB of LDS. A 16-pixel block of the reduction is one patch row, so
// every tap offset is an immediate ((row + [ky == 2]) * pitch + [kx == 2]) and -- unlike the linear pixel ranges of the stride-1
// kernel -- entries outside the image are ZERO-FILLED by the DMA itself (sentinel offset), so the reads need no validity masks.
// Pixels of a ragged patch beyond the map get zero dY rows (their X entries hold some other finite pixel: 0 * finite = 0).
// Output: partial slab ws[split][co][tap*Cs + ci] (fp32), reduced by wgrad_reduce_kernel<3,3> (conv_wgrad.hip) like the stride-1 slabs.
#include "common.h"
#include "wgrad_tr.h"

namespace cdet {

struct WS2Args {
    const uint16_t* x;
    const uint16_t* dy;
    float* ws;
    int W, Ho, Wo;  // input width (= 2 * Wo), output map
    int Cs, Cd;
    int src_ld, src_coff, dy_ld, dy_coff;
    int Kp, Cd_pad;
    int chunk, S;   // patches per split, splits
    int n_cblk, n_iblk;
    int tx, ty, NP;  // patches per row / column of an image, patches in all
    unsigned x_bytes, dy_bytes;
};

constexpr int S2_DYB = 40 * 1024;                       // dY bytes per stage: 8 patch rows x 5 cout pairs, 1 KiB each
constexpr int S2_OO = 0, S2_OE = 160, S2_EO = 320, S2_EE = 480, S2_ENT = 608;  // plane starts / total, in 32-byte entries
constexpr int S2_XP = S2_ENT * 32;                      // bytes of the parity planes of one 16-cin plane
constexpr int S2_STAGE = S2_DYB + 2 * S2_XP;            // 79872
constexpr int S2_NXP = 2 * S2_ENT / 32;                 // 38 X pieces of 1 KiB per stage

// (plane start, pitch, extra row, extra column) of tap tp = ky*3 + kx, folded into the read immediate of patch row r
template <int TP, int R>
struct S2Tap {
    static constexpr int ky = TP / 3, kx = TP % 3;
    static constexpr bool p17 = kx != 1;                // odd-column planes carry the left halo column: pitch 17
    static constexpr int start = ky == 1 ? (kx == 1 ? S2_EE : S2_EO) : (kx == 1 ? S2_OE : S2_OO);
    static constexpr int imm = (start + (R + (ky == 2 ? 1 : 0)) * (p17 ? 17 : 16) + (kx == 2 ? 1 : 0)) * 32;
};

template <int DT>
__global__ __launch_bounds__(512, 2) void wgrad_s2_kernel(const WS2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = wave >> 2, wco = (wave >> 1) & 1, wci = wave & 1;  // patch rows 4*half .. 4*half+3, 80 couts, 16 cins
    const int q = lane >> 4, li = lane & 15;

    int L;  // workgroup -> (split, tile): the tiles of ONE split on one XCD (they share the split's X / dY patches)
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, qq = nwg >> 3, rr = nwg & 7, j = b >> 3;
        L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + j;
    }
    const int ntiles = a.n_cblk * a.n_iblk;
    const int split = L / ntiles;
    const int tile = L - split * ntiles;
    const int cblk = tile / a.n_iblk, iblk = tile - cblk * a.n_iblk;
    const int c0 = cblk * 160, i0 = iblk * 32;
    const int pbeg = split * a.chunk;
    const int pend = min(pbeg + a.chunk, a.NP);
    const int nst = pbeg < pend ? pend - pbeg : 0;
    const int W = a.W, Wo = a.Wo, Ho = a.Ho;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    const int ldyB = a.dy_ld * 2, ldxB = a.src_ld * 2;

    // ---- DMA duties of this wave, 1 KiB = one instruction each, ten per stage:
    //   dY image di = wave + 8*idx (idx 0..4) = (patch row di/5, cout pair di%5); lane -> (cout group lane>>5, column (lane>>1)&15, 16-byte
    //      half lane&1);
    //   X piece xi = wave + 8*(idx-5) < 38 = (16-cin plane xi/19, 32-entry block xi%19); lane -> (entry lane>>1, half lane&1). Which input
    //      pixel an entry holds depends on the lane only up to the patch origin, so the offset relative to the origin and the two border
    //      flags are computed once here.
    const int yco_l = (lane >> 5) * 16 + (lane & 1) * 8;
    const int ypx = (lane >> 1) & 15;
    const unsigned ydl = (unsigned)(ypx * ldyB + yco_l * 2);
    int xrel[5];           // byte offset of the lane's 16 bytes relative to pixel (2*oh0, 2*ow0) of the patch's image
    unsigned xfl = 0;      // per duty k: bit 3k never valid (allocation pad / cins beyond Cs), 3k+1 top halo row, 3k+2 left halo column
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int xi = wave + 8 * k;
        const int pl = xi >= 19 ? 1 : 0, blk = xi - pl * 19;
        const int e = blk * 32 + (lane >> 1);
        int start, pitch, ro, co;
        if (e < S2_OE) { start = S2_OO; pitch = 17; ro = 1; co = 1; }
        else if (e < S2_EO) { start = S2_OE; pitch = 16; ro = 1; co = 0; }
        else if (e < S2_EE) { start = S2_EO; pitch = 17; ro = 0; co = 1; }
        else { start = S2_EE; pitch = 16; ro = 0; co = 0; }
        const int e2 = e - start;
        const int er = e2 / pitch, ec = e2 - er * pitch;
        const bool never = xi >= S2_NXP || er >= 8 + ro || i0 + pl * 16 >= a.Cs;
        xrel[k] = ((2 * er - ro) * W + (2 * ec - co)) * ldxB + (a.src_coff + i0 + pl * 16 + (lane & 1) * 8) * 2;
        xfl |= (never ? 1u : 0u) << (3 * k) | ((ro && er == 0) ? 1u : 0u) << (3 * k + 1) | ((co && ec == 0) ? 1u : 0u) << (3 * k + 2);
    }

    // patch whose operands are fetched next (one stage ahead of the one being multiplied)
    int f_n, f_ty, f_tx;
    {
        const int per = a.ty * a.tx;
        f_n = pbeg / per;
        const int r = pbeg - f_n * per;
        f_ty = r / a.tx;
        f_tx = r - f_ty * a.tx;
    }
    int oh0 = 0, ow0 = 0;
    unsigned ox = 0, oy = 0;
    auto next_patch = [&]() {  // (oh0, ow0, byte origins) of the fetch patch, then advance it
        oh0 = f_ty * 8;
        ow0 = f_tx * 16;
        ox = (unsigned)(((f_n * 2 * Ho + 2 * oh0) * W + 2 * ow0) * ldxB);
        oy = (unsigned)(((f_n * Ho + oh0) * Wo + ow0) * ldyB + (a.dy_coff + c0) * 2);
        if (++f_tx == a.tx) {
            f_tx = 0;
            if (++f_ty == a.ty) {
                f_ty = 0;
                ++f_n;
            }
        }
    };
    auto issue_dma = [&](int idx, int st, int buf) {  // idx 0..4: dY images, 5..9: X pieces, of the fetch patch -> buffer buf
        if (st >= nst) return;
        unsigned char* base = smem + buf * S2_STAGE;
        if (idx < 5) {
            const int di = wave + 8 * idx;  // wave-uniform
            const int pb = di / 5, cp = di - pb * 5;
            const int cobase = cp * 32;
            const unsigned sc = (unsigned)(pb * Wo * ldyB + cobase * 2);
            const bool ok = yco_l < a.Cd - (c0 + cobase) && oh0 + pb < Ho && ow0 + ypx < Wo;
            wh_dma16(rs_y, ok ? oy + sc + ydl : WH_SENT, base + di * 1024);
        } else {
            const int k = idx - 5;
            const int xi = wave + 8 * k;
            if (xi < S2_NXP) {
                const unsigned f = xfl >> (3 * k);
                const bool zero = (f & 1u) || ((f & 2u) && oh0 == 0) || ((f & 4u) && ow0 == 0);
                wh_dma16(rs_x, zero ? WH_SENT : ox + (unsigned)xrel[k], base + S2_DYB + xi * 1024);
            }
        }
    };

    // ---- transpose-read bases ("8 bytes at base + 8*lane" + immediate)
    const int yb0 = half * 4 * 5120 + wco * 2560 + lane * 8;
    const int xb17 = S2_DYB + wci * S2_XP + half * 4 * 17 * 32 + lane * 8;
    const int xb16 = S2_DYB + wci * S2_XP + half * 4 * 16 * 32 + lane * 8;

    f32x4 acc[9][5];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[tp][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    next_patch();
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_dma(i, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        int vy = yb0 + cur * S2_STAGE, v17 = xb17 + cur * S2_STAGE, v16 = xb16 + cur * S2_STAGE;
        asm volatile("" : "+v"(vy), "+v"(v17), "+v"(v16));
        next_patch();
        wh_static_for(std::make_integer_sequence<int, 2>{}, [&](auto KS) {
            constexpr int ksl = decltype(KS)::value;  // 32-pixel reduction step = patch rows 4*half + 2*ksl, + 1
            u32x2 blo[5], bhi[5];
            wh_static_for(std::make_integer_sequence<int, 5>{}, [&](auto J) {
                constexpr int j = decltype(J)::value;
                blo[j] = wh_tr<(2 * ksl) * 5120 + j * 512>(vy);
                bhi[j] = wh_tr<(2 * ksl + 1) * 5120 + j * 512>(vy);
            });
            u32x2 alo[2], ahi[2];
            alo[0] = wh_tr<S2Tap<0, 2 * ksl>::imm>(v17);
            ahi[0] = wh_tr<S2Tap<0, 2 * ksl + 1>::imm>(v17);
            wh_static_for(std::make_integer_sequence<int, 9>{}, [&](auto TP) {
                constexpr int tp = decltype(TP)::value;
                constexpr int cs = tp & 1, ns = cs ^ 1;
                if constexpr (tp + 1 < 9) {
                    using T0 = S2Tap<tp + 1, 2 * ksl>;
                    using T1 = S2Tap<tp + 1, 2 * ksl + 1>;
                    alo[ns] = wh_tr<T0::imm>(T0::p17 ? v17 : v16);
                    ahi[ns] = wh_tr<T1::imm>(T1::p17 ? v17 : v16);
                }
                // the next patch's operands, one DMA instruction per tap of the first step and one more in the second
                if constexpr (ksl == 0) issue_dma(tp, st + 1, cur ^ 1);
                if constexpr (ksl == 1 && tp == 0) issue_dma(9, st + 1, cur ^ 1);
                if constexpr (tp == 0) wh_wait_b<2>(blo, bhi, alo[0], ahi[0]);
                else if constexpr (tp + 1 < 9) wh_wait<2>(alo[cs], ahi[cs]);
                else wh_wait<0>(alo[cs], ahi[cs]);
                const u32x4 av{alo[cs][0], alo[cs][1], ahi[cs][0], ahi[cs][1]};
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const u32x4 bv{blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                    wh_mfma<DT>(av, bv, acc[tp][j]);
                }
            });
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- waves 4..7 (patch rows 4..7) hand their partial tiles to waves 0..3 through LDS (two rounds: 23 + 22 tiles of 1 KiB)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int t0 = rd * 23, t1 = rd ? 45 : 23;
        unsigned char* slot = smem + ((wave & 3) * 23) * 1024 + lane * 16;
        if (half == 1) {
#pragma unroll
            for (int tl = 0; tl < 45; ++tl)
                if (tl >= t0 && tl < t1) *reinterpret_cast<f32x4*>(slot + (tl - t0) * 1024) = acc[tl / 5][tl % 5];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int tl = 0; tl < 45; ++tl)
                if (tl >= t0 && tl < t1) acc[tl / 5][tl % 5] += *reinterpret_cast<const f32x4*>(slot + (tl - t0) * 1024);
        }
        __syncthreads();
    }
    // ---- partial slab: C[ci][co] tiles -> ws[split][co][tap*Cs + ci], 4 consecutive cins per lane (not the cins beyond Cs)
    if (half == 0 && i0 + wci * 16 < a.Cs) {
        float* wsp = a.ws + (int64_t)split * a.Cd_pad * a.Kp;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int co = c0 + (wco * 5 + j) * 16 + li;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int k = tp * a.Cs + i0 + wci * 16 + 4 * q;
                *reinterpret_cast<f32x4*>(wsp + (int64_t)co * a.Kp + k) = acc[tp][j];
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool wgrad_s2_plan(const cdet_conv_desc* d, WgradHaloPlan* out) {
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1)) return false;
    if ((d->Hs & 1) || (d->Ws & 1) || d->Hd * 2 != d->Hs || d->Wd * 2 != d->Ws) return false;
    if (d->Cs % 16 != 0 || d->Cd < 128) return false;
    if (d->Cs * 10 < (d->Cs + 31) / 32 * 32 * 7) return false;  // last cin tile more than 30 % empty
    if (!(d->dtype == CDET_BF16 || d->dtype == CDET_F16)) return false;
    if (sw(SW_WGRAD_S2) == 0) return false;  // 0: im2col kernel (the tests compare the two)
    const int ty = div_up(d->Hd, 8), tx = div_up(d->Wd, 16);
    if ((int64_t)d->Hd * d->Wd * 4 < (int64_t)ty * 8 * tx * 16 * 3) return false;  // patches less than 75 % full (the 20x20 maps: 52 %)
    const int64_t Min = (int64_t)d->N * d->Hs * d->Ws, Mout = (int64_t)d->N * d->Hd * d->Wd;
    if (Min * d->src_ld * 2 >= 0xC0000000ll || Mout * d->dst_ld * 2 >= 0xC0000000ll) return false;
    const int n_cblk = div_up(d->Cd, 160), n_iblk = div_up(d->Cs, 32);
    const int tiles = n_cblk * n_iblk;
    if (tiles > 256) return false;
    const int NP = d->N * ty * tx;
    int S = 256 / tiles;  // one workgroup per CU, one round
    const int maxS = (NP + 3) / 4;  // at least 4 patches per split
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    const int chunk = (NP + S - 1) / S;
    S = (NP + chunk - 1) / chunk;
    out->S = S;
    out->chunk = chunk;
    out->Kp = 9 * d->Cs;
    out->Cd_pad = n_cblk * 160;
    out->n_cblk = n_cblk;
    out->n_iblk = n_iblk;
    out->XH = 0;
    out->nci = 2;
    out->narrow = 0;
    out->patch = 0;
    out->lds = 2 * (size_t)S2_STAGE;
    return true;
}

template <int DT>
static void wgrad_s2_launch_t(const WS2Args& a, int grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)wgrad_s2_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_s2_kernel<DT>), dim3(grid), dim3(512), lds, s, a);
}

int wgrad_s2_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s) {
    WS2Args a;
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.ws = ws;
    a.W = d->Ws; a.Ho = d->Hd; a.Wo = d->Wd;
    a.Cs = d->Cs; a.Cd = (d->Cd + 7) / 8 * 8;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dy_ld = d->dst_ld; a.dy_coff = d->dst_coff;
    a.Kp = p.Kp; a.Cd_pad = p.Cd_pad; a.chunk = p.chunk; a.S = p.S; a.n_cblk = p.n_cblk; a.n_iblk = p.n_iblk;
    a.ty = div_up(d->Hd, 8); a.tx = div_up(d->Wd, 16); a.NP = d->N * a.ty * a.tx;
    a.x_bytes = (unsigned)((int64_t)d->N * d->Hs * d->Ws * d->src_ld * 2);
    a.dy_bytes = (unsigned)((int64_t)d->N * d->Hd * d->Wd * d->dst_ld * 2);
    const int grid = p.S * p.n_cblk * p.n_iblk;
    if (d->dtype == CDET_BF16) wgrad_s2_launch_t<CDET_BF16>(a, grid, p.lds, s);
    else wgrad_s2_launch_t<CDET_F16>(a, grid, p.lds, s);
    CDET_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdet
