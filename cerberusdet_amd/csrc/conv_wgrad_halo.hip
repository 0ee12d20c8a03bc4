// Tap-resident weight gradient of the stride-1 3x3 convolution (gfx950, bf16 / f16 in, fp32 accumulate):
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[p + off(tap)][ci]          (autograd's convolution_backward(weight), models/common.py:57)
//
// The im2col form in conv_wgrad.hip gathers every pixel's 9 taps separately (a 128 x 160 output tile re-fetches X once per k-block:
// 36 KB of L2 -> LDS traffic per 64-pixel step, 56 B/clk per workgroup against the ~40 B/clk/CU the DMA path delivers -> 600 TF/s).
// Here, as in conv_halo.hip, the pixel tile is staged ONCE with its halo and serves all nine taps:
//   * a workgroup (8 waves, one per CU) owns the output tile [160 couts] x [16*NCI cins] x [9 taps] for one split of the pixel range.
//     Per stage of 128 consecutive pixels it stages dY [128][160] (40 KB) and the X halo [128 + 2W + 2][16*NCI] (LDS-DMA, double
//     buffered). NCI = 4 (64 cins): every wave runs all 128 pixels for its own [80 couts] x [16 cins] x [9 taps] = 45 accumulator
//     tiles of v_mfma_f32_16x16x32 (180 fp32 registers per lane). NCI = 2 (32 cins, for Cs % 64 != 0 or wide rows whose halo does not
//     fit): waves 0-3 take pixels 0..63 of a stage and waves 4-7 pixels 64..127 of the same tiles; the halves are summed through LDS
//     at the end, so a workgroup always writes ONE fp32 slab;
//   * both operands are reduction-major in LDS ([pixel][channel] rows); ds_read_b64_tr_b16 transposes them on the way to the registers
//     (semantics probed in profiles/r01_probe_ds_read_tr_b16.txt). A lane's 8 reduction indices are pixels {4q..4q+3, 16+4q..16+4q+3}
//     of a 32-pixel step for BOTH operands. Every transpose read of the kernel is the pattern "lane l reads 8 bytes at base + 8*l":
//     dY sits as 1 KiB images [16-pixel block][cout pair][2 cout groups][16 px][32 B], X as NCI planes of 32-byte rows
//     [16-cin plane][halo row][32 B] -- 512 contiguous bytes per instruction, conflict-free (profiles/r02_tr_bank_probe.txt);
//   * the addresses are affine: one base register per operand and stage, everything else is an instruction immediate. Taps that leave
//     the image are redirected per lane to a zero region with ONE v_cndmask on a precomputed 64-lane mask (top / bottom / left /
//     right validity of the lane's pixel; corner taps AND two masks on the scalar unit). The transpose reads are inline asm with
//     explicit lgkmcnt accounting: the compiler would otherwise drain vmcnt(0) -- the next stage's DMA -- in front of every LDS read.
//   * NARROW (Cout <= 96): 80-cout x 32-cin tile; the eight waves are four pixel groups x two cin halves (see the kernel);
//   * grouped launches: with a.items set, the workgroups of ONE launch are spread over n layers of identical geometry (device table of
//     per-layer tensors), so the pixel split per layer -- and the slab below -- shrinks by n (cdet_conv2d_wgrad_grouped).
// Output: partial slab ws[layer][split][co][tap*Cs + ci] (fp32) -- the layout wgrad_reduce_kernel<3,3> of conv_wgrad.hip reduces.
// wgrad_gemm_kernel further down is the 1x1 form of the same machinery.
#include "common.h"
#include "wgrad_tr.h"

namespace cdet {

struct WHArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* ws;
    int H, W, M;
    int Cs, Cd;
    int src_ld, src_coff, dy_ld, dy_coff;
    int Kp, Cd_pad;
    int chunk, S;
    int n_cblk, n_iblk;
    int XH;  // halo rows per stage (multiple of 32)
    const cdet_wgrad_item* items;  // grouped launch: per-layer tensors (nullptr: the single layer above)
    unsigned magicW, magicH;  // 2^32 / W + 1, 2^32 / H + 1: exact quotients by v_mul_hi for every pixel index of the path
                              // (PATCH form: 2^32 / tiles_x + 1 and 2^32 / tiles_per_img + 1, applied to the patch index)
    int tiles_x, tiles_per_img;  // PATCH form: 8 x 16 pixel patches per image row / per image
    unsigned x_bytes, dy_bytes;
};

constexpr int WH_P = 128;            // pixels per stage
constexpr int WH_DYB = 40 * 1024;    // dY bytes per stage: 8 pixel blocks of 16 x 5 cout pairs, 1 KiB each
constexpr int WH_ZERO = 4608;        // zero region: 512 lane bytes + the largest X immediate (114 rows of 32 B)
#ifndef WH_ABL  // profiling builds only (make EXTRA="-DWH_ABL=n"): 1 no DMA in the loop, 2 no stage barrier, 4 no MFMA, 8 no A reads, 16 no B reads
#define WH_ABL 0
#endif

// NARROW (Cout <= 96, NCI = 2): the tile is 80 couts x 32 cins; the factor two the cout halves would take goes to the pixels -- four
// groups of two waves, each reducing 32 of the stage's 128 pixels, summed through LDS in two levels at the end.
// PATCH (NCI = 2 forms; H % 8 == 0, W % 16 == 0 -- the 160 x 160 and 80 x 80 maps): a stage is an 8 x 16 pixel PATCH instead of 128 consecutive
// pixels. On a 160-wide map a run of 128 pixels drags 128 + 2 W + 2 = 450 halo rows along (3.5 rows of X per pixel), the patch 10 x 18 = 180; with
// the dY images packed for 80 couts (3 cout pairs instead of 5) a stage shrinks from 70 KB to 36 KB of LDS-DMA traffic, and FOUR stage buffers fit
// where two did: the DMA of stage st + 3 is issued during stage st and waited for with a counted vmcnt. Measured on the six grouped 160 x 160
// 80 -> 80 layers (profiles/r04_wgrad_patch.txt): the linear form's stage loop is bound by its DMA stream (0.89 ms with the MFMAs removed, 0.47 ms
// with the DMA removed, 0.94 ms complete). The halo is zero-filled by the DMA's out-of-range offsets, so the tap masks disappear: tap (ky, kx) of
// patch row r is 16 consecutive halo rows from (r + ky) * 18 + kx. The 160-cout tile takes the same form with THREE stage buffers (40 KB of dY +
// 12 KB of halo per stage; the linear form's two buffers + zero region leave no room for a third).
template <int DT, int NCI, bool NARROW = false, bool PATCH = false>
__global__ __launch_bounds__(512, 2) void wgrad_halo_kernel(const WHArgs a) {
    static_assert(!NARROW || NCI == 2, "the 80-cout tile exists for the 32-cin form only");
    static_assert(!PATCH || NCI == 2, "the patch form exists for the 32-cin tiles only");
    constexpr int KSL = NARROW ? 1 : (NCI == 4 ? 4 : 2);  // 32-pixel reduction steps a wave runs per stage
    constexpr int CO = NARROW ? 80 : 160;                  // couts of the tile
    constexpr int NBUF = PATCH ? (NARROW ? 4 : 3) : 2;     // stage buffers
    constexpr int NPAIR = PATCH && NARROW ? 3 : 5;         // 32-cout pairs of dY images per 16-pixel block
    constexpr int DYBLK = NPAIR * 1024;                    // bytes of the dY images of one 16-pixel block
    constexpr int DYB = 8 * DYBLK;                         // dY bytes per stage (= WH_DYB for five pairs)
    constexpr int HP = 18;                                 // PATCH: halo pitch (16 + 2)
    constexpr int NDY = NPAIR;                             // PATCH: dY DMA instructions per wave and stage (8 NPAIR images over 8 waves)
    constexpr int KW = NDY + 1;                            // PATCH: DMA instructions per stage of waves 4 - 7 (one X piece); waves 0 - 3: + 1
    constexpr int AHEAD = NBUF - 1;                        // PATCH: stages the DMA runs ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = NCI == 4 ? 0 : wave >> 2;             // (senders / receivers of the first end-of-kernel exchange)
    const int prow = NARROW ? (wave >> 1) * 32 : half * 64;  // first pixel row of the stage this wave reduces
    const int wco = NARROW ? 0 : (NCI == 4 ? wave >> 2 : (wave >> 1) & 1);
    const int wci = NCI == 4 ? wave & 3 : wave & 1;
    const int q = lane >> 4, li = lane & 15;

    // workgroup -> (split, tile): consecutive logical ids = the tiles of ONE split, on one XCD (they share the split's X / dY rows)
    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, qq = nwg >> 3, rr = nwg & 7, j = b >> 3;
        L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + j;
    }
    const int ntiles = a.n_cblk * a.n_iblk;
    const int slab_id = L / ntiles;  // (layer, split): the slab this workgroup writes
    const int tile = L - slab_id * ntiles;
    const int layer = slab_id / a.S;
    const int split = slab_id - layer * a.S;
    const uint16_t* xptr = a.x;
    const uint16_t* dyptr = a.dy;
    int src_ld = a.src_ld, src_coff = a.src_coff;
    unsigned x_bytes = a.x_bytes;
    if (a.items != nullptr) {  // wave-uniform index: scalar loads
        const cdet_wgrad_item it = a.items[layer];
        xptr = (const uint16_t*)it.x;
        dyptr = (const uint16_t*)it.dy;
        src_ld = it.src_ld;
        src_coff = it.src_coff;
        x_bytes = (unsigned)a.M * (unsigned)src_ld * 2u;
    }
    const int cblk = tile / a.n_iblk, iblk = tile - cblk * a.n_iblk;
    const int c0 = cblk * CO, i0 = iblk * 16 * NCI;
    const int pbeg = split * a.chunk;
    const int pend = min(pbeg + a.chunk, a.M);
    const int nst = pbeg < pend ? (pend - pbeg + WH_P - 1) / WH_P : 0;
    const int W = a.W;
    const int XP = a.XH * 32;  // bytes of one 16-cin plane
    const int STAGE = DYB + NCI * XP;
    const int ZOFF = NBUF * STAGE;
    if (!PATCH)
        for (int i = t; i < WH_ZERO / 4; i += 512) reinterpret_cast<uint32_t*>(smem + ZOFF)[i] = 0u;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xptr, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)dyptr, 0, (int)a.dy_bytes, 0x00020000);

    // ---- DMA duties of this wave (1 KiB = one instruction each; out-of-range offsets fetch zeros, which covers the rows above the
    //      first and below the last pixel of the tensor, and every pixel >= M of a ragged last stage):
    //   dY image di = wave + 8*idx (idx 0..4) = (16-pixel block di/5, cout pair di%5); lane -> (cout group lane>>5, pixel (lane>>1)&15,
    //      16-byte half lane&1): 64 contiguous bytes per pixel;
    //   X piece xi = wave + 8*idx = (32-row block xi / NCI, plane xi % NCI); lane -> (row lane>>1, half lane&1).
    const int ldyB = a.dy_ld * 2, ldxB = src_ld * 2;
    const int nxp = NCI * (a.XH >> 5);
    const int yco_l = (lane >> 5) * 16 + (lane & 1) * 8;
    const unsigned ydl = (unsigned)(((lane >> 1) & 15) * ldyB + yco_l * 2);
    const unsigned xdl = (unsigned)((lane >> 1) * ldxB + (lane & 1) * 16);
    // PATCH: the stage's patch (pid = first patch of the split + st) -> image n, origin (y0, x0); idx 0..NPAIR-1: dY images di = wave + 8 idx =
    // (patch row di / NPAIR, cout pair di % NPAIR), idx NPAIR, NPAIR+1: X pieces xi = wave + 8 (idx - NPAIR) < 12 = (32-row block xi / 2, plane xi & 1)
    // of the 10 x 18 halo
    auto issue_dma_patch = [&](int idx, int st, int buf) {
        if (st >= nst) return;
        if ((WH_ABL & 1) && st > 0) return;
        unsigned char* base = smem + buf * STAGE;
        const unsigned pid = (unsigned)(pbeg / WH_P + st);
        // (2^32 / 1 + 1 does not fit the magic word: a divisor of one is its own case)
        const unsigned n = a.tiles_per_img == 1 ? pid : __umulhi(pid, a.magicH);
        const unsigned r = pid - n * (unsigned)a.tiles_per_img;
        const unsigned ty = a.tiles_x == 1 ? r : __umulhi(r, a.magicW);
        const int y0 = (int)ty * 8, x0 = (int)(r - ty * (unsigned)a.tiles_x) * 16;
        if (idx < NDY) {
            const int di = wave + 8 * idx;  // wave-uniform
            const int pb = di / NPAIR, cp = di - pb * NPAIR;
            const int cobase = c0 + cp * 32;
            const unsigned sc = (unsigned)((((int)n * a.H + y0 + pb) * W + x0) * ldyB + (a.dy_coff + cobase) * 2);
            const unsigned v = yco_l < a.Cd - cobase ? ydl + sc : WH_SENT;
            wh_dma16(rs_y, v, base + pb * DYBLK + cp * 1024);
        } else {
            const int xi = wave + 8 * (idx - NDY);  // wave-uniform
            if (xi < NCI * 6) {
                const int blk = xi / NCI, pl = xi - blk * NCI;
                const int hr = blk * 32 + (lane >> 1);
                const int hy = (hr * 3641) >> 16, hx = hr - hy * HP;  // hr / 18 for hr < 192
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool ok = hr < 10 * HP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)W && i0 + pl * 16 < a.Cs;
                const unsigned off = (unsigned)((((int)n * a.H + y) * W + x) * ldxB + (src_coff + i0 + pl * 16) * 2 + (lane & 1) * 16);
                wh_dma16(rs_x, ok ? off : WH_SENT, base + DYB + pl * XP + blk * 1024);
            }
        }
    };
    auto issue_dma = [&](int idx, int st, int buf) {  // idx 0..4: dY images, 5..9: X pieces, of stage st -> buffer buf
        if (PATCH) {
            if (idx < NDY + 2) issue_dma_patch(idx, st, buf);
            return;
        }
        if (st >= nst) return;
        if ((WH_ABL & 1) && st > 0) return;
        unsigned char* base = smem + buf * STAGE;
        const int pb0 = pbeg + st * WH_P;
        if (idx < 5) {
            const int di = wave + 8 * idx;  // wave-uniform
            const int pb = di / 5, cp = di - pb * 5;
            const int cobase = c0 + cp * 32;
            if (NARROW && cp >= 3) return;  // an 80-cout tile reads cout pairs 0..2 only
            const unsigned sc = (unsigned)((pb0 + pb * 16) * ldyB + (a.dy_coff + cobase) * 2);
            const unsigned v = yco_l < a.Cd - cobase ? ydl + sc : WH_SENT;
            wh_dma16(rs_y, v, base + di * 1024);
        } else {
            const int xi = wave + 8 * (idx - 5);  // wave-uniform
            if (xi < nxp) {
                const int blk = xi / NCI, pl = xi - blk * NCI;
                // (rows above the tensor give a negative pixel: the 32-bit offset wraps far beyond the buffer -> zeros;
                //  a plane beyond Cs -- the last cin tile of an 80-channel layer -- is filled with zeros as well)
                const unsigned sc = (unsigned)((pb0 + blk * 32 - (W + 1)) * ldxB + (src_coff + i0 + pl * 16) * 2);
                wh_dma16(rs_x, i0 + pl * 16 < a.Cs ? xdl + sc : WH_SENT, base + WH_DYB + pl * XP + blk * 1024);
            }
        }
    };

    // ---- transpose-read bases: every read is "8 bytes at base + 8*lane" (+ immediate)
    //   dY: K-step ks (32 px), 16-pixel block 2*ks + hh, cout group cog = wco*5 + j -> image (2*ks + hh)*5 + cog/2, half cog&1
    //       = byte (2*ks + hh)*5120 + cog*512
    //   X : halo row of pixel px and tap (ky, kx) = px + ky*W + kx, plane wci
    const int yb0 = wco * 2560 + lane * 8 + (prow >> 5) * (2 * DYBLK);
    // (PATCH: the wave's 32 pixels are patch rows 2 g, 2 g + 1 with g = prow / 32: halo rows from 2 g * 18)
    const int xb0 = DYB + wci * XP + (PATCH ? (prow >> 4) * HP : prow) * 32 + lane * 8;
    const int zaddr = ZOFF + lane * 8;
    const int pix_l = prow + (lane >> 2);  // the lane's pixel inside a 16-pixel block, + the first pixel row of the wave's share

    f32x4 acc[9][5];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[tp][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stage 0 (PATCH: stages 0 .. 2; every wave issues the same number of DMA instructions per stage -- 5 for waves 0 - 3, 4 for
    //      waves 4 - 7 --, so "all but the last two stages' instructions have landed" is a constant vmcnt)
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_dma(i, 0, 0);
    if (PATCH) {
#pragma unroll
        for (int b = 1; b < AHEAD; ++b)
#pragma unroll
            for (int i = 0; i < NDY + 2; ++i) issue_dma(i, b, b);
        if (nst >= AHEAD) {  // AHEAD stages in flight: all but the last AHEAD - 1 of them must have landed
            if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * (KW + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * KW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int st = 0; st < nst; ++st) {
        const int cur = PATCH ? st % NBUF : (st & 1);
        const int nxt = PATCH ? (st + AHEAD) % NBUF : (cur ^ 1);   // buffer of the stage whose DMA is issued during this one
        const int ahead = PATCH ? AHEAD : 1;
        int vy = yb0 + cur * STAGE;
        int vx[3];
        vx[0] = xb0 + cur * STAGE;
        vx[1] = vx[0] + (PATCH ? HP : W) * 32;
        vx[2] = vx[1] + (PATCH ? HP : W) * 32;
        asm volatile("" : "+v"(vy), "+v"(vx[0]), "+v"(vx[1]), "+v"(vx[2]));
        const int pst = pbeg + st * WH_P + pix_l;
        wh_static_for(std::make_integer_sequence<int, KSL>{}, [&](auto KS) {
            constexpr int ksl = decltype(KS)::value;
            // validity of the lane's two pixels of this step: top / bottom / left / right neighbours inside the image
            uint64_t mt[2], mb[2], ml[2], mr[2];
#pragma unroll
            for (int hh = 0; hh < (PATCH ? 0 : 2); ++hh) {
                const unsigned p = (unsigned)(pst + ksl * 32 + hh * 16);
                const unsigned r = __umulhi(p, a.magicW);
                const int xx = (int)(p - r * (unsigned)W);
                const int yy = (int)(r - __umulhi(r, a.magicH) * (unsigned)a.H);
                mt[hh] = __builtin_amdgcn_ballot_w64(yy > 0);
                mb[hh] = __builtin_amdgcn_ballot_w64(yy < a.H - 1);
                ml[hh] = __builtin_amdgcn_ballot_w64(xx > 0);
                mr[hh] = __builtin_amdgcn_ballot_w64(xx < W - 1);
            }
            auto x_addr = [&](auto TP, int hh) -> int {
                constexpr int tp = decltype(TP)::value;
                constexpr int ky = tp / 3, kx = tp % 3;
                if (tp == 4 || PATCH) return vx[ky];
                uint64_t m = ky == 0 ? mt[hh] : (ky == 2 ? mb[hh] : ~0ull);
                if (kx == 0) m = ky == 1 ? ml[hh] : (m & ml[hh]);
                if (kx == 2) m = ky == 1 ? mr[hh] : (m & mr[hh]);
                return wh_sel(m, vx[ky], zaddr);
            };
            // B fragments: 5 cout groups x this step's 32 pixels
            u32x2 blo[5], bhi[5];
            wh_static_for(std::make_integer_sequence<int, 5>{}, [&](auto J) {
                constexpr int j = decltype(J)::value;
                if ((WH_ABL & 16) && (st > 0 || ksl > 0)) return;
                blo[j] = wh_tr<(2 * ksl) * DYBLK + j * 512>(vy);
                bhi[j] = wh_tr<(2 * ksl + 1) * DYBLK + j * 512>(vy);
            });
            u32x2 alo[2], ahi[2];
            {
                const int a0 = x_addr(std::integral_constant<int, 0>{}, 0), a1 = x_addr(std::integral_constant<int, 0>{}, 1);
                alo[0] = wh_tr<(ksl * (PATCH ? 2 * HP : 32)) * 32>(a0);
                ahi[0] = wh_tr<(ksl * (PATCH ? 2 * HP : 32) + (PATCH ? HP : 16)) * 32>(a1);
            }
            wh_static_for(std::make_integer_sequence<int, 9>{}, [&](auto TP) {
                constexpr int tp = decltype(TP)::value;
                constexpr int cs = tp & 1, ns = cs ^ 1;
                if constexpr (tp + 1 < 9 && !(WH_ABL & 8)) {
                    constexpr int kx = (tp + 1) % 3;
                    const int a0 = x_addr(std::integral_constant<int, tp + 1>{}, 0);
                    const int a1 = x_addr(std::integral_constant<int, tp + 1>{}, 1);
                    alo[ns] = wh_tr<(ksl * (PATCH ? 2 * HP : 32) + kx) * 32>(a0);
                    ahi[ns] = wh_tr<(ksl * (PATCH ? 2 * HP : 32) + (PATCH ? HP : 16) + kx) * 32>(a1);
                }
                // the next stage's operands, one DMA instruction per tap of the first two steps
                if constexpr (ksl == 0) issue_dma(tp, st + ahead, nxt);
                if constexpr (ksl == 1 && tp == 0) issue_dma(9, st + ahead, nxt);
                if constexpr (tp == 0) wh_wait_b<2>(blo, bhi, alo[0], ahi[0]);
                else if constexpr (tp + 1 < 9) wh_wait<2>(alo[cs], ahi[cs]);
                else wh_wait<0>(alo[cs], ahi[cs]);
                const u32x4 av{alo[cs][0], alo[cs][1], ahi[cs][0], ahi[cs][1]};
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const u32x4 bv{blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                    if (!(WH_ABL & 4)) wh_mfma<DT>(av, bv, acc[tp][j]);
                    else asm volatile("" : "+v"(acc[tp][j]) : "v"(av), "v"(bv));
                }
            });
        });
        // the next stage's operands must have landed (PATCH: everything but the AHEAD - 1 stages issued last, when one was issued in this stage)
        if (PATCH && st + AHEAD < nst) {
            if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * (KW + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * KW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(WH_ABL & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (NCI == 2) {
        // ---- waves 4..7 hand their partial tiles to waves 0..3 through LDS (two rounds: 23 + 22 tiles of 1 KiB per wave pair);
        //      NARROW: then waves 2, 3 to waves 0, 1 the same way
#pragma unroll
        for (int lvl = 0; lvl < (NARROW ? 2 : 1); ++lvl)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int t0 = rd * 23, t1 = rd ? 45 : 23;
            const bool snd = lvl == 0 ? half == 1 : (wave == 2 || wave == 3);
            const bool rcv = lvl == 0 ? half == 0 : wave < 2;
            unsigned char* slot = smem + ((lvl == 0 ? wave & 3 : wave & 1) * 23) * 1024 + lane * 16;
            if (snd) {
#pragma unroll
                for (int tl = 0; tl < 45; ++tl)
                    if (tl >= t0 && tl < t1) *reinterpret_cast<f32x4*>(slot + (tl - t0) * 1024) = acc[tl / 5][tl % 5];
            }
            __syncthreads();
            if (rcv) {
#pragma unroll
                for (int tl = 0; tl < 45; ++tl)
                    if (tl >= t0 && tl < t1) acc[tl / 5][tl % 5] += *reinterpret_cast<const f32x4*>(slot + (tl - t0) * 1024);
            }
            __syncthreads();
        }
    }
    // ---- partial slab: C[ci][co] tiles -> ws[split][co][tap*Cs + ci], 4 consecutive cins per lane (not the cins beyond Cs)
    if ((NARROW ? wave < 2 : half == 0) && i0 + wci * 16 < a.Cs) {
        float* wsp = a.ws + (int64_t)slab_id * a.Cd_pad * a.Kp;
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

// =====================================================================================================================================
// 1x1 layers: dW[co][ci] = sum_p dY[p][co] * X[p][ci] -- the same machinery without halo, taps or masks. The nine "taps" of the register
// tile become nine 16-cin planes: a workgroup owns 160 couts x 288 cins (wave: 80 couts x 144 cins = 45 accumulator tiles) and streams
// 64-pixel stages (dY 20 KB + X 36 KB, double buffered); waves 0-3 reduce pixels 0..31 of a stage, waves 4-7 pixels 32..63, summed through
// LDS at the end. Slab: ws[split][co][n_iblk * 288] (cins beyond Cs hold zeros; the reduction never reads them).
// =====================================================================================================================================
constexpr int WG_P = 64;
constexpr int WG_DYB = 20 * 1024;
constexpr int WG_NPL = 18;  // plane = tp * 2 + wci
constexpr int WG_STAGE = WG_DYB + WG_NPL * WG_P * 32;

template <int DT>
__global__ __launch_bounds__(512, 2) void wgrad_gemm_kernel(const WHArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = wave >> 2, wco = (wave >> 1) & 1, wci = wave & 1;
    const int q = lane >> 4, li = lane & 15;
    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, qq = nwg >> 3, rr = nwg & 7, j = b >> 3;
        L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + j;
    }
    const int ntiles = a.n_cblk * a.n_iblk;
    const int split = L / ntiles;
    const int tile = L - split * ntiles;
    const int cblk = tile / a.n_iblk, iblk = tile - cblk * a.n_iblk;
    const int c0 = cblk * 160, i0 = iblk * 16 * WG_NPL;
    const int pbeg = split * a.chunk;
    const int pend = min(pbeg + a.chunk, a.M);
    const int nst = pbeg < pend ? (pend - pbeg + WG_P - 1) / WG_P : 0;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    // DMA duties (1 KiB per instruction): dY image di = wave + 8*idx < 20 = (16-pixel block di/5, cout pair di%5), lanes as in the 3x3
    // kernel; X piece xi = wave + 8*idx < 36 = (plane xi>>1, 32-row block xi&1), lane -> (row lane>>1, 16-byte half lane&1)
    const int ldyB = a.dy_ld * 2, ldxB = a.src_ld * 2;
    const int yco_l = (lane >> 5) * 16 + (lane & 1) * 8;
    const unsigned ydl = (unsigned)(((lane >> 1) & 15) * ldyB + yco_l * 2);
    const unsigned xdl = (unsigned)((lane >> 1) * ldxB + (lane & 1) * 16);
    auto issue_dma = [&](int idx, int st, int buf) {  // idx 0..2: dY images, 3..7: X pieces
        if (st >= nst) return;
        unsigned char* base = smem + buf * WG_STAGE;
        const int pb0 = pbeg + st * WG_P;
        if (idx < 3) {
            const int di = wave + 8 * idx;
            if (di < 20) {
                const int pb = di / 5, cp = di - pb * 5;
                const int cobase = c0 + cp * 32;
                const unsigned sc = (unsigned)((pb0 + pb * 16) * ldyB + (a.dy_coff + cobase) * 2);
                const unsigned v = yco_l < a.Cd - cobase ? ydl + sc : WH_SENT;
                wh_dma16(rs_y, v, base + di * 1024);
            }
        } else {
            const int xi = wave + 8 * (idx - 3);
            if (xi < 2 * WG_NPL) {
                const int pl = xi >> 1, blk = xi & 1;
                const int cb = i0 + pl * 16;
                const unsigned sc = (unsigned)((pb0 + blk * 32) * ldxB + (a.src_coff + cb) * 2);
                wh_dma16(rs_x, cb < a.Cs ? xdl + sc : WH_SENT, base + WG_DYB + pl * (WG_P * 32) + blk * 1024);
            }
        }
    };
    const int yb0 = wco * 2560 + lane * 8 + half * 10240;
    const int xb0 = WG_DYB + wci * (WG_P * 32) + half * 1024 + lane * 8;

    f32x4 acc[9][5];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[tp][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_dma(i, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        int vy = yb0 + cur * WG_STAGE, vx = xb0 + cur * WG_STAGE;
        asm volatile("" : "+v"(vy), "+v"(vx));
        u32x2 blo[5], bhi[5];
        wh_static_for(std::make_integer_sequence<int, 5>{}, [&](auto J) {
            constexpr int j = decltype(J)::value;
            blo[j] = wh_tr<j * 512>(vy);
            bhi[j] = wh_tr<5120 + j * 512>(vy);
        });
        u32x2 alo[2], ahi[2];
        alo[0] = wh_tr<0>(vx);
        ahi[0] = wh_tr<512>(vx);
        wh_static_for(std::make_integer_sequence<int, 9>{}, [&](auto TP) {
            constexpr int tp = decltype(TP)::value;
            constexpr int cs = tp & 1, ns = cs ^ 1;
            if constexpr (tp + 1 < 9) {
                alo[ns] = wh_tr<(tp + 1) * 2 * WG_P * 32>(vx);
                ahi[ns] = wh_tr<(tp + 1) * 2 * WG_P * 32 + 512>(vx);
            }
            if constexpr (tp < 8) issue_dma(tp, st + 1, cur ^ 1);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {  // waves 4..7 -> waves 0..3 (23 + 22 tiles of 1 KiB per wave pair)
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
    if (half == 0) {
        float* wsp = a.ws + (int64_t)split * a.Cd_pad * a.Kp;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int co = c0 + (wco * 5 + j) * 16 + li;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int k = i0 + (tp * 2 + wci) * 16 + 4 * q;
                *reinterpret_cast<f32x4*>(wsp + (int64_t)co * a.Kp + k) = acc[tp][j];
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool wgrad_halo_plan(const cdet_conv_desc* d, WgradHaloPlan* out, int n_items) {
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1)) return false;
    if (d->Hs != d->Hd || d->Ws != d->Wd) return false;
    const bool narrow = d->Cd <= 96;  // 80-cout tile (the Cout-80 layers: C2f 160x160 Bottlenecks, the box-branch stems of the heads)
    if (narrow ? (d->Cs % 16 != 0 || d->Cd < 64) : (d->Cs % 32 != 0 || d->Cd < 128)) return false;
    if (d->Cs * 10 < (d->Cs + 31) / 32 * 32 * 7) return false;  // last cin tile more than 30 % empty
    if (!(d->dtype == CDET_BF16 || d->dtype == CDET_F16)) return false;
    int force_nci = 0;
    if (sw_is(SW_WGRAD_HALO)) {  // 0: im2col kernel (the tests compare the two); 4: 64-cin tile where it fits
        force_nci = sw(SW_WGRAD_HALO);
        if (force_nci == 0) return false;
    }
    const int64_t M = (int64_t)d->N * d->Hs * d->Ws;
    // maps that split into 8 x 16 patches: the patch form of the 32-cin tiles (four / three stage buffers; CDET_WGRAD_PATCH=0 keeps the linear
    // form -- the tests compare the two; 1: the 80-cout tile only)
    bool patch = force_nci != 4 && d->Hs % 8 == 0 && d->Ws % 16 == 0;
    if (sw_is(SW_WGRAD_PATCH)) patch = patch && sw(SW_WGRAD_PATCH) != 0 && (narrow || sw(SW_WGRAD_PATCH) != 1);
    const int XH = patch ? 192 : (WH_P + 2 * (d->Ws + 1) + 31) / 32 * 32;
    auto lds_of = [&](int nci) {
        return patch ? (narrow ? 4 * (size_t)(8 * 3072 + nci * XH * 32) : 3 * (size_t)(WH_DYB + nci * XH * 32))
                     : 2 * (size_t)(WH_DYB + nci * XH * 32) + WH_ZERO;
    };
    // The 64-cin tile halves the L2 -> LDS traffic per flop and needs no half-sum (main loop 4 % faster on 40x40x320), but every
    // workgroup then writes a slab twice the size: with one workgroup per CU the split-K workspace is 256 x (tile bytes) whatever
    // the layer, 47 MB at 32 cins against 94 MB -- measured 0.096 vs 0.105 ms per launch (profiles/r02_wgrad_kstats.txt). The
    // 64-cin form is kept for CDET_WGRAD_HALO=4 (parity-tested): it is the right tile once launches are grouped and S drops to 1.
    int nci = 2;
    if (!narrow && force_nci == 4 && d->Cs % 64 == 0 && lds_of(4) <= 160 * 1024) nci = 4;
    if (lds_of(nci) > 160 * 1024) return false;
    if (!patch && nci * (XH / 32) > (narrow ? 32 : 40)) return false;  // 5 X pieces per wave (4 for the 80-cout form: one reduction step per stage)
    if (M * d->src_ld * 2 >= 0xC0000000ll || M * d->dst_ld * 2 >= 0xC0000000ll) return false;
    if (M >= (1 << 24)) return false;  // the magic-multiply quotients are exact far beyond this; keep a wide margin
    const int co_tile = narrow ? 80 : 160;
    const int n_cblk = div_up(d->Cd, co_tile), n_iblk = div_up(d->Cs, 16 * nci);
    const int tiles = n_cblk * n_iblk * (n_items < 1 ? 1 : n_items);  // a grouped launch spreads the CUs over all its layers
    if (n_cblk * n_iblk > 256) return false;
    // one workgroup per CU: the largest split count that keeps the grid within one round; every split costs a slab write + re-read
    int S = 256 / tiles;
    const int maxS = (int)((M + 511) / 512);  // at least 512 pixels (4 stages) per split
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    int chunk = (int)((M + S - 1) / S);
    chunk = (chunk + WH_P - 1) / WH_P * WH_P;
    S = (int)((M + chunk - 1) / chunk);
    out->S = S;
    out->chunk = chunk;
    out->Kp = 9 * d->Cs;
    out->Cd_pad = n_cblk * co_tile;
    out->n_cblk = n_cblk;
    out->n_iblk = n_iblk;
    out->XH = XH;
    out->nci = nci;
    out->narrow = narrow ? 1 : 0;
    out->patch = patch ? 1 : 0;
    out->lds = lds_of(nci);
    return true;
}

template <int DT, int NCI, bool NARROW = false, bool PATCH = false>
static void wgrad_halo_launch_t(const WHArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)wgrad_halo_kernel<DT, NCI, NARROW, PATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_halo_kernel<DT, NCI, NARROW, PATCH>), dim3(grid), dim3(512), lds, s, a);
}

int wgrad_halo_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s,
                      const cdet_wgrad_item* items_dev, int n_items) {
    WHArgs a;
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.ws = ws;
    a.items = items_dev;
    a.H = d->Hs; a.W = d->Ws; a.M = d->N * d->Hs * d->Ws;
    a.Cs = d->Cs; a.Cd = (d->Cd + 7) / 8 * 8;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dy_ld = d->dst_ld; a.dy_coff = d->dst_coff;
    a.Kp = p.Kp; a.Cd_pad = p.Cd_pad; a.chunk = p.chunk; a.S = p.S; a.n_cblk = p.n_cblk; a.n_iblk = p.n_iblk; a.XH = p.XH;
    a.magicW = (unsigned)((1ull << 32) / (unsigned)a.W + 1);
    a.magicH = (unsigned)((1ull << 32) / (unsigned)a.H + 1);
    a.tiles_x = a.tiles_per_img = 0;
    if (p.patch) {
        a.tiles_x = a.W / 16;
        a.tiles_per_img = (a.H / 8) * a.tiles_x;
        a.magicW = (unsigned)((1ull << 32) / (unsigned)a.tiles_x + 1);
        a.magicH = (unsigned)((1ull << 32) / (unsigned)a.tiles_per_img + 1);
    }
    a.x_bytes = (unsigned)((int64_t)a.M * d->src_ld * 2);
    a.dy_bytes = (unsigned)((int64_t)a.M * d->dst_ld * 2);
    const int grid = (items_dev ? n_items : 1) * p.S * p.n_cblk * p.n_iblk;
    if (d->dtype == CDET_BF16) {
        if (p.patch && p.narrow) wgrad_halo_launch_t<CDET_BF16, 2, true, true>(a, grid, p.lds, s);
        else if (p.patch) wgrad_halo_launch_t<CDET_BF16, 2, false, true>(a, grid, p.lds, s);
        else if (p.narrow) wgrad_halo_launch_t<CDET_BF16, 2, true>(a, grid, p.lds, s);
        else if (p.nci == 4) wgrad_halo_launch_t<CDET_BF16, 4>(a, grid, p.lds, s);
        else wgrad_halo_launch_t<CDET_BF16, 2>(a, grid, p.lds, s);
    } else {
        if (p.patch && p.narrow) wgrad_halo_launch_t<CDET_F16, 2, true, true>(a, grid, p.lds, s);
        else if (p.patch) wgrad_halo_launch_t<CDET_F16, 2, false, true>(a, grid, p.lds, s);
        else if (p.narrow) wgrad_halo_launch_t<CDET_F16, 2, true>(a, grid, p.lds, s);
        else if (p.nci == 4) wgrad_halo_launch_t<CDET_F16, 4>(a, grid, p.lds, s);
        else wgrad_halo_launch_t<CDET_F16, 2>(a, grid, p.lds, s);
    }
    CDET_LAUNCH_CHECK();
    return 0;
}


bool wgrad_gemm_plan(const cdet_conv_desc* d, WgradHaloPlan* out) {
    if (!(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0)) return false;
    if (d->Hs != d->Hd || d->Ws != d->Wd) return false;
    // Both 1x1 kernels are bound by the L2 -> LDS stream, not by the MFMA pipe: 160 x 288 tiles move 56 KB per 5.9 MFLOP (103 FLOP/B),
    // measured 6.6 TB/s = 690 TF/s in the main loop (tools/kstats.sh, 40x40 2560->640: 242 us against 309 us for the im2col kernel).
    // Measured per shape at batch 32 (profiles/r02_conv_shapes.txt): 3+ cout blocks (Cout 640) 0-20 % faster, Cout 320 at 80x80
    // 15 % slower (1280 / 960 cins), Cout 160 level -- so only the wide layers come here.
    if (d->Cs % 16 != 0 || d->Cd < 128) return false;
    if (!(d->dtype == CDET_BF16 || d->dtype == CDET_F16)) return false;
    bool wide_only = true;
    if (sw_is(SW_WGRAD_HALO)) {  // 0: im2col kernel everywhere; 3: this kernel for every Cout >= 128 (tests)
        if (sw(SW_WGRAD_HALO) == 0) return false;
        wide_only = sw(SW_WGRAD_HALO) != 3;
    }
    if (wide_only && d->Cd <= 320) return false;
    const int64_t M = (int64_t)d->N * d->Hs * d->Ws;
    if (M * d->src_ld * 2 >= 0xC0000000ll || M * d->dst_ld * 2 >= 0xC0000000ll) return false;
    const int tile_ci = 16 * WG_NPL;
    const int n_cblk = div_up(d->Cd, 160), n_iblk = div_up(d->Cs, tile_ci);
    if (d->Cs * 10 < n_iblk * tile_ci * 7) return false;  // more than 30 % of the cin tile empty: the im2col kernel's 128-wide K blocks fit better
    const int tiles = n_cblk * n_iblk;
    if (tiles > 256) return false;
    int S = 256 / tiles;
    const int maxS = (int)((M + 511) / 512);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    int chunk = (int)((M + S - 1) / S);
    chunk = (chunk + WG_P - 1) / WG_P * WG_P;
    S = (int)((M + chunk - 1) / chunk);
    out->S = S;
    out->chunk = chunk;
    out->Kp = n_iblk * tile_ci;
    out->Cd_pad = n_cblk * 160;
    out->n_cblk = n_cblk;
    out->n_iblk = n_iblk;
    out->XH = 0;
    out->nci = 2;
    out->narrow = 0;
    out->patch = 0;
    out->lds = 2 * (size_t)WG_STAGE;
    return true;
}

template <int DT>
static void wgrad_gemm_launch_t(const WHArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)wgrad_gemm_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_gemm_kernel<DT>), dim3(grid), dim3(512), lds, s, a);
}

int wgrad_gemm_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s) {
    WHArgs a;
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.ws = ws;
    a.items = nullptr;
    a.H = d->Hs; a.W = d->Ws; a.M = d->N * d->Hs * d->Ws;
    a.Cs = d->Cs; a.Cd = (d->Cd + 7) / 8 * 8;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dy_ld = d->dst_ld; a.dy_coff = d->dst_coff;
    a.Kp = p.Kp; a.Cd_pad = p.Cd_pad; a.chunk = p.chunk; a.S = p.S; a.n_cblk = p.n_cblk; a.n_iblk = p.n_iblk; a.XH = 0;
    a.magicW = a.magicH = 0;
    a.x_bytes = (unsigned)((int64_t)a.M * d->src_ld * 2);
    a.dy_bytes = (unsigned)((int64_t)a.M * d->dst_ld * 2);
    const int grid = p.S * p.n_cblk * p.n_iblk;
    if (d->dtype == CDET_BF16) wgrad_gemm_launch_t<CDET_BF16>(a, grid, p.lds, s);
    else wgrad_gemm_launch_t<CDET_F16>(a, grid, p.lds, s);
    CDET_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdet
