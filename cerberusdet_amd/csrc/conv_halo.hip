// Tap-resident implicit-GEMM convolution for gfx950 (bf16 / f16 in, fp32 accumulate): stride-1 3x3 and 1x1, FWD and (through a
// tap-flipped, transposed weight pack) the stride-1 data gradient.
//
// What round 1 measured on the register/LDS-staged kernels of conv_igemm.hip (profiles/r01_conv_ablation.txt): the
// MFMAs alone and the L2 -> LDS operand stream alone each take ~2/3 of the full kernel's time -- a 160 x 128 tile re-fetches every
// activation pixel once per tap (9x for a 3x3) and moves 36.9 KB per 2.6 MFLOP. This kernel removes the tap redundancy instead of
// hiding it:
//   * the K loop runs CHUNK-outer / TAP-inner: for one 32-channel chunk the block's pixels AND THEIR HALO are staged in LDS once
//     and serve all nine taps. The pixel tile is 256 consecutive pixels of the flattened (n, y, x) index, so its halo is simply the
//     linear range [p0 - W - 1, p0 + 256 + W + 1): tap (dy, dx) of tile pixel i is halo row i + (W + 1) + dy * W + dx. Taps that leave
//     the image are redirected per lane to a zero row (9-bit validity mask per pixel), which also covers tiles that straddle
//     images. X traffic per tap step: 338 rows / 9 instead of 256 rows (W = 40) -> L2 -> LDS bytes per FLOP fall 3.3x;
//   * the weight operand is pre-packed by cdet_pack_weights_tiled as one contiguous, pre-swizzled 10 KiB tile per (cout block, chunk,
//     tap): its LDS image is a plain linear copy (buffer_load ... lds, no per-lane address arithmetic, perfectly coalesced);
//   * MFMA 32x32x16 (the shape that reaches the 2.5 PF/s peak; 16x16x32 tops out ~15 % lower): a wave owns 64 pixels x 160 couts
//     = 2 x 5 tiles (160 fp32 accumulators), a block = 4 waves = 256 pixels x 160 couts, two blocks per CU (76 KiB LDS each) so
//     one block's barrier / fragment-read bubbles run under the other's MFMAs;
//   * 3-stage weight ring with counted vmcnt (never 0 in the loop when LDS allows 3 stages), one s_barrier per 32-deep K step
//     (20 MFMAs of 32 cycles per wave between barriers).
// LDS rows are 64 B (32 channels); 16-B slot s of row r sits at physical slot s ^ ((r >> 2) & 3), which makes every
// ds_read_b128 lane group (MI355X_MICROARCH.md LDS table) hit 16 distinct slots of the 256-B bank row.
#include <stdlib.h>

#include "halo_common.h"
#include "bn_fold.h"

#include <type_traits>

namespace cdet {

struct HaloArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* bias;
    const uint16_t* res;
    void* y;
    float* stats;
    const BnFold* fold;  // train form: the partial sums are reduced and finalized inside this launch (bn_fold.h); null: the caller runs bn_finalize
    int H, W, Cd;
    int M;  // N*H*W
    int src_ld, src_coff, dst_ld, dst_coff, res_ld, res_coff;
    int nchunk;  // ceil(Cs / 32)
    int Cs;      // reduction channels (a last partial chunk is zero-filled: multiples of 8)
    int n_pblk, n_cblk;
    int act;
    int XH;  // halo rows per chunk buffer (multiple of 16)
    int tiles_x, tiles_per_img;  // patch mode: 16x16 pixel patches per image row / per image
    unsigned x_bytes, w_bytes;
    int wts, wt0;  // weight tile of K step `step` = tile step * wts + wt0 of the packed operand (1, 0: the operand's own order; 9, 4: the centre
                   // tap of a 9-tap operand -- the (0, 0) parity class of a stride-2 data gradient, conv_vt.hip)
    int Hd, Wd, cp, cq;  // OMAP: tile pixel (n, y, x) is written to pixel (2y + cp, 2x + cq) of an Hd x Wd destination
    int halfk;  // 3x3 only: the last 32-channel chunk holds at most 16 channels (Cs = 80): its second k16 half is all zeros and is skipped
    int accum;  // HEPI_F32 only: y (fp32) += result
    CatSrcs cat;  // CAT instantiations (1x1 only): the source is a virtual Concat of up to three buffers (halo_common.h)
    int ks_region;  // KS = 2: bytes of LDS of one K half (zero row + pixel buffers + weight ring)
};

constexpr int HEPI_STAGE_OFF = 6912;  // epilogue LDS map (after HZERO): statistics scratch [4][2][HC] fp32, scale/bias [2][HC] fp32, then the store staging

#ifdef CDET_PROFILING
// per-workgroup timeline for tools/halo_timeline.py: [xcc id, hw id, t_start, t_loop, t_epilogue, t_end] (s_memtime clocks)
__device__ unsigned long long* g_halo_dbg = nullptr;
#define CDET_HALO_STAMP(slot)                                                                                   \
    do {                                                                                                        \
        if (g_halo_dbg != nullptr && threadIdx.x == 0) g_halo_dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define CDET_HALO_STAMP(slot) \
    do {                      \
    } while (0)
#endif

// NT: taps (1 or 9). NF: 32-cout fragments per wave (5 -> 160 couts per block, 3 -> 96 for the 80-channel layers). NSW: weight ring
// stages (2 or 3). PATCH: the pixel tile is a 16 x 16 patch (halo 18 x 18 = 324 rows; maps whose sides are multiples of 16) instead
// of 256 consecutive pixels (halo 256 + 2W + 2 rows) -- half the halo on the 80-wide maps, and the only form that fits for 160-wide.
// ABL (timing experiments only, -DCDET_PROFILING): 1 = no DMA in the loop, 2 = no fragment reads, 4 = no MFMA, 8 = no epilogue
// stores, 16 = no K loop, 32 = plus the arithmetic of an in-LDS BatchNorm + SiLU rewrite of one pixel piece per K step (round-6 pricing experiment)
// NG = 4 (round 5): the 512-pixel tile. ONE workgroup per CU, one wave per SIMD with all 512 registers of a lane: a wave owns 128 pixels x 160 couts =
// 20 accumulator tiles (16 of them in AccVGPRs). Per k16 it reads (128 + 160) x 32 B of fragments for 20 MFMAs -- 461 B per MFMA against the 717 B of
// the 64 x 160 wave tile -- and the workgroup's weight ring serves twice the pixels: LDS reads per FLOP -36 %, weight LDS-DMA per FLOP -50 %, one
// barrier per 40 MFMAs instead of per 20. (Measured on the 256-pixel form, profiling build, 40 x 40 320 -> 320: the K loop takes 98 us, its MFMAs alone
// 55, everything but the MFMAs alone 48 -- the two do not overlap; tools/conv_tiled_bench.py with CDET_HALO_ABLATE.)
// KS = 2 (round 5, late): the half-tile form with the K loop SPLIT INSIDE THE WORKGROUP. The 128-pixel launches (20 x 20 maps at batch 32: 200 workgroups, at
// most one per CU) are a single 90-step dependent chain per workgroup with one wave per SIMD -- nothing covers a wave's DMA issue cost or its barrier
// waits (34 us for 0.015 TFLOP). Eight waves: waves 0-3 run the first half of the channel chunks, waves 4-7 the second half of the SAME pixels and couts
// in an LDS region of their own (two waves per SIMD: each covers the other's bubbles, and the chain is half as long); at the end the upper half hands
// its accumulators over through LDS and exits (a finished wave no longer takes part in s_barrier), the lower half adds them and runs the unchanged
// epilogue. Sum order: (first half's chunks in order) + (second half's chunks in order) -- deterministic, not the one-chain order.
template <int DT, int NT, int NF, int EPI, int NSW, bool PATCH, int NG = 2, int ABL = 0, bool OMAP = false, bool CAT = false, int KS = 1>
__global__ __launch_bounds__(KS == 2 ? 512 : 256, (NG == 4 || KS == 2) ? 1 : ((PATCH && NSW == 2) ? 3 : 2)) void conv_halo_kernel(const HaloArgs a) {
    static_assert(KS == 1 || (KS == 2 && NG == 1 && NT == 9 && NSW == 3 && !PATCH && !OMAP && !CAT && ABL == 0), "the in-workgroup K split: 3x3 half tiles only");
    // TRI (round 5, CDET_HALO_WG3=1): THREE workgroups per CU for the 96-cout patch form (the 80-channel layers of the 160 x 160 stage, where a
    // workgroup's life is a chain of latencies -- tile fetch, 27 short K steps, statistics, store -- and two resident workgroups leave the MFMA pipe idle
    // 65 % of the time). 160 KB / 3 in the LDS allocation granule of gfx950 (1280 B) = 42 granules = 53 760 B: two 18 x 18-row pixel buffers without
    // padding (2 x 20 736 B; the 21st DMA piece of a buffer is four rows = lanes 0-15 only), a two-stage weight ring (2 x 6 144 B), and NO zero row -- a
    // patch's halo rows are all real neighbours or DMA-filled padding, no tap is ever redirected. <= 168 registers per lane (launch bounds).
    constexpr bool TRI = PATCH && NSW == 2;
    static_assert(!TRI || (NF == 3 && NG == 2 && NT == 9 && !OMAP && !CAT), "three workgroups per CU: the 96-cout 16 x 16 patch form only");
    constexpr int HZ = TRI ? 0 : HZERO;  // bytes in front of the pixel buffers
    static_assert(!CAT || (NT == 1 && !PATCH && !OMAP), "a virtual-Concat source is a 1x1 convolution's");
    static_assert(NG != 4 || (NT == 9 && !PATCH && !OMAP && !CAT && NSW == 6), "the 512-pixel tile: 3x3, linear halo, six-stage weight ring");
    static_assert(NSW == 2 || NSW == 3 || (NSW == 6 && NT == 9), "weight ring depths");
    constexpr int XPS = NG >= 3 ? 2 : 1;  // pixel DMA pieces a wave issues per K step (taps 0..6 of a chunk fetch the next chunk's halo)
    constexpr int MAXXPK = TRI ? 6 : MAXXP * XPS;   // pieces per wave per chunk: XH <= 448 (896) rows; TRI: 21 pieces over four waves
    constexpr int HPB = 128 * NG;         // pixels per block: a wave owns NG 32-pixel fragments (NG = 1: half tiles for layers whose
                                          // 256-pixel tiles would leave most CUs idle, e.g. the 20 x 20 maps at batch 32)
    static_assert(NG == 2 || (NG == 3 && NF == 3) || !PATCH, "patch forms: 16 x 16 (256 pixels), or 24 rows x 16 columns (384 pixels) for the 96-cout tile");
    constexpr int PH = HPB / PATCH_W;     // patch rows: 16, or 24 in the NG = 3 form (round 5): the 80-channel layers of the 160 x 160 stage stream
                                          // 162 KB of weights per workgroup for 65 KB of pixels -- a 384-pixel patch (9 accumulator tiles per wave
                                          // = 144 registers, still two workgroups per CU) takes a third off the weight LDS-DMA per pixel, a fifth
                                          // off the fragment reads per MFMA, and puts 18 instead of 12 MFMAs behind every barrier. A map whose
                                          // height is not a multiple of 24 gets a last patch row that hangs over the bottom edge (160 = 6 x 24 +
                                          // 16: 5 % idle rows; they read zeros and are never stored)
    constexpr int HC = NF * 32;           // couts per block
    constexpr int WTILE = HC * HROW;      // bytes per (cblk, chunk, tap) weight tile: 10240 / 6144
    constexpr int WQ = WTILE / 4;         // bytes of the tile each wave copies: 2 (1) full 1-KiB pieces + one half piece (lanes 0-31)
    constexpr int NWP = (WQ + 1023) / 1024;  // DMA instructions per wave per tile: 3 / 2 -- the same for every wave
    constexpr int NXP1 = 2 * NG;          // 1x1: pixel DMA pieces per wave per chunk (HPB rows / 16 / 4 waves)
    constexpr int NM = NG * NF;           // MFMAs per phase (one k16 half of a step)
    constexpr int NR = NF + NG;           // fragment reads per phase (two per MFMA slot)
    static_assert(NM >= (NR + 1) / 2 + 1, "phase B needs a slot for the weight DMA in front of the fragment reads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int khalf = KS == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;  // which half of the K loop this wave runs
    unsigned char* const smem = smem_all + (KS == 2 ? khalf * a.ks_region : 0);              // ... in an LDS region of its own
    const int t = KS == 2 ? (int)threadIdx.x & 255 : (int)threadIdx.x;                       // thread / wave index inside the half
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int cbeg = KS == 2 ? khalf * (a.nchunk >> 1) : 0;                                  // first channel chunk of this half

    // XCD-aware remap (bijective): consecutive logical ids -- the cout blocks of one pixel tile, then the next pixel tile -- run on ONE XCD
    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int cblk = L % a.n_cblk;
    const int pblk = L / a.n_cblk;
    const int c0 = cblk * HC;
    const int W = a.W;
    const int XHB = a.XH * HROW;
    const int nsteps = a.nchunk * NT;
    // geometry of the pixel tile. linear: pixels p0 .. p0+255, halo row of tile pixel i = i + W + 1, tap pitch W.
    // patch: tile (n, ty, tx) = 16 x 16 pixels at (ty*16, tx*16); halo row of pixel (iy, ix) = (iy+1)*18 + ix+1, tap pitch 18.
    const int tpitch = NT == 9 ? (PATCH ? PATCH_HPW : W) : 0;
    const int halo0 = NT == 9 ? tpitch + 1 : 0;
    const int p0 = pblk * HPB;  // linear mode
    int pn = 0, py0 = 0, px0 = 0;  // patch mode: image, top-left pixel
    if (PATCH) {
        pn = pblk / a.tiles_per_img;
        const int r = pblk - pn * a.tiles_per_img;
        py0 = (r / a.tiles_x) * PH;
        px0 = (r % a.tiles_x) * PATCH_W;
    }
    unsigned char* const xbase = smem + HZ;
    constexpr int NXB = NT == 1 ? 3 : 2;  // pixel buffers
    unsigned char* const wbase = smem + HZ + NXB * XHB;

    if (!TRI && t < 16) reinterpret_cast<uint32_t*>(smem)[t] = 0u;  // zero row (visible after the first barrier)
#ifdef CDET_PROFILING
    if (g_halo_dbg != nullptr && t == 0) {
        g_halo_dbg[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        g_halo_dbg[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    }
#endif
    CDET_HALO_STAMP(2);

    // ---- X DMA pieces of this wave: piece id 4*i + wave covers halo rows 16*id .. 16*id+15, 4 lanes (64 B) per row --------------
    const int nxp_total = (a.XH + 15) >> 4;  // (TRI: 324 rows = 20 pieces + 4 rows)
    const int nxpw = (nxp_total - wave + 3) >> 2;  // wave-uniform
    unsigned xvoff[MAXXPK];
#pragma unroll
    for (int i = 0; i < MAXXPK; ++i) {
        const int hrow = 16 * (4 * i + wave) + (lane >> 2);
        int g;
        bool ok;
        if (PATCH) {
            const int hy = hrow / PATCH_HPW, hx = hrow - hy * PATCH_HPW;
            const int y = py0 - 1 + hy, x = px0 - 1 + hx;
            ok = hy < PH + 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)W;  // outside the image: zeros = the padding
            g = (pn * a.H + y) * W + x;
        } else {
            g = p0 - halo0 + hrow;
            ok = g >= 0 && g < a.M;
        }
        const unsigned off = ((unsigned)g * (unsigned)a.src_ld + (unsigned)a.src_coff) * 2u + ((unsigned)((lane & 3) ^ ((hrow >> 2) & 3)) << 4);
        xvoff[i] = ok ? off : HSENT;
    }
    // CAT: the same pieces inside segments 1 and 2 (segment 0 lives in xvoff)
    constexpr int NXC = CAT ? 2 * NG : 1;
    unsigned xv1[NXC], xv2[NXC];
    if (CAT) {
#pragma unroll
        for (int i = 0; i < NXC; ++i) {
            const int hrow = 16 * (4 * i + wave) + (lane >> 2);
            const int g = p0 + hrow;
            const bool ok = g < a.M;
            const unsigned slot = (unsigned)((lane & 3) ^ ((hrow >> 2) & 3)) << 4;
            xvoff[i] = ok ? cat_pixel_off(a.cat.up0, a.cat.ld0, a.cat.co0, a.cat.H, a.cat.W, g) + slot : HSENT;
            xv1[i] = (ok && a.cat.n > 1) ? cat_pixel_off(a.cat.up1, a.cat.ld1, a.cat.co1, a.cat.H, a.cat.W, g) + slot : HSENT;
            xv2[i] = (ok && a.cat.n > 2) ? cat_pixel_off(a.cat.up2, a.cat.ld2, a.cat.co2, a.cat.H, a.cat.W, g) + slot : HSENT;
        }
    }
    // CAT: the segment table as uniform VGPR values (see dma_x). Scalars, not arrays: a select over array elements became a
    // dynamically indexed private-memory array (scratch loads inside the K loop)
    unsigned cl0 = 0u, cl1 = 0u, cl2 = 0u, ch0 = 0u, ch1 = 0u, ch2 = 0u;
    int cat_c1 = 0x7fffffff, cat_c2 = 0x7fffffff;
    if (CAT) {
        cl0 = (unsigned)(uint64_t)a.cat.x0; ch0 = (unsigned)((uint64_t)a.cat.x0 >> 32);
        cl1 = (unsigned)(uint64_t)a.cat.x1; ch1 = (unsigned)((uint64_t)a.cat.x1 >> 32);
        cl2 = (unsigned)(uint64_t)a.cat.x2; ch2 = (unsigned)((uint64_t)a.cat.x2 >> 32);
        asm volatile("" : "+v"(cl0), "+v"(cl1), "+v"(cl2), "+v"(ch0), "+v"(ch1), "+v"(ch2));
        cat_c1 = a.cat.c1;
        cat_c2 = a.cat.c2;
    }
    const bool partial = (a.Cs & 31) != 0;                   // wave-uniform
    const int xls = (lane & 3) ^ ((lane >> 4) & 3);          // logical 16-byte slot this lane fetches (the same for every piece)
    // ---- W DMA: a plain linear copy of the packed tile; wave w copies bytes [w*WQ, (w+1)*WQ) -----------------------------------
    const unsigned wvoff = (unsigned)(wave * WQ + lane * 16);
    const unsigned wtile0 = (unsigned)cblk * (unsigned)(nsteps * (OMAP ? a.wts : 1)) * (unsigned)WTILE;

    // Steps / chunks beyond the end are requested through an EMPTY descriptor (every load returns zeros), so the number of DMA
    // instructions per step -- what the counted vmcnt waits rely on -- never changes.
    auto dma_w1 = [&](int step, int stage, int j) {  // piece j of this wave of the weight tile of K step `step` -> ring stage `stage`
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, step < nsteps ? (int)a.w_bytes : 0, 0x00020000);
        unsigned char* dst = wbase + stage * WTILE + wave * WQ + j * 1024;
        const unsigned soff = wtile0 + (OMAP ? (unsigned)(step * a.wts + a.wt0) : (unsigned)step) * (unsigned)WTILE;
        if (j < NWP - 1) dma16(rs, wvoff + (unsigned)j * 1024u, soff, dst);
        else if (j == NWP - 1 && lane < 32) dma16(rs, wvoff + (unsigned)j * 1024u, soff, dst);  // the 512-byte tail of the wave's share
    };
    // (the segment table is captured BY VALUE: through by-reference captures the selects below became selects of addresses inside a
    //  closure object kept in private memory)
    auto dma_x = [&, xvoff, xv1, xv2, cl0, cl1, cl2, ch0, ch1, ch2, cat_c1, cat_c2](int i, int chunk, int xb) {  // piece i of this wave, channels of `chunk` -> pixel buffer xb
        unsigned char* dst = xbase + xb * XHB + (4 * i + wave) * 1024;
        if constexpr (CAT) {  // the chunk's segment (wave-uniform): its buffer, its pixel offsets, the chunk's position inside it.
            // The segment table lives in VGPRs (uniform values, selected by v_cndmask and read back with v_readfirstlane): as scalars the
            // three pointers / extents / boundaries stayed live through the whole K loop, the kernel ran out of SGPRs, and the spill
            // code's scratch loads drained the DMA queue every step (3x slower than the plain form)
            const bool sg2 = chunk >= cat_c2, sg1 = chunk >= cat_c1;
            // every candidate is read into a register and made opaque BEFORE the select: a select of loads from the (by-value) closure is
            // rewritten by the optimiser into one load at a selected address, the closure then stays in private memory, and with it the
            // kernel arguments it refers to -- their scratch loads share the DMA's counter (the first CAT build ran 3x slower)
            unsigned l0 = cl0, l1 = cl1, l2 = cl2, h0 = ch0, h1 = ch1, h2 = ch2;
            unsigned t0 = xvoff[i], t1 = xv1[i < NXC ? i : 0], t2 = xv2[i < NXC ? i : 0];
            asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(t0), "+v"(t1), "+v"(t2));
            const unsigned plo = sg2 ? l2 : (sg1 ? l1 : l0);
            const unsigned phi = sg2 ? h2 : (sg1 ? h1 : h0);
            // (readfirstlane returns int: without the unsigned cast a low word with its top bit set sign-extends into the high word)
            const uint64_t pa = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane(plo) | ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane(phi) << 32);
            const int cf = sg2 ? cat_c2 : (sg1 ? cat_c1 : 0);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, chunk < a.nchunk ? (int)0xC0000000u : 0, 0x00020000);  // (every real offset lies inside its buffer -- host check; the extent only has to exclude the HSENT sentinel)
            unsigned v = sg2 ? t2 : (sg1 ? t1 : t0);
            if (partial && chunk * 32 + 8 * xls >= a.Cs) v = HSENT;
            dma16<CDET_HALO_X_AUX>(rs, v, (unsigned)(chunk - cf) * 64u, dst);
            return;
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, chunk < a.nchunk ? (int)a.x_bytes : 0, 0x00020000);
        unsigned v = xvoff[i] + (unsigned)chunk * 64u;
        // last partial chunk (Cs % 32 != 0, the 80-channel layers): the lanes whose 8-channel slot lies beyond Cs fetch zeros -- what
        // sits there in memory is a neighbouring channel slice (possibly never written), and 0-weight x NaN would still be NaN
        if (partial && chunk * 32 + 8 * xls >= a.Cs) v = HSENT;
        if (TRI && 4 * i + wave == 20) {  // the buffer's last four rows: the rest of the piece would land in the next buffer / the weight ring
            if (lane < 16) dma16<CDET_HALO_X_AUX>(rs, v, 0u, dst);
        } else {
            dma16<CDET_HALO_X_AUX>(rs, v, 0u, dst);
        }
    };

    auto dma_x_dead = [&](int i, int xb) {  // a prefetch behind the last chunk: same instruction, empty descriptor (the DMA writes zeros into the idle buffer)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
        unsigned char* dst = xbase + xb * XHB + (4 * i + wave) * 1024;
        if (TRI && 4 * i + wave == 20) {
            if (lane < 16) dma16<CDET_HALO_X_AUX>(rs, 0u, 0u, dst);
        } else {
            dma16<CDET_HALO_X_AUX>(rs, 0u, 0u, dst);
        }
    };

    // ---- prologue DMA: chunk 0 of X (1x1: chunks 0 and 1), weight tiles of steps 0 .. NSW-1. Issued HERE (round 6), in front of the tap-mask
    //      divisions and the accumulator initialisation below: that arithmetic (~700 VALU cycles) now runs under the fetch latency
#pragma unroll
    for (int i = 0; i < MAXXPK; ++i)
        if (i < nxpw) dma_x(i, cbeg, cbeg & 1);
#pragma unroll
    for (int j = 0; j < NWP; ++j) dma_w1(cbeg * NT, 0, j);
    if (NT == 1) {
#pragma unroll
        for (int i = 0; i < NXP1; ++i) dma_x(i, 1, 1);
    }
#pragma unroll
    for (int sg = 1; sg < NSW; ++sg) {
#pragma unroll
        for (int j = 0; j < NWP; ++j) dma_w1(cbeg * NT + sg, sg, j);
    }

    // ---- fragment read offsets -------------------------------------------------------------------------------------------------
    // A (weights): row f*32 + l31 of the stage, k-slot 2*s + h
    const int aoff0 = l31 * HROW + ((h ^ ((l31 >> 2) & 3)) << 4);
    // B (pixels): tile pixel wave*64 + g*32 + l31: its output pixel index, the halo row of its centre, validity bit per tap
    int pixh[NG], pout[NG];
    unsigned vmask[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = wave * (32 * NG) + g * 32 + l31;
        unsigned m = 0u;
        if (PATCH) {
            const int iy = i / PATCH_W, ix = i % PATCH_W;
            pixh[g] = iy * PATCH_HPW + ix + halo0;
            pout[g] = (pn * a.H + py0 + iy) * W + px0 + ix;
            // every halo row is the right neighbour or the zero padding; a pixel of a patch row that hangs over the bottom edge (NG = 3) takes the
            // zero row for all taps -- its window would reach the image's last row, and its accumulator must stay 0 for the BatchNorm partial sums
            m = (py0 + iy < a.H) ? 0x1ffu : 0u;
        } else {
            pixh[g] = i + halo0;
            const int p = p0 + i;
            pout[g] = p;
            if (OMAP) {  // destination pixel of the strided form (also addresses the residual = the gradient already there)
                const int n_ = p / (a.H * W), r_ = p - n_ * (a.H * W);
                const int y_ = r_ / W, x_ = r_ - y_ * W;
                pout[g] = p < a.M ? (n_ * a.Hd + 2 * y_ + a.cp) * a.Wd + 2 * x_ + a.cq : a.M;
            }
            if (p < a.M) {
                if (NT == 9) {
                    const int x = p % W;
                    const int y = (p / W) % a.H;
                    unsigned rb = 0u, cb = 0u;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if ((unsigned)(y + k - 1) < (unsigned)a.H) rb |= 1u << k;
                        if ((unsigned)(x + k - 1) < (unsigned)W) cb |= 1u << k;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if ((rb >> k) & 1u) m |= cb << (3 * k);
                } else {
                    m = 1u;
                }
            }
        }
        vmask[g] = m;
    }

    f32x16 acc[NF][NG];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;

    // (the prologue's DMA was issued above, in front of the index arithmetic)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NT == 1 ? NXP1 : 0) + (NSW - 1) * NWP) : "memory");  // tile 0 and chunk 0 have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // B-fragment byte offsets (relative to smem) of a K step: halo row of each of the lane's two pixels for the step's tap
    auto b_offsets = [&](int xoff, int tap_, int (&bo)[NG]) {
        const int dy_ = tap_ / 3 - 1, dx_ = tap_ % 3 - 1;
        // opaque copies: without them the compiler hoists the nine per-tap offset pairs (and their scalar parts) out of the chunk
        // loop as loop invariants -- 18 VGPRs + ~30 SGPRs the 256-register budget does not have (spills); recomputing costs 8 VALU
        int tp = tpitch, p_[NG];
        asm volatile("" : "+s"(tp));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            p_[g] = pixh[g];
            asm volatile("" : "+v"(p_[g]));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int hrow = p_[g] + (NT == 9 ? dy_ * tp + dx_ : 0);
            const int off = hrow * HROW + ((h ^ ((hrow >> 2) & 3)) << 4);
            const bool ok = TRI || ((vmask[g] >> tap_) & 1u);  // (TRI: a 16 x 16 patch of a map whose sides are multiples of 16 -- every tap is real)
            bo[g] = ok ? off + xoff : 0;  // invalid tap / pixel: the zero row
        }
    };
    // fragment i of a k16 half: i < 2 -> pixel rows (B operand), else weight rows (A operand); read order = use order
    auto frag = [&](const unsigned char* ws_, const int (&bo)[NG], int s_, int i, u32x4 (&af)[NF], u32x4 (&bf)[NG]) {
        if (ABL & 2) {
            if (i < NG) bf[i] = u32x4{(unsigned)lane, 1u, 2u, 3u};
            else af[i - NG] = u32x4{(unsigned)lane, 1u, 2u, 3u};
        } else if (i < NG) {
            bf[i] = *reinterpret_cast<const u32x4*>(smem + (bo[i] ^ (s_ << 5)));
        } else {
            af[i - NG] = *reinterpret_cast<const u32x4*>(ws_ + ((aoff0 ^ (s_ << 5)) + (i - NG) * 32 * HROW));
        }
    };

    CDET_HALO_STAMP(3);
#ifdef CDET_PROFILING
    unsigned long long t_lgkm = 0;
    unsigned long long t_wait = 0, t_bar = 0;  // clocks wave 0 spends in the mid-step counted wait / in the barrier behind it
#endif
    int bo_cur[NG], bo_nxt[NG];
    u32x4 a0[NF], b0[NG], a1[NF], b1[NG];
    b_offsets(HZ + (cbeg & 1) * XHB, 0, bo_cur);
#pragma unroll
    for (int i = 0; i < NR; ++i) frag(wbase, bo_cur, 0, i, a0, b0);

    // One K step. `u` is the step's position inside the unrolled group (3x3: the tap, 9 per chunk; 1x1: 3 chunks per group), a
    // compile-time constant after unrolling, so the tap offsets, the ring stage (NSW == 3) and the pixel-piece schedule fold away;
    // `chunk` is the step's channel chunk, `st` its index.
    auto step = [&](int st, int chunk, int u, auto HK) {
        constexpr bool halfk = decltype(HK)::value;  // the step's second k16 half holds zero channels only: no reads, no MFMAs for it
        // ring stage of this step's tile: compile-time for NSW == 3 (nine taps = three turns of the ring); NSW == 6 (the 512-pixel tile): st % 6 =
        // (3 chunk + u) % 6 -- one of two compile-time constants, selected by the chunk's parity (a scalar select)
        const int sc = NSW == 3 ? u % 3 : (NSW == 6 ? ((chunk & 1) ? (u + 3) % 6 : u % 6) : (st & 1));
        const int sn = NSW == 3 ? (u + 1) % 3 : (NSW == 6 ? ((chunk & 1) ? (u + 4) % 6 : (u + 1) % 6) : ((st + 1) & 1));
        const unsigned char* ws = wbase + sc * WTILE;
        const unsigned char* wsn = wbase + sn * WTILE;
        const int tapn = NT == 9 ? (u + 1) % 9 : 0;
        const int chunkn = NT == 9 ? chunk + (u == 8 ? 1 : 0) : chunk + 1;
        const int xbn = NT == 9 ? (chunkn & 1) : (u + 1) % 3;  // pixel buffer of step st+1
        // pixel pieces of the next chunk issued in this step's phase A (wave-uniform): pieces XPS u .. XPS u + XPS - 1 at taps u = 0..6
        const int na = (NT == 9 && u < MAXXP) ? min(max(nxpw - XPS * u, 0), XPS) : 0;
        const bool xa = na > 0;
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase A: MFMAs of (st, k16 #0); the fragment reads of (st, k16 #1), the address arithmetic of step st+1 and the pixel
        //      pieces of the next chunk (3x3: one per step; 1x1: the whole chunk st+2 into the third buffer) in their shadow
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (!(ABL & 4)) {
                if (NG == 4 && i < 16) mfma32a<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);  // (tiles 0..15 of the 512-pixel form: AccVGPRs)
                else mfma32<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);
            }
            // two fragment reads per MFMA slot: they are all in flight after the first half of the phase, so the lgkmcnt(0) at its
            // end waits for LDS latency that the second half has already covered (one read per slot left ~140 clocks per step exposed)
            if (!halfk && 2 * i < NR) frag(ws, bo_cur, 1, 2 * i, a1, b1);
            if (!halfk && 2 * i + 1 < NR) frag(ws, bo_cur, 1, 2 * i + 1, a1, b1);
            if (i == NM - 1) b_offsets(HZ + xbn * XHB, tapn, bo_nxt);
            if (!(ABL & 1)) {
                if (NT == 9) {
                    // (the half-k tail is the LAST chunk: its prefetches only keep the instruction count the waits rely on -- an empty descriptor at
                    //  offset 0, so that the pixel offsets need not outlive the main loop: they were spilled to scratch for the tail alone)
                    if (i == NM - 2 && u < MAXXP) {
                        if (xa) {
                            if (halfk) dma_x_dead(XPS * u, (chunk + 1) & 1);
                            else dma_x(XPS * u, chunk + 1, (chunk + 1) & 1);
                        }
                    }
                    if (XPS == 2 && i == NM - 3 && u < MAXXP) {
                        if (na > 1) {
                            if (halfk) dma_x_dead(XPS * u + 1, (chunk + 1) & 1);
                            else dma_x(XPS * u + 1, chunk + 1, (chunk + 1) & 1);
                        }
                    }
                } else if (i < NXP1) {
                    dma_x(i, chunk + 2, (u + 2) % 3);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile st+1 (and, at a chunk boundary, the next chunk's pixels) have landed -- the weight pieces issued in the previous phase B
        // and this phase's pixel pieces may stay in flight; nobody reads tile st's stage any more
#ifdef CDET_PROFILING
        const unsigned long long twl = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long tw0 = __builtin_readcyclecounter();
        t_lgkm += tw0 - twl;
#endif
        if (NT == 9) {
            // vmcnt retires in order: what must have landed is the weight tile of step st+1 (issued two phase-Bs ago); everything
            // issued after it may stay in flight -- the newest weight tile and the pixel pieces of this and the previous phase A
            // (pixel pieces are only issued at taps 0..6, so at tap 8, when phase B first reads the next chunk, none is left)
            // (a TWO-stage ring issues tile st+1 one phase B ago, i.e. BEHIND the previous phase A's pixel pieces: only this phase's may stay in flight.
            //  Until round 5 the two-stage form waited with the three-stage count -- up to one weight piece of tile st+1 could still be in flight at
            //  the barrier; never observed at two workgroups per CU, but it showed as wrong results as soon as three shared one)
            const int np = (NSW != 2 && u >= 1 && u - 1 < MAXXP) ? min(max(nxpw - XPS * (u - 1), 0), XPS) : 0;  // pieces of the previous step's phase A
            switch (na + np) {  // wave-uniform
                case 0: wait_vm_lgkm0<(NSW - 2) * NWP>(); break;
                case 1: wait_vm_lgkm0<(NSW - 2) * NWP + 1>(); break;
                case 2: wait_vm_lgkm0<(NSW - 2) * NWP + 2>(); break;
                case 3: wait_vm_lgkm0<(NSW - 2) * NWP + 3>(); break;
                default: wait_vm_lgkm0<(NSW - 2) * NWP + 4>(); break;
            }
        } else {
            wait_vm_lgkm0<(NSW - 2) * NWP + NXP1>();
        }
#ifdef CDET_PROFILING
        const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef CDET_PROFILING
        t_wait += tw1 - tw0;
        t_bar += __builtin_readcyclecounter() - tw1;
#endif
        if ((ABL & 32) && NT == 9 && u >= 2) {
            // PRICING EXPERIMENT (round 6, VERDICT r05 item 4, profiling builds only; results are wrong by design): what "normalise in LDS" would add to a K
            // step -- the wave rewrites ONE of its pixel pieces per step in place: a ds_read_b128 (8 values of one pixel), scale / shift / SiLU
            // (v_exp + v_rcp per value), pack, a ds_write_b128. Seven pieces per chunk = the 42 - 44 values per lane and chunk of the halo tile.
            unsigned char* pp_ = smem + HZ + ((chunk + 1) & 1) * XHB + (4 * (u - 2) + wave) * 1024 + lane * 16;
            u32x4 v4 = *reinterpret_cast<u32x4*>(pp_);
            float f8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f8[2 * r] = Elem<DT>::to_f32((uint16_t)(v4[r] & 0xffff));
                f8[2 * r + 1] = Elem<DT>::to_f32((uint16_t)(v4[r] >> 16));
            }
            const float ksc = 1.0009765625f + (float)(lane & 3) * 0.001f, ksh = 0.01f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float a_ = f8[r] * ksc + ksh;
                f8[r] = a_ * __builtin_amdgcn_rcpf(1.0f + __expf(-a_));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = hpack2<DT>(f8[2 * r], f8[2 * r + 1]);
            *reinterpret_cast<u32x4*>(pp_) = v4;
        }
        // ---- phase B: MFMAs of (st, k16 #1); DMA of the tile NSW steps ahead into the stage just freed and the fragment reads of
        //      (st+1, k16 #0) in their shadow
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (!(ABL & 4) && !halfk) {
                if (NG == 4 && i < 16) mfma32a<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
                else mfma32<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
            }
            if (!(ABL & 1)) {
                if (i == 0) dma_w1(st + NSW, sc, 0);
                if (i == (NM >= 10 ? 3 : (NM >= 6 ? 2 : 1))) dma_w1(st + NSW, sc, 1);
                if (i == (NM >= 10 ? 6 : (NM >= 6 ? 4 : 2))) dma_w1(st + NSW, sc, 2);
            }
            if (i >= 1 && 2 * (i - 1) < NR) frag(wsn, bo_nxt, 0, 2 * (i - 1), a0, b0);
            if (i >= 1 && 2 * (i - 1) + 1 < NR) frag(wsn, bo_nxt, 0, 2 * (i - 1) + 1, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) bo_cur[g] = bo_nxt[g];
    };

    if (ABL & 16) {
    } else if (NT == 9) {
        const int nfull = KS == 2 ? cbeg + (a.nchunk >> 1) : (a.halfk ? a.nchunk - 1 : a.nchunk);  // (KS = 2: an even number of full chunks -- host check)
        for (int chunk = cbeg; chunk < nfull; ++chunk) {
#pragma unroll
            for (int u = 0; u < 9; ++u) step(chunk * 9 + u, chunk, u, std::false_type{});
        }
        if (KS == 1 && a.halfk) {
#pragma unroll
            for (int u = 0; u < 9; ++u) step(nfull * 9 + u, nfull, u, std::true_type{});
        }
    } else {
        for (int c3 = 0; c3 < a.nchunk; c3 += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (c3 + u < a.nchunk) step(c3 + u, c3 + u, u, std::false_type{});
        }
    }
    CDET_HALO_STAMP(4);
#ifdef CDET_PROFILING
    if (g_halo_dbg != nullptr && t == 0) {
        g_halo_dbg[(size_t)blockIdx.x * 8 + 6] = t_wait;
        g_halo_dbg[(size_t)blockIdx.x * 8 + 1] = t_lgkm;  // (overwrites the hw id: not needed by this experiment)
        g_halo_dbg[(size_t)blockIdx.x * 8 + 7] = t_bar;
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");  // asm MFMAs are opaque to the hazard recogniser; trailing (dead) DMA drained
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (KS == 2) {
        // the upper half's accumulators -> LDS [register][256 lanes of the half] (conflict-free: consecutive lanes, consecutive words), then it is done
        float* const ex = reinterpret_cast<float*>(smem_all);
        if (khalf == 1) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[((f * NG + g) * 16 + r) * 256 + t] = acc[f][g][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (khalf == 1) return;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][g][r] += ex[((f * NG + g) * 16 + r) * 256 + t];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (the four remaining waves: the epilogue reuses this LDS)
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- BN statistics of the raw convolution (train mode): per (pixel block, channel) partial sums ---------------------------------
    if (a.stats != nullptr) {
        float* stl = reinterpret_cast<float*>(smem + HZERO);  // [4 waves][2][HC]
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float s16[16], q16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s_ = acc[f][0][r];  // (starting from the first fragment, not from 0.f: "0 + x" is an instruction under IEEE signed-zero rules)
                float q_ = s_ * s_;
#pragma unroll
                for (int g = 1; g < NG; ++g) {
                    const float v0 = acc[f][g][r];
                    s_ += v0;
                    q_ = fmaf(v0, v0, q_);
                }
                s16[r] = s_;
                q16[r] = q_;
            }
            tile_stats32(s16, q16, stl + (wave * 2 + 0) * HC + f * 32, stl + (wave * 2 + 1) * HC + f * 32, lane);
        }
        // (the four waves' rows are added up and stored at the very end of the kernel: no block-wide barrier between a wave's statistics and its
        //  output epilogue, and the short serial tail -- 160 threads, 8 LDS reads, 2 stores -- runs under the other waves' stores)
    }

    // ---- epilogue: a lane holds, per 32x32 tile, 4 runs (q) of 4 consecutive couts of ONE pixel; its partner lane (+32) holds the
    //      neighbouring 4. v_permlane32_swap of run q (even) against run q+1 leaves 8 consecutive couts per lane -> scale / bias /
    //      residual as 16-byte vectors and ONE 16-byte NHWC store per run pair (half the store instructions of the 8-byte form)
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    // per-channel scale / bias of the block's couts go through LDS once: as global loads inside the unrolled epilogue they were
    // 80 dependent L2 round trips per lane (the 160 live accumulators leave no registers to prefetch them) -- 2/3 of the epilogue's time
    float* const sbl = reinterpret_cast<float*>(smem + HZERO + 5120);  // [2][HC], behind the statistics scratch
    if (EPI != HEPI_RAW) {
        if (t < HC) {
            const int c = c0 + t < a.Cd ? c0 + t : a.Cd - 1;
            sbl[t] = a.scale ? a.scale[c] : 1.f;
            sbl[HC + t] = a.bias ? a.bias[c] : 0.f;
        }
        __syncthreads();
    }
    // The packed 16-byte pieces (8 couts of one pixel) go through a per-wave LDS staging tile [32 pixels][HC couts] and leave as whole
    // NHWC rows: a store instruction then covers 3.2 consecutive pixel rows (64 x 16 B contiguous when the block spans all couts)
    // instead of 32 pixels x 32 B -- the scattered form cost 4.6 - 7.7 us per launch (timeline ablation, profiles/r02_*).
    constexpr int RS = HC * 2 + 16;               // staging row stride (bytes): +16 spreads the 8-lane ds_write_b128 groups over all banks
    constexpr int CH = HC / 8;                    // 16-byte chunks per pixel row
    unsigned char* const stg = smem + HZERO + HEPI_STAGE_OFF + wave * (32 * RS);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int p = pout[g];
        const bool pok = OMAP ? (p0 + wave * (32 * NG) + g * 32 + l31) < a.M
                              : (p < a.M && (!PATCH || py0 + (wave * (32 * NG) + g * 32 + l31) / PATCH_W < a.H));  // (a 24-row patch may hang over the bottom edge)
        const int64_t rb = (int64_t)p * a.res_ld + a.res_coff;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (never __builtin_bit_cast a vector ELEMENT expression: clang reads element 0 of the vector instead)
                    const float lo = acc[f][g][4 * q + r], hi = acc[f][g][4 * q + 4 + r];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    v[r] = __uint_as_float(s0);
                    v[4 + r] = __uint_as_float(s1);
                }
                const int cl = f * 32 + 8 * (q + h);
                const int co = c0 + cl;
                if (EPI != HEPI_RAW) {
                    {
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sbl + cl), s1 = *reinterpret_cast<const f32x4*>(sbl + cl + 4);
                        const f32x4 b0v = *reinterpret_cast<const f32x4*>(sbl + HC + cl), b1v = *reinterpret_cast<const f32x4*>(sbl + HC + cl + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = v[r] * s0[r] + b0v[r];
                            v[4 + r] = v[4 + r] * s1[r] + b1v[r];
                        }
                    }
                    if (a.act == CDET_ACT_SILU) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                    }
                    if (a.res && pok && co < a.Cd) {
                        const u32x4 rv = *reinterpret_cast<const u32x4*>(a.res + rb + co);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[2 * r] += Elem<DT>::to_f32((uint16_t)(rv[r] & 0xffff));
                            v[2 * r + 1] += Elem<DT>::to_f32((uint16_t)(rv[r] >> 16));
                        }
                    }
                }
                if (EPI == HEPI_F32) {
                    // fp32 destination (Detect's biased projections, gradient fan-in, the split-operand accuracy chain of tests/hiprec.py):
                    // 8 consecutive couts of one pixel = 32 contiguous bytes per lane, stored (or accumulated) straight from the registers --
                    // these launches are short or run against HBM, the LDS-staged row form of the 16-bit epilogue would buy nothing
                    if (pok && co < a.Cd) {
                        float* const yo = reinterpret_cast<float*>(a.y) + (int64_t)p * a.dst_ld + a.dst_coff + co;
                        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                        if (a.accum) {
                            o0 += *reinterpret_cast<const f32x4*>(yo);
                            o1 += *reinterpret_cast<const f32x4*>(yo + 4);
                        }
                        *reinterpret_cast<f32x4*>(yo) = o0;
                        *reinterpret_cast<f32x4*>(yo + 4) = o1;
                    }
                    continue;
                }
                u32x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = hpack2<DT>(v[2 * r], v[2 * r + 1]);
                *reinterpret_cast<u32x4*>(stg + l31 * RS + cl * 2) = pk;
            }
        }
        if (EPI == HEPI_F32) continue;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // whole rows out: chunk id -> (pixel of this fragment, 16-byte chunk of its row)
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
            const int id = it * 64 + lane;
            const int px = id / CH, c = id - px * CH;
            if (id < 32 * CH) {
                const u32x4 pk = *reinterpret_cast<const u32x4*>(stg + px * RS + c * 16);
                const int i = wave * (32 * NG) + g * 32 + px;
                int po;
                if (PATCH) po = (pn * a.H + py0 + i / PATCH_W) * W + px0 + i % PATCH_W;
                else po = p0 + i;
                const int co = c0 + 8 * c;
                bool in = po < a.M && (!PATCH || py0 + i / PATCH_W < a.H);
                if (OMAP && in) {
                    const int n_ = po / (a.H * W), r_ = po - n_ * (a.H * W);
                    const int y_ = r_ / W, x_ = r_ - y_ * W;
                    po = (n_ * a.Hd + 2 * y_ + a.cp) * a.Wd + 2 * x_ + a.cq;
                }
                if (in && co < a.Cd) {
                    if (!(ABL & 8)) *reinterpret_cast<u32x4*>(yp + (int64_t)po * a.dst_ld + a.dst_coff + co) = pk;
                    else asm volatile("" ::"v"(pk));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // the next fragment overwrites the staging tile
    }
    if (a.stats != nullptr) {
        const float* stl = reinterpret_cast<const float*>(smem + HZERO);
        __syncthreads();
        if (t < HC && c0 + t < a.Cd) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sv += stl[(m * 2 + 0) * HC + t];
                qv += stl[(m * 2 + 1) * HC + t];
            }
            if (CDET_FOLD(a.fold)) {  // write-through (sc1) stores: the workgroup that draws the last ticket reads them from any XCD
                const __amdgpu_buffer_rsrc_t rs_ = bnf_rsrc(a.stats);
                bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 0) * a.Cd + c0 + t) * 4), sv);
                bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 1) * a.Cd + c0 + t) * 4), qv);
            } else {
                a.stats[((int64_t)pblk * 2 + 0) * a.Cd + c0 + t] = sv;
                a.stats[((int64_t)pblk * 2 + 1) * a.Cd + c0 + t] = qv;
            }
        }
        if (CDET_FOLD(a.fold)) bn_fold_finish<true>(a.fold, a.stats, pblk, c0, HC, cblk, reinterpret_cast<volatile int*>(smem), 256);
    }
#ifdef CDET_PROFILING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CDET_HALO_STAMP(5);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Tiled weight pack: OIHW fp32 master -> [row block of RB][chunk of 32 reduction channels][tap][RB rows][32 k], 16-B slots swizzled;
// RB = 160 (96 when there are at most 96 rows: the 80-channel layers run 3 instead of 5 cout fragments per wave).
//   forward operand : rows = o, reduction = i, tap as stored
//   DGRAD operand   : rows = i, reduction = o, taps flipped (kh, kw) -> (KH-1-kh, KW-1-kw): the stride-1 data gradient becomes a
//                     plain forward convolution of dY with this operand
// One workgroup transposes a [32 o][32 i][taps] tile through LDS (the master is read once, in runs of 32*taps floats).
// ------------------------------------------------------------------------------------------------------------------------------------
// (CDET_RB160_AS_96, experiment builds only: 160-row operands as two 96-row blocks -- the 174-179-VGPR instantiation instead of the 251-VGPR one, so that
//  a BatchNorm wave of the other task's pass fits beside two convolution workgroups on a SIMD; profiles/r05_bn_beside_conv.txt)
#if defined(CDET_EXPERIMENTS) && defined(CDET_RB160_AS_96)
static __host__ __device__ __forceinline__ int row_block(int rows) { return (rows <= 96 || rows == 160) ? 96 : 160; }
#else
static __host__ __device__ __forceinline__ int row_block(int rows) { return rows <= 96 ? 96 : 160; }
#endif

__device__ __forceinline__ int64_t tiled_elem(int row, int chunk, int tap, int k, int nchunk, int taps, int rb) {
    const int cb = row / rb, r = row - cb * rb;
    const int64_t tile = ((int64_t)cb * nchunk + chunk) * taps + tap;
    return tile * (rb * 32) + r * 32 + ((((k >> 3) ^ ((r >> 2) & 3)) << 3) | (k & 7));
}

__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const cdet_pack_tiled_item* __restrict__ items, int n, cdet_pack_tiled_item single,
                                                                  int dtype) {
    __shared__ int s_it;
    __shared__ float tile[32][32 * 9 + 1];
    cdet_pack_tiled_item p;
    if (items != nullptr) {
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
        p = items[s_it];
    } else {
        p = single;
    }
    const int taps = p.kh * p.kw;
    const int tiles_i = (p.I + 31) / 32;
    const int lb = (int)blockIdx.x - p.first_block;
    const int o0 = (lb / tiles_i) * 32, i0 = (lb % tiles_i) * 32;
    const int ni = min(32, p.I - i0), no = min(32, p.O - o0);
    const int run = ni * taps;
    if ((((int64_t)p.I * taps) & 3) == 0 && ((i0 * taps) & 3) == 0 && (run & 3) == 0 && (reinterpret_cast<uintptr_t>(p.w_oihw) & 15) == 0) {
        // 16-byte loads: a row's ni * taps master values are contiguous and start 16-byte aligned (Cin a multiple of 4)
        const int R4 = 8 * taps, r4 = run >> 2;
        for (int e = threadIdx.x; e < 32 * R4; e += 256) {
            const int o = e / R4, q = e - o * R4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o < no && q < r4) v = *reinterpret_cast<const f32x4*>(p.w_oihw + ((int64_t)(o0 + o) * p.I + i0) * taps + 4 * q);
            tile[o][4 * q + 0] = v[0];
            tile[o][4 * q + 1] = v[1];
            tile[o][4 * q + 2] = v[2];
            tile[o][4 * q + 3] = v[3];
        }
    } else {
        for (int e = threadIdx.x; e < 32 * 32 * taps; e += 256) {
            const int o = e / (32 * taps), r = e - o * (32 * taps);
            tile[o][r] = (o < no && r < run) ? p.w_oihw[((int64_t)(o0 + o) * p.I + i0) * taps + r] : 0.f;
        }
    }
    __syncthreads();
    uint16_t* wf = reinterpret_cast<uint16_t*>(p.w_fwd);
    uint16_t* wd = reinterpret_cast<uint16_t*>(p.w_dgrad);
    const int rbf = row_block(p.O), rbd = row_block(p.I);
    // one 16-byte piece (8 k values) per work item: (row 0..31, tap, k-slot 0..3)
    for (int e = threadIdx.x; e < 32 * taps * 4; e += 256) {
        const int kq = e & 3, tap = (e >> 2) % taps, row = (e >> 2) / taps;
        if (wf != nullptr && row < no) {  // rows = o, k = i - i0
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = tile[row][(kq * 8 + j) * taps + tap];
                v[j] = dtype == CDET_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
            }
            const int64_t at = tiled_elem(o0 + row, i0 >> 5, tap, kq * 8, (p.I + 31) >> 5, taps, rbf);
            u32x4 pk = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16), (uint32_t)v[4] | ((uint32_t)v[5] << 16),
                        (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
            *reinterpret_cast<u32x4*>(wf + at) = pk;
        }
        if (wd != nullptr && row < ni) {  // rows = i, k = o - o0, flipped tap
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = tile[kq * 8 + j][row * taps + tap];
                v[j] = dtype == CDET_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
            }
            const int64_t at = tiled_elem(i0 + row, o0 >> 5, taps - 1 - tap, kq * 8, (p.O + 31) >> 5, taps, rbd);
            u32x4 pk = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16), (uint32_t)v[4] | ((uint32_t)v[5] << 16),
                        (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
            *reinterpret_cast<u32x4*>(wd + at) = pk;
        }
    }
}

// launch geometry of a supported convolution
struct HaloPlan {
    bool ok, patch;
    int nf, ng, hp, XH, nsw, nxb;
    bool tri;
    int ks;  // 2: the K loop split inside the workgroup (3x3 half tiles with an even number of full channel chunks)
    size_t lds;
};

static HaloPlan halo_plan(const cdet_conv_desc* d) {
    HaloPlan pl = {};
    if (!(d->kh == d->kw && (d->kh == 1 || d->kh == 3))) return pl;
    if (d->stride != 1 || d->pad != d->kh / 2) return pl;
    if (d->Hs != d->Hd || d->Ws != d->Wd) return pl;
    if (d->Cs % 8 != 0 || d->src_ld % 8 != 0 || d->src_coff % 8 != 0) return pl;
    if (d->Cd % 8 != 0 || d->dst_ld % 8 != 0 || d->dst_coff % 8 != 0) return pl;
    if (!(d->dtype == CDET_BF16 || d->dtype == CDET_F16)) return pl;
    if (d->out_dtype != d->dtype && d->out_dtype != CDET_F32) return pl;   // 16-bit in == out, or an fp32 destination (HEPI_F32)
    if (d->accumulate && d->out_dtype != CDET_F32) return pl;              // y += result needs the fp32 destination
    if (d->out_dtype == CDET_F32 && (int64_t)d->N * d->Hd * d->Wd * d->dst_ld >= (1ll << 31)) return pl;
    const int rb = row_block(d->Cd);
    pl.nf = rb / 32;
    const int64_t M = (int64_t)d->N * d->Hs * d->Ws;
    // half tiles (128 pixels, one 32-pixel fragment per wave) when 256-pixel tiles would not even give every CU one workgroup
    pl.ng = (div_up(M, HP) * div_up(d->Cd, rb) < 256 && M > 128) ? 1 : 2;
    if (sw_is(SW_HALO_NG)) {  // halo_ng = 1 | 2 pins the tile size (tests cover the forms on the same shapes; A/B timing)
        const int force = sw(SW_HALO_NG);
        if (force == 1 || force == 2) pl.ng = force;
#ifdef CDET_EXPERIMENTS
        // the 512-pixel tile (one workgroup per CU): 3x3, 160-cout blocks, 16-bit output, linear halo (maps up to 95 wide) -- 9-14 % slower, kept for the record
        if (force == 4 && d->kh == 3 && rb == 160 && d->out_dtype != CDET_F32 && d->Ws <= 95) pl.ng = 4;
#endif
    }
    pl.hp = 128 * pl.ng;
    if (d->kh == 1) {
        pl.patch = false;
        pl.XH = pl.hp;
    } else {
        const int lin = (pl.hp + 2 * (d->Ws + 1) + 15) / 16 * 16;
        const int pat = (PATCH_HPW * PATCH_HPW + 15) / 16 * 16;  // 336
        pl.patch = pl.ng != 4 && d->Hs % PATCH_W == 0 && d->Ws % PATCH_W == 0 && pat < (HP + 2 * (d->Ws + 1) + 15) / 16 * 16;
        if (pl.patch) {  // 16 x 16 patches are 256-pixel tiles
            pl.ng = 2;
            pl.hp = HP;
        }
        pl.XH = pl.patch ? pat : (pl.ng == 2 ? (HP + 2 * (d->Ws + 1) + 15) / 16 * 16 : lin);
        // CDET_HALO_NG=3 (round-5 experiment, opt-in): the 96-cout tile on tall maps as 24-row x 16-column patches (384 pixels) when at most 6 % of
        // the patch rows hang over the bottom edge (160 = 6 x 24 + 16: 5 %). Measured on 160 x 160 80 -> 80: 0.1381 against 0.1372 ms -- a third less
        // weight LDS-DMA, a fifth fewer fragment reads and barriers per MFMA buy NOTHING there (profiles/r05_halo_ng4.txt, section 3)
#ifdef CDET_EXPERIMENTS
        if (pl.patch && pl.nf == 3 && d->out_dtype != CDET_F32 && sw(SW_HALO_NG) == 3) {
            const int rows3 = div_up(d->Hs, 24) * 24;
            if ((rows3 - d->Hs) * 100 <= 6 * d->Hs) {
                pl.ng = 3;
                pl.hp = 384;
                pl.XH = (26 * PATCH_HPW + 15) / 16 * 16;  // 480
            }
        }
#endif
        if (pl.XH > 16 * 4 * MAXXP * (pl.ng >= 3 ? 2 : 1)) return pl;
        // three workgroups per CU for the 96-cout 16 x 16 patch form (see TRI in the kernel) when the grid holds at least three full rounds of them
        // (160 x 160 80 -> 80 at batch 32: 3 200 workgroups, 0.142 -> 0.131 ms; 80 x 80 80 -> 80 with 800 loses 6 % and keeps two).
        // CDET_HALO_WG3=1 forces the form (tests), 0 disables it
        const int w3v = sw_is(SW_HALO_WG3) ? sw(SW_HALO_WG3) : -1;
        const int64_t tiles = (int64_t)d->N * (d->Hs / PATCH_W) * (d->Ws / PATCH_W) * div_up(d->Cd, rb);
        if (pl.patch && pl.ng == 2 && pl.nf == 3 && d->out_dtype != CDET_F32 && (w3v == 1 || (w3v < 0 && tiles >= 3 * 768))) {
            pl.tri = true;
            pl.XH = PATCH_HPW * PATCH_HPW;  // 324
        }
    }
    if (M >= (1ll << 31) - HP) return pl;
    if (M * d->src_ld * 2 >= 0xC0000000ll) return pl;
    const int64_t wb = (int64_t)div_up(d->Cd, rb) * div_up(d->Cs, 32) * d->kh * d->kw * rb * HROW;
    if (wb >= 0xC0000000ll) return pl;
    pl.nxb = d->kh == 1 ? 3 : 2;
    // ring depth: three stages when two workgroups still fit a CU (80 KiB each); the patch, 1x1 and half-tile forms always do
    const size_t base = (size_t)HZERO + (size_t)pl.nxb * pl.XH * HROW;
    // (the 512-pixel tile has the CU's LDS to itself: a six-stage ring keeps five weight tiles in flight -- the LDS-DMA stream of a workgroup is
    //  latency-bound, (stages - 1) x 10 KB per ~2 us)
    pl.nsw = pl.ng == 4 ? 6 : (base + 3 * (size_t)rb * HROW <= 80 * 1024 ? 3 : 2);
    if (pl.tri) pl.nsw = 2;
    if ((pl.patch || d->kh == 1 || pl.ng == 1) && pl.nsw != 3 && !pl.tri) return pl;
    pl.lds = base + (size_t)pl.nsw * rb * HROW - (pl.tri ? HZERO : 0);
    const size_t epi = (size_t)HZERO + HEPI_STAGE_OFF + 4 * 32 * (size_t)(rb * 2 + 16);  // the epilogue's store staging
    if (pl.lds < epi) pl.lds = epi;
    if (pl.tri && pl.lds != 42 * 1280) return HaloPlan{};
    // the half-tile form's K loop split inside the workgroup (KS = 2 in the kernel): 3x3, 160-cout blocks, >= 4 full chunks, an even number of them, and a
    // grid that gives no CU a second workgroup anyway (the split form owns its CU: 20 x 20 320 -> 320 at batch 32, 200 workgroups, 34.3 -> 31.2 us;
    // 640 -> 320 57.5 -> 51.6; a 400-workgroup grid loses 7 %). CDET_HALO_KS=1 keeps the one-chain form, =2 forces the split (tests)
    pl.ks = 1;
    {
        const int nchunk = div_up(d->Cs, 32);
        const int64_t nblk = (int64_t)div_up(M, pl.hp) * div_up(d->Cd, rb);
        const int ev = sw_is(SW_HALO_KS) ? sw(SW_HALO_KS) : 0;
        if (ev != 1 && (ev == 2 || nblk <= 256) && pl.ng == 1 && d->kh == 3 && pl.nf == 5 && pl.nsw == 3 && !pl.patch && d->Cs % 32 == 0 && nchunk >= 4 &&
            nchunk % 2 == 0 && 2 * pl.lds <= 160 * 1024)
            pl.ks = 2;
    }
    pl.ok = true;
    return pl;
}

template <int DT, int NT, int NF, int EPI, int NSW, bool PATCH, int NG>
static void launch_halo(const HaloArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
#ifdef CDET_PROFILING
    {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = -1;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG>, 256, lds);
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void*)conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG>);
            fprintf(stderr, "[cdet] conv_halo_kernel<%d,%d,%d,%d,%d,%d,%d>: lds %zu B, occupancy API %d blocks/CU (%s), regs %d\n", DT, NT, NF, EPI, NSW, (int)PATCH, NG, lds, nb,
                    hipGetErrorString(e), fa.numRegs);
        }
    }
    static int abl = -1;
    if (abl < 0) {
        const char* e = getenv("CDET_HALO_ABLATE");
        abl = e ? atoi(e) : 0;
        if (abl) fprintf(stderr, "[cdet] CDET_HALO_ABLATE=%d: conv results are WRONG by design (timing experiment)\n", abl);
    }
    if constexpr (DT == CDET_BF16 && NT == 9 && EPI == HEPI_FULL && NSW == 3 && NG == 2 && (NF == 5 || (NF == 3 && PATCH))) {
#define CDET_HABL(N)                                                                                                                                      \
    case N:                                                                                                                                                \
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG, N>), dim3(nblocks), dim3(256), lds, s, a);                                       \
        return;
        switch (abl) {
            CDET_HABL(1) CDET_HABL(2) CDET_HABL(3) CDET_HABL(4) CDET_HABL(7) CDET_HABL(8) CDET_HABL(15) CDET_HABL(16) CDET_HABL(24) CDET_HABL(32)
            default: break;
        }
#undef CDET_HABL
    }
#endif
    hipLaunchKernelGGL((conv_halo_kernel<DT, NT, NF, EPI, NSW, PATCH, NG>), dim3(nblocks), dim3(256), lds, s, a);
}

// the in-workgroup K split of the half-tile form: 512 threads, two LDS regions of the one-chain form's size (at least the 80 KiB of the hand-over)
template <int DT, int NF, int EPI>
static void launch_halo_ks2(HaloArgs a, size_t lds_half, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<DT, 9, NF, EPI, 3, false, 1, 0, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = true;
    }
    a.ks_region = (int)lds_half;
    size_t lds = 2 * lds_half;
    const size_t ex = (size_t)NF * 16 * 256 * 4;
    if (lds < ex) lds = ex;
    hipLaunchKernelGGL((conv_halo_kernel<DT, 9, NF, EPI, 3, false, 1, 0, false, false, 2>), dim3(nblocks), dim3(512), lds, s, a);
}

template <int DT, int NF, int NG>
static void launch_halo_cat(const HaloArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<DT, 1, NF, HEPI_FULL, 3, false, NG, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<DT, 1, NF, HEPI_FULL, 3, false, NG, 0, false, true>), dim3(nblocks), dim3(256), lds, s, a);
}

template <int DT>
static void dispatch_halo_cat(const HaloArgs& a, const HaloPlan& pl, int nblocks, hipStream_t s) {
    if (pl.nf == 5) {
        if (pl.ng == 1) launch_halo_cat<DT, 5, 1>(a, pl.lds, nblocks, s);
        else launch_halo_cat<DT, 5, 2>(a, pl.lds, nblocks, s);
    } else {
        if (pl.ng == 1) launch_halo_cat<DT, 3, 1>(a, pl.lds, nblocks, s);
        else launch_halo_cat<DT, 3, 2>(a, pl.lds, nblocks, s);
    }
}

template <int DT, int NF, int EPI>
static void dispatch_halo2(const HaloArgs& a, int k, const HaloPlan& pl, int nblocks, hipStream_t s) {
#ifdef CDET_EXPERIMENTS
    if (pl.ng == 4) {
        if constexpr (NF == 5 && EPI != HEPI_F32) launch_halo<DT, 9, 5, EPI, 6, false, 4>(a, pl.lds, nblocks, s);
        return;
    }
    if (pl.ng == 3) {
        if constexpr (NF == 3 && EPI != HEPI_F32) launch_halo<DT, 9, 3, EPI, 3, true, 3>(a, pl.lds, nblocks, s);
        return;
    }
#endif
    if (pl.ng == 1) {
        if (k == 1) launch_halo<DT, 1, NF, EPI, 3, false, 1>(a, pl.lds, nblocks, s);
        else if (pl.ks == 2) {
            if constexpr (NF == 5) launch_halo_ks2<DT, 5, EPI>(a, pl.lds, nblocks, s);
        } else launch_halo<DT, 9, NF, EPI, 3, false, 1>(a, pl.lds, nblocks, s);
    } else if (k == 1) launch_halo<DT, 1, NF, EPI, 3, false, 2>(a, pl.lds, nblocks, s);
    else if (pl.tri) {
        if constexpr (NF == 3 && EPI != HEPI_F32) launch_halo<DT, 9, 3, EPI, 2, true, 2>(a, pl.lds, nblocks, s);
    } else if (pl.patch) launch_halo<DT, 9, NF, EPI, 3, true, 2>(a, pl.lds, nblocks, s);
    else if (pl.nsw == 3) launch_halo<DT, 9, NF, EPI, 3, false, 2>(a, pl.lds, nblocks, s);
    else launch_halo<DT, 9, NF, EPI, 2, false, 2>(a, pl.lds, nblocks, s);
}

template <int DT>
static void dispatch_halo(const HaloArgs& a, int k, bool full, bool f32out, const HaloPlan& pl, int nblocks, hipStream_t s) {
    if (f32out) {
        if (pl.nf == 5) dispatch_halo2<DT, 5, HEPI_F32>(a, k, pl, nblocks, s);
        else dispatch_halo2<DT, 3, HEPI_F32>(a, k, pl, nblocks, s);
        return;
    }
    if (pl.nf == 5) {
        if (full) dispatch_halo2<DT, 5, HEPI_FULL>(a, k, pl, nblocks, s);
        else dispatch_halo2<DT, 5, HEPI_RAW>(a, k, pl, nblocks, s);
    } else {
        if (full) dispatch_halo2<DT, 3, HEPI_FULL>(a, k, pl, nblocks, s);
        else dispatch_halo2<DT, 3, HEPI_RAW>(a, k, pl, nblocks, s);
    }
}

}  // namespace cdet

namespace cdet {
// csrc/conv_pair.hip: 1x1 layers with an even number of 160-cout blocks -- two blocks share the staged pixel tile
bool pair_plan_ok(const cdet_conv_desc* d);
int pair_launch(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias, const void* residual, void* y,
                float* stats, hipStream_t s, const CatSrcs* cat = nullptr, const BnFold* fold = nullptr);

// csrc/conv_pp.hip: the two-phase (ping-pong) 8-wave form of the 3x3 160-cout tile -- two pixel tiles share one weight ring
bool pp_plan_ok(const cdet_conv_desc* d, int nf, int ng, bool patch, int XH);
int pp_launch(const cdet_conv_desc* d, bool patch, int XH, int n_pblk, const void* x, const void* w_tiled, const float* scale, const float* bias,
              const void* residual, void* y, float* stats, hipStream_t s);

// host-side check + fill of a virtual-Concat source list against the convolution's descriptor
static int cat_fill(const cdet_conv_desc* d, const cdet_cat_src* srcs, int n, CatSrcs* c) {
    if (!srcs || n < 1 || n > 3) return 0;
    if (!(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hs == d->Hd && d->Ws == d->Wd)) return 0;
    int ctot = 0;
    const uint16_t* xs[3] = {nullptr, nullptr, nullptr};
    unsigned by[3] = {0u, 0u, 0u};
    int ld[3] = {0, 0, 0}, co[3] = {0, 0, 0}, up[3] = {0, 0, 0}, c0[3] = {0, 0x7fffffff, 0x7fffffff};
    for (int i = 0; i < n; ++i) {
        const cdet_cat_src& sc = srcs[i];
        if (!sc.x || sc.C <= 0 || sc.C % 8 != 0 || sc.ld % 8 != 0 || sc.coff % 8 != 0 || sc.coff + sc.C > sc.ld) return 0;
        if (i + 1 < n && sc.C % 32 != 0) return 0;       // segments start on 32-channel chunks
        if (sc.upsample && (d->Hs % 2 != 0 || d->Ws % 2 != 0)) return 0;
        const int64_t px = sc.upsample ? (int64_t)d->N * (d->Hs / 2) * (d->Ws / 2) : (int64_t)d->N * d->Hs * d->Ws;
        const int64_t bytes = px * sc.ld * 2;
        if (bytes >= 0xC0000000ll) return 0;
        xs[i] = (const uint16_t*)sc.x; by[i] = (unsigned)bytes; ld[i] = sc.ld; co[i] = sc.coff; up[i] = sc.upsample ? 1 : 0;
        c0[i] = ctot / 32;
        ctot += sc.C;
    }
    c->x0 = xs[0]; c->x1 = xs[1]; c->x2 = xs[2];
    c->b0 = by[0]; c->b1 = by[1]; c->b2 = by[2];
    c->ld0 = ld[0]; c->ld1 = ld[1]; c->ld2 = ld[2];
    c->co0 = co[0]; c->co1 = co[1]; c->co2 = co[2];
    c->up0 = up[0]; c->up1 = up[1]; c->up2 = up[2];
    c->c1 = c0[1]; c->c2 = c0[2];
    if (ctot != d->Cs) return 0;
    c->n = n; c->H = d->Hs; c->W = d->Ws;
    return 1;
}
}  // namespace cdet

using namespace cdet;

#ifdef CDET_PROFILING
// profiling builds only (not part of the C-ABI): per-workgroup timeline buffer of 8 u64 per workgroup, or NULL to switch it off
extern "C" int cdet_debug_halo_timeline(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_halo_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int cdet_conv2d_tiled_ok(const cdet_conv_desc* d) { return d && halo_plan(d).ok ? 1 : 0; }

// pixel tiles of a launch (= BatchNorm partial rows): M / tile, or per image ceil(H / patch rows) x (W / 16) patches in the patch forms
static int halo_pixel_tiles(const cdet_conv_desc* d, const HaloPlan& pl) {
    if (pl.ok && pl.patch) return d->N * div_up(d->Hs, pl.hp / PATCH_W) * (d->Ws / PATCH_W);
    return div_up((int64_t)d->N * d->Hd * d->Wd, pl.ok ? pl.hp : HP);
}

extern "C" int cdet_conv2d_tiled_stat_blocks(const cdet_conv_desc* d) { return halo_pixel_tiles(d, halo_plan(d)); }

extern "C" int64_t cdet_tiled_weight_elems(int32_t rows, int32_t red, int32_t kh, int32_t kw) {
    const int rb = row_block(rows);
    return (int64_t)div_up(rows, rb) * div_up(red, 32) * kh * kw * (rb * 32);
}

extern "C" int cdet_pack_weights_tiled(const cdet_pack_tiled_item* items, int32_t n_items, int32_t n_blocks_total, int32_t dtype, void* stream) {
    CDET_CHECK_ARG(items && n_items > 0 && n_blocks_total >= n_items, "cdet_pack_weights_tiled: bad arguments");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_pack_weights_tiled: dtype must be bf16/f16");
    cdet_pack_tiled_item none = {};
    hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3(n_blocks_total), dim3(256), 0, (hipStream_t)stream, items, n_items, none, dtype);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_pack_weight_tiled(const float* w_oihw, void* w_fwd, void* w_dgrad, int32_t O, int32_t I, int32_t kh, int32_t kw, int32_t dtype,
                                      void* stream) {
    CDET_CHECK_ARG(w_oihw && (w_fwd || w_dgrad) && O > 0 && I > 0, "cdet_pack_weight_tiled: bad arguments");
    CDET_CHECK_ARG(kh == kw && (kh == 1 || kh == 3), "cdet_pack_weight_tiled: 1x1 / 3x3 only");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_pack_weight_tiled: dtype must be bf16/f16");
    cdet_pack_tiled_item it = {};
    it.w_oihw = w_oihw; it.w_fwd = w_fwd; it.w_dgrad = w_dgrad;
    it.O = O; it.I = I; it.kh = kh; it.kw = kw; it.first_block = 0;
    it.n_blocks = div_up(O, 32) * div_up(I, 32);
    hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3(it.n_blocks), dim3(256), 0, (hipStream_t)stream, (const cdet_pack_tiled_item*)nullptr, 1, it, dtype);
    CDET_LAUNCH_CHECK();
    return 0;
}

static int conv2d_tiled_impl(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                             const void* residual, void* y, float* stats, void* stream, const CatSrcs* cat, const BnFold* fold);

extern "C" int cdet_conv2d_tiled_cat_ok(const cdet_conv_desc* d, const cdet_cat_src* srcs, int32_t n_src) {
    if (!d) return 0;
    CatSrcs c;
    cdet_conv_desc dd = *d;  // (the descriptor's own source fields are not used: any aligned values pass the geometry check)
    dd.src_ld = (d->Cs + 7) / 8 * 8; dd.src_coff = 0;
    return cat_fill(d, srcs, n_src, &c) && halo_plan(&dd).ok && d->out_dtype == d->dtype ? 1 : 0;
}

extern "C" int cdet_conv2d_tiled_cat(const cdet_conv_desc* d, const cdet_cat_src* srcs, int32_t n_src, const void* w_tiled, const float* scale,
                                     const float* bias, const void* residual, void* y, void* stream) {
    CDET_CHECK_ARG(d && srcs && w_tiled && y, "cdet_conv2d_tiled_cat: null pointer");
    CatSrcs c;
    CDET_CHECK_ARG(cat_fill(d, srcs, n_src, &c), "cdet_conv2d_tiled_cat: bad source list (1x1 stride 1; 1-3 segments, channels / ld / coff multiples of 8, all "
                                                 "but the last a multiple of 32 channels, sum = Cs; upsampled segments need even H, W; buffers < 3 GiB)");
    CDET_CHECK_ARG(d->out_dtype == d->dtype, "cdet_conv2d_tiled_cat: 16-bit in == out");
    cdet_conv_desc dd = *d;
    dd.src_ld = (d->Cs + 7) / 8 * 8; dd.src_coff = 0;
    return conv2d_tiled_impl(&dd, srcs[0].x, w_tiled, scale, bias, residual, y, nullptr, stream, &c, nullptr);
}

extern "C" int cdet_conv2d_tiled(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                                 const void* residual, void* y, float* stats, void* stream) {
    return conv2d_tiled_impl(d, x, w_tiled, scale, bias, residual, y, stats, stream, nullptr, nullptr);
}

// The train form with the BatchNorm statistics finished inside the launch (bn_fold.h). rows / column blocks of the launch: see cdet_bn_fold_ok.
extern "C" int cdet_conv2d_tiled_bn(const cdet_conv_desc* d, const void* x, const void* w_tiled, void* y, float* stats, const cdet_bn_fold* fold_dev,
                                    void* stream) {
    CDET_CHECK_ARG(stats, "cdet_conv2d_tiled_bn: the partial-sum rows are needed with or without the fold");
#ifndef CDET_EXPERIMENTS
    CDET_CHECK_ARG(!fold_dev, "cdet_conv2d_tiled_bn: the in-launch BatchNorm fold is compiled into -DCDET_EXPERIMENTS builds only (cdet_has_experiments())");
#endif
    CDET_CHECK_ARG(!fold_dev || cdet_conv2d_tiled_bn_ok(d), "cdet_conv2d_tiled_bn: this launch has more partial rows / column blocks than the fold takes");
    return conv2d_tiled_impl(d, x, w_tiled, nullptr, nullptr, nullptr, y, stats, stream, nullptr, fold_dev);
}

extern "C" int cdet_conv2d_tiled_bn_ok(const cdet_conv_desc* d) {
#ifndef CDET_EXPERIMENTS
    return 0;  // (the in-launch BatchNorm fold is an experiment build's: bn_fold.h)
#endif
    if (!d || !cdet_conv2d_tiled_ok(d) || d->out_dtype == CDET_F32) return 0;
    const HaloPlan pl = halo_plan(d);
    if (!pl.ok) return 0;
    const int rows = cdet_conv2d_tiled_stat_blocks(d);
    return rows <= BNF_CL * BNF_MAX_CL && div_up(d->Cd, pl.nf * 32) <= BNF_MAX_CB;
}

static int conv2d_tiled_impl(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                             const void* residual, void* y, float* stats, void* stream, const CatSrcs* cat, const BnFold* fold) {
    CDET_CHECK_ARG(d && x && w_tiled && y, "cdet_conv2d_tiled: null pointer");
    const HaloPlan pl = halo_plan(d);
    CDET_CHECK_ARG(pl.ok, "cdet_conv2d_tiled: unsupported geometry (need stride 1, k in {1,3}, Cs/Cd/ld/coff %% 8 == 0, 16-bit in, out = the same "
                          "type or fp32 (accumulate: fp32 only), 3x3: W <= 95 or H, W multiples of 16)");
    const bool f32out = d->out_dtype == CDET_F32;
    CDET_CHECK_ARG(!(f32out && stats), "cdet_conv2d_tiled: BatchNorm partial sums go with the 16-bit raw output");
    CDET_CHECK_ARG(d->mode == CDET_CONV_FWD, "cdet_conv2d_tiled: the data gradient is a FWD call on the DGRAD operand of cdet_pack_weights_tiled");
    CDET_CHECK_ARG(!residual || (d->res_ld % 8 == 0 && d->res_coff % 8 == 0), "cdet_conv2d_tiled: residual ld/coff must be multiples of 8");
    if (!f32out && pl.nf == 5 && pl.ng == 2 && pair_plan_ok(d)) {
        const int rc = pair_launch(d, x, w_tiled, scale, bias, residual, y, stats, (hipStream_t)stream, cat, fold);
        CDET_LAUNCH_CHECK();
        return rc;
    }
    {
        const int pp_mode = sw(SW_CONV_PP);
        // mode 1: launches whose pair grid is a single round (one 8-wave workgroup per CU at most: the 40 x 40 layers at batch 32); a multi-round launch
        // of the 4-wave form hides prologues / epilogues behind its other workgroup's K loop, the one-workgroup-per-CU form cannot. 2: wherever it fits
        const int n_pair_wg = (halo_pixel_tiles(d, pl) + 1) / 2 * div_up(d->Cd, 160);
        if (pp_mode && (pp_mode == 2 || n_pair_wg <= 256) && !cat && !fold && !pl.tri && pl.ks == 1 && pp_plan_ok(d, pl.nf, pl.ng, pl.patch, pl.XH)) {
            const int rc = pp_launch(d, pl.patch, pl.XH, halo_pixel_tiles(d, pl), x, w_tiled, scale, bias, residual, y, stats, (hipStream_t)stream);
            CDET_LAUNCH_CHECK();
            return rc;
        }
    }
    const int rb = pl.nf * 32;
    HaloArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_tiled; a.scale = scale; a.bias = bias; a.res = (const uint16_t*)residual;
    a.y = y; a.stats = stats; a.fold = stats ? fold : nullptr;
    a.H = d->Hs; a.W = d->Ws; a.Cd = d->Cd;
    a.M = d->N * d->Hs * d->Ws;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dst_ld = d->dst_ld; a.dst_coff = d->dst_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.nchunk = div_up(d->Cs, 32);
    a.halfk = (d->kh == 3 && d->Cs % 32 != 0 && d->Cs % 32 <= 16 && tune_env("CDET_HALO_HALFK", 1)) ? 1 : 0;
    a.Cs = d->Cs;
    a.n_pblk = halo_pixel_tiles(d, pl);
    a.n_cblk = div_up(d->Cd, rb);
    a.act = d->act;
    a.XH = pl.XH;
    a.tiles_x = d->Ws / PATCH_W;
    a.tiles_per_img = div_up(d->Hs, pl.hp / PATCH_W) * (d->Ws / PATCH_W);
    a.x_bytes = (unsigned)((int64_t)a.M * d->src_ld * 2);
    a.w_bytes = (unsigned)((int64_t)a.n_cblk * a.nchunk * d->kh * d->kw * rb * HROW);
    a.wts = 1; a.wt0 = 0; a.Hd = d->Hd; a.Wd = d->Wd; a.cp = a.cq = 0;
    a.accum = d->accumulate ? 1 : 0;
    const bool full = scale || bias || residual || d->act != CDET_ACT_NONE;
    const int nblocks = a.n_pblk * a.n_cblk;
    hipStream_t s = (hipStream_t)stream;
    if (cat) {
        a.cat = *cat;
        if (d->dtype == CDET_BF16) dispatch_halo_cat<CDET_BF16>(a, pl, nblocks, s);
        else dispatch_halo_cat<CDET_F16>(a, pl, nblocks, s);
        CDET_LAUNCH_CHECK();
        return 0;
    }
    a.cat.n = 0;
    if (d->dtype == CDET_BF16) dispatch_halo<CDET_BF16>(a, d->kh, full, f32out, pl, nblocks, s);
    else dispatch_halo<CDET_F16>(a, d->kh, full, f32out, pl, nblocks, s);
    CDET_LAUNCH_CHECK();
    return 0;
}

