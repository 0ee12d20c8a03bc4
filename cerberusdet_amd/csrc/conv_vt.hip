// Tap-resident implicit-GEMM kernels for the STRIDE-2 3x3 convolutions (models/common.py:57-62 with s = 2: the four down-sampling rows
// of the backbone and the two of every neck) -- forward and data gradient -- on the machinery of conv_halo.hip (LDS-DMA staged pixel
// tile, pre-packed 10 KiB weight tiles through a 3-stage ring, v_mfma_f32_32x32x16, counted vmcnt, two-phase K step).
//
// A stride-2 convolution re-uses a staged input pixel for 9/4 taps instead of 9, and the pixels a tile of outputs needs are spread over
// 4x its area. Both problems disappear when the input is seen as its four PARITY PLANES X_pq(y', x') = X(2y' + p, 2x' + q), each of
// the output's size:
//     out(oy, ox) = sum_{ky, kx} X(2 oy + ky - 1, 2 ox + kx - 1) w[ky, kx]
//                 = P11: 4 taps (ky, kx in {0, 2}) at (oy - [ky = 0], ox - [kx = 0])   P10: 2 taps (kx = 1)
//                   P01: 2 taps (ky = 1)                                              P00: 1 tap (1, 1)
// FORWARD (MODE 0): per 32-channel chunk the K loop visits the four planes as "virtual chunks" of 4 / 2 / 2 / 1 tap steps. A plane's
// halo (256 consecutive output pixels + one row + one pixel in front, or a 17 x 17 patch) is ONE strided gather of the NHWC input by
// the LDS-DMA (address = plane pixel * 2 + (p, q), any stride is free for per-lane addresses) into one of two pixel buffers, issued
// as soon as the barrier that ends the last read of that buffer has passed; every tap of the plane then reads it with a different row
// offset. 9 steps per chunk as for a stride-1 3x3, 4 x 5 instead of ~5 pixel pieces per wave and chunk.
// DATA GRADIENT (MODES 1-3 + the 1x1 kernel): dX's four parity classes are four stride-1 convolutions of dY (the same decomposition
// read backwards): class (1,1) has 4 taps, (1,0) / (0,1) two, (0,0) one. Each class is one launch that stages dY tiles exactly like a
// stride-1 convolution, takes its weight tiles out of the ordinary 9-tap DGRAD operand of cdet_pack_weights_tiled by tap id, and writes
// (or accumulates onto) every other pixel of every other row of dX. The round-1 kernel ran the same classes as short-K implicit
// GEMMs that re-fetched every dY pixel per tap (329-543 TF/s).
#include <stdlib.h>

#include "halo_common.h"
#include "bn_fold.h"

namespace cdet {

struct VtArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* bias;
    const uint16_t* res;
    void* y;
    float* stats;
    const BnFold* fold;  // see conv_halo.hip
    int Hs, Ws;  // source image size (forward: the input; class: dY)
    int Hp, Wp;  // the tile space: forward = output pixels, class = dY pixels = pixels of one parity plane of dX
    int Hd, Wd;  // destination image size (forward: = Hp, Wp; class: dX = 2 Hp x 2 Wp)
    int Cd, M;   // M = N * Hp * Wp
    int src_ld, src_coff, dst_ld, dst_coff, res_ld, res_coff;
    int nchunk, Cs, n_pblk, n_cblk, act;
    int tiles_x, tiles_per_img;
    int cp, cq;  // class parity: the class writes dX(2y' + cp, 2x' + cq)
    unsigned x_bytes, w_bytes;
    int accum;  // HEPI_F32 only: y (fp32) += result
};

constexpr int VT_S2FWD = 0, VT_CLASS11 = 1, VT_CLASS10 = 2, VT_CLASS01 = 3, VT_CLASS00 = 4;
constexpr int VT_XR = 320;            // rows per pixel buffer: 5 pieces of 16 rows per wave (17 x 17 = 289, 256 + W + 1 <= 297 for W <= 40)
constexpr int VT_NXP = 5;             // pixel DMA pieces per wave and virtual chunk (pieces beyond the halo fetch zeros)
constexpr int VT_HPW = PATCH_W + 1;   // patch mode: halo pitch 17
constexpr int VEPI_STAGE_OFF = 6912;

// steps per (real) chunk
template <int MODE> struct VtMode;
template <> struct VtMode<VT_S2FWD> { static constexpr int NT = 9; };
template <> struct VtMode<VT_CLASS11> { static constexpr int NT = 4; };
template <> struct VtMode<VT_CLASS10> { static constexpr int NT = 2; };
template <> struct VtMode<VT_CLASS01> { static constexpr int NT = 2; };
template <> struct VtMode<VT_CLASS00> { static constexpr int NT = 1; };

// (ky, kx) of step u
template <int MODE>
__device__ __forceinline__ constexpr int vt_ky(int u) {
    if (MODE == VT_S2FWD) return u < 4 ? (u >> 1) * 2 : (u < 6 ? (u - 4) * 2 : 1);
    if (MODE == VT_CLASS11) return (u >> 1) * 2;
    if (MODE == VT_CLASS10) return u * 2;
    return 1;  // CLASS01, CLASS00
}
template <int MODE>
__device__ __forceinline__ constexpr int vt_kx(int u) {
    if (MODE == VT_S2FWD) return u < 4 ? (u & 1) * 2 : (u < 6 ? 1 : (u < 8 ? (u - 6) * 2 : 1));
    if (MODE == VT_CLASS11) return (u & 1) * 2;
    if (MODE == VT_CLASS10 || MODE == VT_CLASS00) return 1;
    return u * 2;
}
// forward: plane index of step u (0 = P11, 1 = P10, 2 = P01, 3 = P00) and whether u is the plane's first step
__device__ __forceinline__ constexpr int vt_plane(int u) { return u < 4 ? 0 : (u < 6 ? 1 : (u < 8 ? 2 : 3)); }

// One workgroup's tile: pixel block pblk, cout block cblk; cp / cq = the parity class a CLASS mode writes.
template <int DT, int NF, int EPI, int MODE, bool PATCH>
__device__ __forceinline__ void vt_body(const VtArgs& a, const int pblk, const int cblk, const int cp, const int cq) {
    constexpr int NG = 2;
    constexpr int NT = VtMode<MODE>::NT;
    constexpr bool FWD = MODE == VT_S2FWD;
    constexpr int HC = NF * 32;
    constexpr int WTILE = HC * HROW;
    constexpr int WPC = WTILE / 1024;     // 1-KiB DMA pieces per weight tile: 10 / 6
    constexpr int NWP = (WPC + 3) / 4;    // pieces per wave: 3 / 2 -- piece j of wave w is piece 4*j + w of the tile; ids beyond the tile
                                          // (two per tile) are issued through an empty descriptor into a dump area, so that every wave
                                          // issues the same number of DMA instructions WITHOUT a divergent branch (a `lane < 32` half piece,
                                          // as conv_halo.hip splits its tiles, made hipcc merge the branch tails here and hoist the M0 of a
                                          // waterfall loop: half of a piece landed 1 KiB off)
    constexpr int NM = NG * NF;
    constexpr int NR = NF + NG;
    constexpr int NSW = 3;
    constexpr int XRB = VT_XR * HROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    const int c0 = cblk * HC;
    const int Wp = a.Wp;
    const int pitch = PATCH ? VT_HPW : Wp;  // halo row pitch of one tile-space row
    const int p0 = pblk * HP;
    int pn = 0, py0 = 0, px0 = 0;
    if (PATCH) {
        pn = pblk / a.tiles_per_img;
        const int r = pblk - pn * a.tiles_per_img;
        py0 = (r / a.tiles_x) * PATCH_W;
        px0 = (r % a.tiles_x) * PATCH_W;
    }
    unsigned char* const xbase = smem + HZERO;
    unsigned char* const wbase = smem + HZERO + 2 * XRB;
    if (t < 16) reinterpret_cast<uint32_t*>(smem)[t] = 0u;  // zero row

    // ---- pixel DMA: piece id 4*i + wave covers halo rows 16*id .. 16*id + 15, 4 lanes (64 B) per row. The halo is the SAME range of
    //      tile-space pixels for every plane (forward: from one row + one pixel in front of the tile; class: from the tile's first pixel
    //      to one row + one pixel behind it), so a plane only adds a constant to the byte offset.
    unsigned xvoff[VT_NXP];
#pragma unroll
    for (int i = 0; i < VT_NXP; ++i) {
        const int hrow = 16 * (4 * i + wave) + (lane >> 2);
        int g;
        bool ok;
        if (PATCH) {
            const int hy = hrow / VT_HPW, hx = hrow - hy * VT_HPW;
            const int y = py0 + hy - (FWD ? 1 : 0), x = px0 + hx - (FWD ? 1 : 0);
            ok = hy < VT_HPW && (unsigned)y < (unsigned)a.Hp && (unsigned)x < (unsigned)Wp;
            g = FWD ? (pn * a.Hs + 2 * y) * a.Ws + 2 * x : (pn * a.Hp + y) * Wp + x;
        } else {
            const int l = p0 + hrow - (FWD ? Wp + 1 : 0);
            ok = l >= 0 && l < a.M;
            if (FWD) {
                const int n = l / (a.Hp * Wp), r = l - n * (a.Hp * Wp);
                const int y = r / Wp, x = r - y * Wp;
                g = (n * a.Hs + 2 * y) * a.Ws + 2 * x;
            } else {
                g = l;
            }
        }
        const unsigned off = ((unsigned)g * (unsigned)a.src_ld + (unsigned)a.src_coff) * 2u + ((unsigned)((lane & 3) ^ ((hrow >> 2) & 3)) << 4);
        xvoff[i] = ok ? off : HSENT;
    }
    const bool partial = (a.Cs & 31) != 0;
    const int xls = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned char* const wdump = wbase + 3 * WTILE;  // 1 KiB nobody reads
    const unsigned wtile0 = (unsigned)cblk * (unsigned)(a.nchunk * 9) * (unsigned)WTILE;  // both weight operands are 9-tap packs

    // weight tile of (chunk, step position u) -> ring stage; chunks beyond the end go through an EMPTY descriptor (zeros, same count)
    auto dma_w1 = [&](int chunk, int wt, int stage, int j) {
        const int id = 4 * j + wave;  // wave-uniform
        const bool real = id < WPC;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (real && chunk < a.nchunk) ? (int)a.w_bytes : 0, 0x00020000);
        unsigned char* dst = real ? wbase + stage * WTILE + id * 1024 : wdump;
        const unsigned soff = wtile0 + (unsigned)(chunk * 9 + wt) * (unsigned)WTILE;
        dma16(rs, (unsigned)((real ? id : 0) * 1024 + lane * 16), soff, dst);
    };
    // pixel piece i of this wave: channels of `chunk`, plane offset `poff` (bytes) -> pixel buffer xb
    auto dma_x = [&](int i, int chunk, unsigned poff, int xb) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, chunk < a.nchunk ? (int)a.x_bytes : 0, 0x00020000);
        unsigned char* dst = xbase + xb * XRB + (4 * i + wave) * 1024;
        // plane and chunk offsets are wave-uniform: they travel in the scalar offset (not part of the range check: a lane whose
        // vector offset is the sentinel still fetches zeros)
        unsigned v = xvoff[i];
        if (partial && chunk * 32 + 8 * xls >= a.Cs) v = HSENT;
        dma16<CDET_HALO_X_AUX>(rs, v, poff + (unsigned)chunk * 64u, dst);
    };
    // byte offset of plane (p, q) of the forward's input
    const unsigned prow = (unsigned)a.Ws * (unsigned)a.src_ld * 2u, pcol = (unsigned)a.src_ld * 2u;
    auto plane_off = [&](int pl) -> unsigned { return FWD ? (pl == 0 ? prow + pcol : (pl == 1 ? prow : (pl == 2 ? pcol : 0u))) : 0u; };

    // ---- prologue DMA: the first virtual chunk's pixels, weight tiles of steps 0 .. 2 -- issued in front of the index divisions and the
    //      accumulator initialisation below (round 6: that arithmetic runs under the fetch latency); waited for right before the first barrier
#pragma unroll
    for (int i = 0; i < VT_NXP; ++i) dma_x(i, 0, plane_off(0), 0);
    if (NT == 1) {
#pragma unroll
        for (int i = 0; i < VT_NXP; ++i) dma_x(i, 1, 0u, 1);
    }
#pragma unroll
    for (int s_ = 0; s_ < NSW; ++s_) {
        const int wt = FWD ? vt_ky<MODE>(s_ % NT) * 3 + vt_kx<MODE>(s_ % NT) : 8 - (vt_ky<MODE>(s_ % NT) * 3 + vt_kx<MODE>(s_ % NT));
#pragma unroll
        for (int j = 0; j < NWP; ++j) dma_w1(s_ / NT, wt, s_, j);
    }

    // ---- fragment offsets ------------------------------------------------------------------------------------------------------------
    const int aoff0 = l31 * HROW + ((h ^ ((l31 >> 2) & 3)) << 4);
    int pixh[NG];
    unsigned vmask[NG];  // bit 0: pixel inside the tensor; bit 1 / bit 2: the row / column neighbour a shifted tap reads exists
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = wave * (32 * NG) + g * 32 + l31;
        unsigned m = 0u;
        if (PATCH) {
            const int iy = i / PATCH_W, ix = i % PATCH_W;
            pixh[g] = FWD ? (iy + 1) * VT_HPW + ix + 1 : iy * VT_HPW + ix;
            m = 7u;  // neighbours outside the image were fetched as zeros
        } else {
            pixh[g] = FWD ? i + Wp + 1 : i;
            const int p = p0 + i;
            const int r = p % (a.Hp * Wp);
            const int y = r / Wp, x = r - y * Wp;
            if (p < a.M) m = 1u | ((FWD ? y > 0 : y < a.Hp - 1) ? 2u : 0u) | ((FWD ? x > 0 : x < Wp - 1) ? 4u : 0u);
        }
        vmask[g] = m;
    }
    // destination pixel of tile pixel i (computed again in the epilogue instead of living through the K loop)
    auto dst_pixel = [&](int i) -> int {
        int n, y, x;
        if (PATCH) {
            n = pn; y = py0 + i / PATCH_W; x = px0 + i % PATCH_W;
        } else {
            const int p = p0 + i;
            if (p >= a.M) return -1;
            n = p / (a.Hp * Wp);
            const int r = p - n * (a.Hp * Wp);
            y = r / Wp; x = r - y * Wp;
        }
        return FWD ? (n * a.Hd + y) * a.Wd + x : (n * a.Hd + 2 * y + cp) * a.Wd + 2 * x + cq;
    };

    f32x16 acc[NF][NG];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;

    wait_vm((NSW - 1) * NWP);  // (the prologue's DMA was issued above, in front of the fragment offsets and the accumulator initialisation)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // B-fragment byte offsets of step position u reading pixel buffer xb
    auto b_offsets = [&](int xb, int u, int (&bo)[NG]) {
        const int dyp = vt_ky<MODE>(u) == 0 ? 1 : 0, dxp = vt_kx<MODE>(u) == 0 ? 1 : 0;
        int tp = pitch, p_[NG];
        asm volatile("" : "+s"(tp));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            p_[g] = pixh[g];
            asm volatile("" : "+v"(p_[g]));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int hrow = FWD ? p_[g] - (dyp * tp + dxp) : p_[g] + (dyp * tp + dxp);
            const int off = hrow * HROW + ((h ^ ((hrow >> 2) & 3)) << 4);
            const unsigned need = 1u | (dyp ? 2u : 0u) | (dxp ? 4u : 0u);
            const bool ok = (vmask[g] & need) == need;
            bo[g] = ok ? off + HZERO + xb * XRB : 0;
        }
    };
    auto frag = [&](const unsigned char* ws_, const int (&bo)[NG], int s_, int i, u32x4 (&af)[NF], u32x4 (&bf)[NG]) {
        if (i < NG) bf[i] = *reinterpret_cast<const u32x4*>(smem + (bo[i] ^ (s_ << 5)));
        else af[i - NG] = *reinterpret_cast<const u32x4*>(ws_ + ((aoff0 ^ (s_ << 5)) + (i - NG) * 32 * HROW));
    };
    // pixel buffer of step position u in chunk `chunk`
    auto xbuf_of = [&](int chunk, int u) -> int { return FWD ? (vt_plane(u) & 1) : (chunk & 1); };

    int bo_cur[NG], bo_nxt[NG];
    u32x4 a0[NF], b0[NG], a1[NF], b1[NG];
    b_offsets(0, 0, bo_cur);
#pragma unroll
    for (int i = 0; i < NR; ++i) frag(wbase, bo_cur, 0, i, a0, b0);

    // One K step; u = position inside the chunk (compile-time after unrolling).
    // Pixel-piece schedule (two buffers; a buffer may be refilled once the barrier behind its last fragment read has passed):
    //   forward  u = 0: P10 of this chunk (phase A)   u = 4: P01 (A)   u = 6: P00 (A)   u = 7: P11 of the NEXT chunk (phase B: it is
    //            needed one step later, behind the single step of P00)
    //   class    u = 0: the next chunk (phase A)
    // Counted wait of the step = DMA instructions issued after the newest one that must have landed (the weight tile of step st + 1,
    // and the pixels of the plane / chunk that step st + 1 starts): derived per u in the comments below.
    // ring stage of the tile of step (chunk, u): steps are numbered st = chunk * NT + u, stage = st % 3
    auto run_step = [&](int chunk, int u, int stage) {
        const int st3 = stage;                 // stage of this step's tile
        const int stn = (stage + 1) % 3;       // stage of the next step's tile
        const unsigned char* ws = wbase + st3 * WTILE;
        const unsigned char* wsn = wbase + stn * WTILE;
        const int un = (u + 1) % NT;
        const int chunkn = chunk + (u + 1 == NT ? 1 : 0);
        // pixel pieces issued in phase A / phase B of this step
        const bool xa = FWD ? (u == 0 || u == 4 || u == 6) : (NT > 1 && u == 0);
        const bool xb_ = (FWD && u == 7) || NT == 1;  // NT == 1: chunk + 2 goes into this chunk's buffer as soon as its last read is done
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma32<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);
            if (2 * i < NR) frag(ws, bo_cur, 1, 2 * i, a1, b1);
            if (2 * i + 1 < NR) frag(ws, bo_cur, 1, 2 * i + 1, a1, b1);
            if (i == NM - 1) b_offsets(xbuf_of(chunkn, un), un, bo_nxt);
            if (xa && i < VT_NXP) {
                if (FWD) {
                    const int pl = u == 0 ? 1 : (u == 4 ? 2 : 3);
                    dma_x(i, chunk, plane_off(pl), pl & 1);
                } else {
                    dma_x(i, chunk + 1, 0u, (chunk + 1) & 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- counted wait. In flight behind the newest instruction that must have landed:
        //   the weight tile of st + 2 (issued in phase B of st - 1)                                     NWP
        //   + the pixel pieces of phase A of this step and of the previous one, unless the plane that starts at st + 1 is among them
        //     (then everything up to its last piece has to land, which leaves only what was issued after it)
        if (FWD) {
            // u: newest needed            issued after it
            // 0: W(st+1) [B of u7 prev]   X(P11') was in front of W there; A(u8) none; B(u8) W; A(u0) 5 pieces        -> NWP + 5
            // 1: W(st+1) [B of u8]        A(u0) 5; B(u0) W                                                             -> NWP + 5
            // 2: W(st+1) [B of u0]        B(u1) W                                                                      -> NWP
            // 3: W(st+1), P10 [A of u0]   B(u2) W                                                                      -> NWP
            // 4: W(st+1) [B of u2]        B(u3) W; A(u4) 5                                                             -> NWP + 5
            // 5: P01 [A of u4]            B(u4) W                                                                      -> NWP
            // 6: W(st+1) [B of u4]        B(u5) W; A(u6) 5                                                             -> NWP + 5
            // 7: P00 [A of u6]            B(u6) W                                                                      -> NWP
            // 8: P11' [B of u7, in front of that phase's W]   B(u7) W                                                  -> NWP
            if (u == 0 || u == 1 || u == 4 || u == 6) wait_vm_lgkm0<NWP + VT_NXP>();
            else wait_vm_lgkm0<NWP>();
        } else if (NT == 4) {
            // 0: W(st+1) [B of u2 prev]   B(u3) W; A(u0) 5 -> NWP + 5     1: W(st+1) [B of u3 prev]  A(u0) 5; B(u0) W -> NWP + 5
            // 2: W(st+1) [B of u0]        B(u1) W -> NWP                  3: W(st+1), next chunk [A of u0]  B(u2) W -> NWP
            if (u < 2) wait_vm_lgkm0<NWP + VT_NXP>();
            else wait_vm_lgkm0<NWP>();
        } else if (NT == 2) {
            // 0: W(st+1) [B of u0 prev]   B(u1 prev) W; A(u0) 5 -> NWP + 5     1: next chunk [A of u0]  B(u0) W -> NWP
            if (u == 0) wait_vm_lgkm0<NWP + VT_NXP>();
            else wait_vm_lgkm0<NWP>();
        } else {
            // NT == 1: the pixels of chunk st + 1 were issued in phase B of st - 1, in front of that phase's W -> NWP
            wait_vm_lgkm0<NWP>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: MFMAs of k16 #1; the weight tile three steps ahead into the stage just freed; fragment reads of (st + 1, k16 #0)
        {
            const int u3 = (u + 3) % NT, c3 = chunk + (u + 3) / NT;
            const int wt3 = FWD ? vt_ky<MODE>(u3) * 3 + vt_kx<MODE>(u3) : 8 - (vt_ky<MODE>(u3) * 3 + vt_kx<MODE>(u3));
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mfma32<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
                if (xb_) {  // the next chunk's P11 (NT == 1: chunk + 2) first: needed one step from now; two pieces per slot when there are only six slots
                    const int xc = NT == 1 ? chunk + 2 : chunk + 1, xbf = NT == 1 ? (chunk & 1) : 0;
                    if (NM >= 10) {
                        if (i < VT_NXP) dma_x(i, xc, plane_off(0), xbf);
                    } else {
                        if (2 * i < VT_NXP) dma_x(2 * i, xc, plane_off(0), xbf);
                        if (2 * i + 1 < VT_NXP) dma_x(2 * i + 1, xc, plane_off(0), xbf);
                    }
                }
                const int w0s = xb_ ? (NM >= 10 ? 5 : 3) : 0;
                const int w1s = xb_ ? (NM >= 10 ? 7 : 4) : (NM >= 10 ? 3 : 2);
                const int w2s = xb_ ? (NM >= 10 ? 9 : 5) : (NM >= 10 ? 6 : 4);
                if (i == w0s) dma_w1(c3, wt3, st3, 0);
                if (i == w1s) dma_w1(c3, wt3, st3, 1);
                if (NWP > 2 && i == w2s) dma_w1(c3, wt3, st3, 2);
                if (i >= 1 && 2 * (i - 1) < NR) frag(wsn, bo_nxt, 0, 2 * (i - 1), a0, b0);
                if (i >= 1 && 2 * (i - 1) + 1 < NR) frag(wsn, bo_nxt, 0, 2 * (i - 1) + 1, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) bo_cur[g] = bo_nxt[g];
    };

    // The ring stage of a step is st % 3; the loops are unrolled over the least common period of (NT, 3) so that it is a constant.
    if (NT == 9) {
        for (int chunk = 0; chunk < a.nchunk; ++chunk) {
#pragma unroll
            for (int u = 0; u < 9; ++u) run_step(chunk, u, u % 3);
        }
    } else if (NT == 4) {
        for (int c3 = 0; c3 < a.nchunk; c3 += 3) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (c3 + k / 4 < a.nchunk) run_step(c3 + k / 4, k % 4, k % 3);
        }
    } else if (NT == 2) {
        for (int c3 = 0; c3 < a.nchunk; c3 += 3) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (c3 + k / 2 < a.nchunk) run_step(c3 + k / 2, k % 2, k % 3);
        }
    } else {
        for (int c3 = 0; c3 < a.nchunk; c3 += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (c3 + k < a.nchunk) run_step(c3 + k, 0, k);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- BN statistics of the raw convolution (forward, train mode) -------------------------------------------------------------------------
    if (FWD && a.stats != nullptr) {
        float* stl = reinterpret_cast<float*>(smem + HZERO);
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
        __syncthreads();
        if (t < HC && c0 + t < a.Cd) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sv += stl[(m * 2 + 0) * HC + t];
                qv += stl[(m * 2 + 1) * HC + t];
            }
            if (CDET_FOLD(a.fold)) {
                const __amdgpu_buffer_rsrc_t rs_ = bnf_rsrc(a.stats);
                bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 0) * a.Cd + c0 + t) * 4), sv);
                bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 1) * a.Cd + c0 + t) * 4), qv);
            } else {
                a.stats[((int64_t)pblk * 2 + 0) * a.Cd + c0 + t] = sv;
                a.stats[((int64_t)pblk * 2 + 1) * a.Cd + c0 + t] = qv;
            }
        }
    }

    // ---- epilogue (as conv_halo.hip): 8 consecutive couts per lane, scale / bias / SiLU / residual, LDS-staged whole-row stores ----------
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    float* const sbl = reinterpret_cast<float*>(smem + HZERO + 5120);
    if (EPI != HEPI_RAW) {
        if (t < HC) {
            const int c = c0 + t < a.Cd ? c0 + t : a.Cd - 1;
            sbl[t] = a.scale ? a.scale[c] : 1.f;
            sbl[HC + t] = a.bias ? a.bias[c] : 0.f;
        }
        __syncthreads();
    }
    constexpr int RS = HC * 2 + 16;
    constexpr int CH = HC / 8;
    unsigned char* const stg = smem + HZERO + VEPI_STAGE_OFF + wave * (32 * RS);
    int* const pol = reinterpret_cast<int*>(smem + HZERO + VEPI_STAGE_OFF + 4 * (32 * RS)) + wave * 32;  // destination pixel of the fragment's 32 rows
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int p = dst_pixel(wave * (32 * NG) + g * 32 + l31);
        const bool pok = p >= 0;
        const int64_t rb = (int64_t)p * a.res_ld + a.res_coff;
        if (h == 0) pol[l31] = p;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
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
                if (EPI == HEPI_F32) {  // fp32 destination, optionally accumulating (see conv_halo.hip): straight from the registers
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
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
            const int id = it * 64 + lane;
            const int px = id / CH, c = id - px * CH;
            if (id < 32 * CH) {
                const u32x4 pk = *reinterpret_cast<const u32x4*>(stg + px * RS + c * 16);
                const int po = pol[px];
                const int co = c0 + 8 * c;
                if (po >= 0 && co < a.Cd) *reinterpret_cast<u32x4*>(yp + (int64_t)po * a.dst_ld + a.dst_coff + co) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // train form with the statistics finished in this launch (bn_fold.h); tickets at the very end, when the staging LDS is free
    if (FWD && EPI == HEPI_RAW && a.stats != nullptr && CDET_FOLD(a.fold))
        bn_fold_finish<true>(a.fold, a.stats, pblk, c0, HC, cblk, reinterpret_cast<volatile int*>(smem));
}

// XCD-aware remap (bijective): consecutive logical ids run on ONE XCD
__device__ __forceinline__ int vt_logical_id() {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int DT, int NF, int EPI, int MODE, bool PATCH>
__global__ __launch_bounds__(256, 2) void conv_vt_kernel(const VtArgs a) {
    const int L = vt_logical_id();
    vt_body<DT, NF, EPI, MODE, PATCH>(a, L / a.n_cblk, L % a.n_cblk, a.cp, a.cq);
}

// The four parity classes of a stride-2 data gradient in ONE launch: logical id = (pixel tile, class, cout block), so the four classes
// of a dY tile (and its cout blocks) run next to each other on one XCD and three of the four reads of the tile come out of its L2 --
// as four launches every class streamed all of dY from HBM again (dX 320 x 320 x 80: 1.05 GB of reads for 0.26 GB of dY).
template <int DT, int NF, int EPI, bool PATCH>
__global__ __launch_bounds__(256, 2) void conv_vt_dgrad4_kernel(const VtArgs a) {
    const int L = vt_logical_id();
    const int cblk = L % a.n_cblk;
    const int r = L / a.n_cblk;
    const int cls = r & 3, pblk = r >> 2;
    if (cls == 0) vt_body<DT, NF, EPI, VT_CLASS11, PATCH>(a, pblk, cblk, 1, 1);
    else if (cls == 1) vt_body<DT, NF, EPI, VT_CLASS10, PATCH>(a, pblk, cblk, 1, 0);
    else if (cls == 2) vt_body<DT, NF, EPI, VT_CLASS01, PATCH>(a, pblk, cblk, 0, 1);
    else vt_body<DT, NF, EPI, VT_CLASS00, PATCH>(a, pblk, cblk, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------------------------
struct VtPlan {
    bool ok, patch;
    int nf;
    size_t lds;
};

#if defined(CDET_EXPERIMENTS) && defined(CDET_RB160_AS_96)
static int vt_row_block(int rows) { return (rows <= 96 || rows == 160) ? 96 : 160; }  // (experiment builds only: see conv_halo.hip::row_block)
#else
static int vt_row_block(int rows) { return rows <= 96 ? 96 : 160; }
#endif

// d describes the FORWARD stride-2 convolution (Hs x Ws x Cs -> Hd x Wd x Cd) for mode 0, or the data gradient (CDET_CONV_DGRAD
// convention of cdet_conv2d: source = dY [Hs x Ws x Cs], destination = dX [Hd x Wd x Cd] = 2 Hs x 2 Ws) for the class launches
static VtPlan vt_plan(const cdet_conv_desc* d, bool dgrad) {
    VtPlan pl = {};
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1)) return pl;
    const int Hp = dgrad ? d->Hs : d->Hd, Wp = dgrad ? d->Ws : d->Wd;  // tile space
    const int Hb = dgrad ? d->Hd : d->Hs, Wb = dgrad ? d->Wd : d->Ws;  // the big image
    if (Hb != 2 * Hp || Wb != 2 * Wp) return pl;
    if (d->Cs % 8 != 0 || d->src_ld % 8 != 0 || d->src_coff % 8 != 0) return pl;
    if (d->Cd % 8 != 0 || d->dst_ld % 8 != 0 || d->dst_coff % 8 != 0) return pl;
    if (!(d->dtype == CDET_BF16 || d->dtype == CDET_F16)) return pl;
    if (d->out_dtype != d->dtype && d->out_dtype != CDET_F32) return pl;   // 16-bit in == out, or an fp32 destination (HEPI_F32)
    if (d->accumulate && d->out_dtype != CDET_F32) return pl;
    if (d->out_dtype == CDET_F32 && (int64_t)d->N * d->Hd * d->Wd * d->dst_ld >= (1ll << 31)) return pl;
    pl.nf = vt_row_block(d->Cd) / 32;
    pl.patch = Hp % PATCH_W == 0 && Wp % PATCH_W == 0;
    if (!pl.patch && HP + Wp + 1 > VT_XR) return pl;  // linear halo: 256 pixels + one row + one pixel
    const int64_t M = (int64_t)d->N * Hp * Wp;
    if (M >= (1ll << 31) - HP) return pl;
    if ((int64_t)d->N * d->Hs * d->Ws * d->src_ld * 2 >= 0xC0000000ll) return pl;
    if ((int64_t)d->N * d->Hd * d->Wd * d->dst_ld * 2 >= 0x7fffffffll * 2) return pl;
    const int rb = pl.nf * 32;
    const int64_t wb = (int64_t)div_up(d->Cd, rb) * div_up(d->Cs, 32) * 9 * rb * HROW;
    if (wb >= 0xC0000000ll) return pl;
    pl.lds = (size_t)HZERO + 2 * (size_t)VT_XR * HROW + 3 * (size_t)rb * HROW + 1024;  // + the dump piece
    const size_t epi = (size_t)HZERO + VEPI_STAGE_OFF + 4 * 32 * (size_t)(rb * 2 + 16) + 4 * 32 * sizeof(int);
    if (pl.lds < epi) pl.lds = epi;
    pl.ok = true;
    return pl;
}

template <int DT, int NF, int EPI, int MODE, bool PATCH>
static void launch_vt(const VtArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_vt_kernel<DT, NF, EPI, MODE, PATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((conv_vt_kernel<DT, NF, EPI, MODE, PATCH>), dim3(nblocks), dim3(256), lds, s, a);
}

template <int DT, int NF, int MODE>
static void dispatch_vt2(const VtArgs& a, int full, bool patch, size_t lds, int nblocks, hipStream_t s) {
    if (full == HEPI_F32) {
        if (patch) launch_vt<DT, NF, HEPI_F32, MODE, true>(a, lds, nblocks, s);
        else launch_vt<DT, NF, HEPI_F32, MODE, false>(a, lds, nblocks, s);
    } else if (full) {
        if (patch) launch_vt<DT, NF, HEPI_FULL, MODE, true>(a, lds, nblocks, s);
        else launch_vt<DT, NF, HEPI_FULL, MODE, false>(a, lds, nblocks, s);
    } else {
        if (patch) launch_vt<DT, NF, HEPI_RAW, MODE, true>(a, lds, nblocks, s);
        else launch_vt<DT, NF, HEPI_RAW, MODE, false>(a, lds, nblocks, s);
    }
}

template <int DT, int MODE>
static void dispatch_vt(const VtArgs& a, int nf, int full, bool patch, size_t lds, int nblocks, hipStream_t s) {
    if (nf == 5) dispatch_vt2<DT, 5, MODE>(a, full, patch, lds, nblocks, s);
    else dispatch_vt2<DT, 3, MODE>(a, full, patch, lds, nblocks, s);
}

template <int DT, int NF, int EPI, bool PATCH>
static void launch_dgrad4(const VtArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_vt_dgrad4_kernel<DT, NF, EPI, PATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((conv_vt_dgrad4_kernel<DT, NF, EPI, PATCH>), dim3(nblocks), dim3(256), lds, s, a);
}

template <int DT>
static void dispatch_dgrad4(const VtArgs& a, int nf, int full, bool patch, size_t lds, int nblocks, hipStream_t s) {
#define CDET_D4(NF_, EPI_)                                                   \
    do {                                                                     \
        if (patch) launch_dgrad4<DT, NF_, EPI_, true>(a, lds, nblocks, s);   \
        else launch_dgrad4<DT, NF_, EPI_, false>(a, lds, nblocks, s);        \
    } while (0)
    if (nf == 5) {
        if (full == HEPI_F32) CDET_D4(5, HEPI_F32); else if (full) CDET_D4(5, HEPI_FULL); else CDET_D4(5, HEPI_RAW);
    } else {
        if (full == HEPI_F32) CDET_D4(3, HEPI_F32); else if (full) CDET_D4(3, HEPI_FULL); else CDET_D4(3, HEPI_RAW);
    }
#undef CDET_D4
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_conv2d_s2_tiled_ok(const cdet_conv_desc* d) {
    if (!d) return 0;
    return vt_plan(d, d->mode == CDET_CONV_DGRAD).ok ? 1 : 0;
}

extern "C" int cdet_conv2d_s2_tiled_stat_blocks(const cdet_conv_desc* d) { return div_up((int64_t)d->N * d->Hd * d->Wd, HP); }

static void vt_fill(VtArgs& a, const cdet_conv_desc* d, const VtPlan& pl, bool dgrad, const void* x, const void* w, const float* scale,
                    const float* bias, const void* residual, void* y, float* stats) {
    const int rb = pl.nf * 32;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w; a.scale = scale; a.bias = bias; a.res = (const uint16_t*)residual;
    a.y = y; a.stats = stats; a.fold = nullptr;
    a.Hs = d->Hs; a.Ws = d->Ws; a.Hd = d->Hd; a.Wd = d->Wd;
    a.Hp = dgrad ? d->Hs : d->Hd; a.Wp = dgrad ? d->Ws : d->Wd;
    a.Cd = d->Cd;
    a.M = d->N * a.Hp * a.Wp;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dst_ld = d->dst_ld; a.dst_coff = d->dst_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.nchunk = div_up(d->Cs, 32);
    a.Cs = d->Cs;
    a.n_pblk = div_up(a.M, HP);
    a.n_cblk = div_up(d->Cd, rb);
    a.act = d->act;
    a.tiles_x = a.Wp / PATCH_W;
    a.tiles_per_img = (a.Hp / PATCH_W) * (a.Wp / PATCH_W);
    a.cp = a.cq = 0;
    a.x_bytes = (unsigned)((int64_t)d->N * d->Hs * d->Ws * d->src_ld * 2);
    a.w_bytes = (unsigned)((int64_t)a.n_cblk * a.nchunk * 9 * rb * HROW);
    a.accum = d->accumulate ? 1 : 0;
}

static int s2_tiled_impl(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias, const void* residual,
                         void* y, float* stats, void* stream, const BnFold* fold);

extern "C" int cdet_conv2d_s2_tiled(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                                    const void* residual, void* y, float* stats, void* stream) {
    return s2_tiled_impl(d, x, w_tiled, scale, bias, residual, y, stats, stream, nullptr);
}

static int s2_tiled_impl(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias, const void* residual,
                         void* y, float* stats, void* stream, const BnFold* fold) {
    CDET_CHECK_ARG(d && x && w_tiled && y, "cdet_conv2d_s2_tiled: null pointer");
    CDET_CHECK_ARG(d->mode == CDET_CONV_FWD, "cdet_conv2d_s2_tiled: forward descriptor expected");
    const VtPlan pl = vt_plan(d, false);
    CDET_CHECK_ARG(pl.ok, "cdet_conv2d_s2_tiled: unsupported geometry (3x3 stride 2 pad 1, even H and W, Cs/Cd/ld/coff %% 8 == 0, 16-bit in == out, "
                          "output width <= 63 or output H, W multiples of 16)");
    CDET_CHECK_ARG(!residual || (d->res_ld % 8 == 0 && d->res_coff % 8 == 0), "cdet_conv2d_s2_tiled: residual ld/coff must be multiples of 8");
    VtArgs a;
    vt_fill(a, d, pl, false, x, w_tiled, scale, bias, residual, y, stats);
    a.fold = stats ? fold : nullptr;
    CDET_CHECK_ARG(!(d->out_dtype == CDET_F32 && stats), "cdet_conv2d_s2_tiled: BatchNorm partial sums go with the 16-bit raw output");
    const int full = d->out_dtype == CDET_F32 ? HEPI_F32 : ((scale || bias || residual || d->act != CDET_ACT_NONE) ? HEPI_FULL : HEPI_RAW);
    const int nblocks = a.n_pblk * a.n_cblk;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == CDET_BF16) dispatch_vt<CDET_BF16, VT_S2FWD>(a, pl.nf, full, pl.patch, pl.lds, nblocks, s);
    else dispatch_vt<CDET_F16, VT_S2FWD>(a, pl.nf, full, pl.patch, pl.lds, nblocks, s);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_conv2d_s2_tiled_bn_ok(const cdet_conv_desc* d) {
#ifndef CDET_EXPERIMENTS
    return 0;  // (the in-launch BatchNorm fold is an experiment build's: bn_fold.h)
#endif
    if (!d || !cdet_conv2d_s2_tiled_ok(d) || d->out_dtype == CDET_F32 || d->mode != CDET_CONV_FWD) return 0;
    const VtPlan pl = vt_plan(d, false);
    return pl.ok && cdet_conv2d_s2_tiled_stat_blocks(d) <= BNF_CL * BNF_MAX_CL && div_up(d->Cd, pl.nf * 32) <= BNF_MAX_CB;
}

// train form with the BatchNorm statistics finished inside the launch (bn_fold.h)
extern "C" int cdet_conv2d_s2_tiled_bn(const cdet_conv_desc* d, const void* x, const void* w_tiled, void* y, float* stats, const cdet_bn_fold* fold_dev,
                                       void* stream) {
    CDET_CHECK_ARG(stats, "cdet_conv2d_s2_tiled_bn: the partial-sum rows are needed with or without the fold");
#ifndef CDET_EXPERIMENTS
    CDET_CHECK_ARG(!fold_dev, "cdet_conv2d_s2_tiled_bn: the in-launch BatchNorm fold is compiled into -DCDET_EXPERIMENTS builds only (cdet_has_experiments())");
#endif
    CDET_CHECK_ARG(!fold_dev || cdet_conv2d_s2_tiled_bn_ok(d), "cdet_conv2d_s2_tiled_bn: this launch has more partial rows / column blocks than the fold takes");
    return s2_tiled_impl(d, x, w_tiled, nullptr, nullptr, nullptr, y, stats, stream, fold_dev);
}

extern "C" int cdet_conv2d_s2_tiled_dgrad(const cdet_conv_desc* d, const void* dy, const void* w_dgrad_tiled, const float* scale, const float* bias,
                                          const void* residual, void* dx, float* stats, void* stream) {
    CDET_CHECK_ARG(d && dy && w_dgrad_tiled && dx, "cdet_conv2d_s2_tiled_dgrad: null pointer");
    CDET_CHECK_ARG(!scale && !bias && !stats, "cdet_conv2d_s2_tiled_dgrad: scale / bias / stats must be NULL");
    CDET_CHECK_ARG(d->mode == CDET_CONV_DGRAD, "cdet_conv2d_s2_tiled_dgrad: CDET_CONV_DGRAD descriptor expected (source = dY, destination = dX)");
    const VtPlan pl = vt_plan(d, true);
    CDET_CHECK_ARG(pl.ok, "cdet_conv2d_s2_tiled_dgrad: unsupported geometry");
    CDET_CHECK_ARG(!residual || (d->res_ld % 8 == 0 && d->res_coff % 8 == 0), "cdet_conv2d_s2_tiled_dgrad: residual ld/coff must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    VtArgs a;
    vt_fill(a, d, pl, true, dy, w_dgrad_tiled, nullptr, nullptr, residual, dx, nullptr);
    const int full = d->out_dtype == CDET_F32 ? HEPI_F32 : (residual != nullptr ? HEPI_FULL : HEPI_RAW);
    const int nblocks = a.n_pblk * 4 * a.n_cblk;  // (pixel tile, class, cout block)
    if (d->dtype == CDET_BF16) dispatch_dgrad4<CDET_BF16>(a, pl.nf, full, pl.patch, pl.lds, nblocks, s);
    else dispatch_dgrad4<CDET_F16>(a, pl.nf, full, pl.patch, pl.lds, nblocks, s);
    CDET_LAUNCH_CHECK();
    return 0;
}
