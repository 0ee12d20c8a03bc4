// 1x1 convolutions with Cout >= 320 (the cv1 / cv2 of every C2f, SPPF; models/common.py:174-191, 230-245): the tap-resident kernel of
// conv_halo.hip with the pixel tile SHARED by two cout blocks.
//
// A 1x1 convolution re-uses a staged pixel for one K step only, so conv_halo.hip's 256 px x 160 cout workgroup streams 16 KiB of
// pixels + 10 KiB of weights through the LDS-DMA per 32-deep step; two such workgroups per CU need ~40 B/clk -- what the DMA path of a
// CU delivers (conv_wgrad_halo.hip measured the same ceiling) -- and the 1x1 layers sat at 600-900 TF/s where the 3x3 layers reach
// 1050-1230. Here ONE workgroup of eight waves per CU owns 256 pixels x 320 couts: waves 0-3 and 4-7 compute the two 160-cout blocks
// from the SAME pixel buffers, which cuts the stream to 16 + 20 KiB per two blocks (28 B/clk). Everything else is conv_halo.hip's
// pipeline: v_mfma_f32_32x32x16, 160 fp32 accumulators per lane, 3 pixel buffers (chunk st + 2 is fetched during step st), 3-stage
// weight ring, counted vmcnt, one barrier per K step, BN partial sums / scale-bias-SiLU-residual epilogue with LDS-staged row stores.
// Weight pieces are spread over the eight waves without a divergent branch (ids beyond the tile go through an empty descriptor into
// a dump area; see conv_vt.hip for what a `lane < 32` half piece made hipcc do).
#include <stdlib.h>

#include "halo_common.h"
#include "bn_fold.h"

namespace cdet {

struct PairArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* bias;
    const uint16_t* res;
    void* y;
    float* stats;
    const BnFold* fold;  // see conv_halo.hip
    int Cd, M;
    int src_ld, src_coff, dst_ld, dst_coff, res_ld, res_coff;
    int nchunk, Cs, n_pblk, n_pair, act;
    unsigned x_bytes, w_bytes;
    CatSrcs cat;  // CAT instantiations: the source is a virtual Concat of up to three buffers (halo_common.h)
};

constexpr int PR_XB = HP * HROW;          // one pixel buffer: 256 rows of 64 B
constexpr int PR_STAGE_OFF = 13312;       // epilogue LDS map: statistics scratch [8][2][160] fp32 (10 KiB), scale / bias [2][320] fp32, staging

template <int DT, int EPI, bool CAT = false>
__global__ __launch_bounds__(512, 1) void conv_pair_kernel(const PairArgs a) {
    constexpr int NF = 5, NG = 2, HC = 160;
    constexpr int WTILE = HC * HROW;      // one cout block's tile: 10 KiB
    constexpr int WPC = 2 * WTILE / 1024; // 1-KiB pieces of the two tiles of a step: 20
    constexpr int NWP = (WPC + 7) / 8;    // per wave: 3 (24 slots, four of them dummies)
    constexpr int NXP = 2;                // pixel pieces per wave and chunk (16 pieces of 16 rows over eight waves)
    constexpr int NM = NG * NF, NR = NF + NG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pw = wave & 3, cb = wave >> 2;  // pixel quarter of the tile, cout block of the pair
    const int l31 = lane & 31, h = lane >> 5;

    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // (the quotient / remainder come out of VALU sequences: tell the compiler they are wave-uniform, or every DMA whose scalar offset
    //  depends on them is wrapped in a waterfall loop)
    const int pblk = __builtin_amdgcn_readfirstlane(L / a.n_pair);
    const int pair = __builtin_amdgcn_readfirstlane(L - pblk * a.n_pair);
    const int c0 = (pair * 2 + cb) * HC;
    const int p0 = pblk * HP;
    unsigned char* const xbase = smem + HZERO;
    unsigned char* const wbase = xbase + 3 * PR_XB;
    unsigned char* const wdump = wbase + 3 * 2 * WTILE;
    if (t < 16) reinterpret_cast<uint32_t*>(smem)[t] = 0u;  // zero row

    unsigned xvoff[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int hrow = 16 * (8 * i + wave) + (lane >> 2);
        const int g = p0 + hrow;
        const unsigned off = ((unsigned)g * (unsigned)a.src_ld + (unsigned)a.src_coff) * 2u + ((unsigned)((lane & 3) ^ ((hrow >> 2) & 3)) << 4);
        xvoff[i] = g < a.M ? off : HSENT;
    }
    unsigned xv1[NXP], xv2[NXP];  // CAT: the same pieces inside segments 1 and 2 (segment 0 lives in xvoff)
    if (CAT) {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int hrow = 16 * (8 * i + wave) + (lane >> 2);
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
    const bool partial = (a.Cs & 31) != 0;
    const int xls = (lane & 3) ^ ((lane >> 4) & 3);

    // (the segment table is captured BY VALUE: through by-reference captures the selects below became selects of addresses inside a
    //  closure object kept in private memory)
    auto dma_x = [&, xvoff, xv1, xv2, cl0, cl1, cl2, ch0, ch1, ch2, cat_c1, cat_c2](int i, int chunk, int xb) {
        unsigned char* dst = xbase + xb * PR_XB + (8 * i + wave) * 1024;
        if constexpr (CAT) {  // the chunk's segment (wave-uniform): its buffer, its pixel offsets, the chunk's position inside it.
            // The segment table lives in VGPRs (uniform values, selected by v_cndmask and read back with v_readfirstlane): as scalars the
            // three pointers / extents / boundaries stayed live through the whole K loop, the kernel ran out of SGPRs, and the spill
            // code's scratch loads drained the DMA queue every step (3x slower than the plain form)
            const bool sg2 = chunk >= cat_c2, sg1 = chunk >= cat_c1;
            // every candidate is read into a register and made opaque BEFORE the select: a select of loads from the (by-value) closure is
            // rewritten by the optimiser into one load at a selected address, the closure then stays in private memory, and with it the
            // kernel arguments it refers to -- their scratch loads share the DMA's counter (the first CAT build ran 3x slower)
            unsigned l0 = cl0, l1 = cl1, l2 = cl2, h0 = ch0, h1 = ch1, h2 = ch2;
            unsigned t0 = xvoff[i], t1 = xv1[i], t2 = xv2[i];
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
        unsigned v = xvoff[i];
        if (partial && chunk * 32 + 8 * xls >= a.Cs) v = HSENT;
        dma16<CDET_HALO_X_AUX>(rs, v, (unsigned)chunk * 64u, dst);
    };
    // piece j of this wave of the two weight tiles of K step `step` -> ring stage `stage`
    auto dma_w = [&](int step, int stage, int j) {
        const int id = 8 * j + wave;  // wave-uniform
        const bool real = id < WPC;
        const int cbsel = id >= WPC / 2 ? 1 : 0, pid = id - cbsel * (WPC / 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (real && step < a.nchunk) ? (int)a.w_bytes : 0, 0x00020000);
        unsigned char* dst = real ? wbase + stage * (2 * WTILE) + id * 1024 : wdump;
        const unsigned soff = ((unsigned)((pair * 2 + cbsel) * a.nchunk + step)) * (unsigned)WTILE;
        dma16(rs, (unsigned)((real ? pid : 0) * 1024 + lane * 16), soff, dst);
    };

    // ---- prologue DMA: pixel chunks 0 and 1, weight tiles of steps 0 .. 2 -- issued in front of the fragment offsets and the accumulator
    //      initialisation (round 6: ~400 VALU cycles under the fetch latency); waited for right before the first barrier
#pragma unroll
    for (int i = 0; i < NXP; ++i) dma_x(i, 0, 0);
#pragma unroll
    for (int j = 0; j < NWP; ++j) dma_w(0, 0, j);
#pragma unroll
    for (int i = 0; i < NXP; ++i) dma_x(i, 1, 1);
#pragma unroll
    for (int j = 0; j < NWP; ++j) dma_w(1, 1, j);
#pragma unroll
    for (int j = 0; j < NWP; ++j) dma_w(2, 2, j);

    const int aoff0 = cb * WTILE + l31 * HROW + ((h ^ ((l31 >> 2) & 3)) << 4);
    int bo0[NG];  // byte offset of the lane's pixel rows inside a pixel buffer (0: the zero row)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = pw * 64 + g * 32 + l31;
        bo0[g] = p0 + i < a.M ? i * HROW + ((h ^ ((i >> 2) & 3)) << 4) + HZERO : -1;
    }

    f32x16 acc[NF][NG];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;

    wait_vm(NXP + 2 * NWP);  // chunk 0 and tile 0 have landed (the prologue's DMA was issued above, in front of the accumulator initialisation)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    auto frag = [&](const unsigned char* ws_, int xb, int s_, int i, u32x4 (&af)[NF], u32x4 (&bf)[NG]) {
        if (i < NG) {
            const int bo = bo0[i] < 0 ? 0 : bo0[i] + xb * PR_XB;
            bf[i] = *reinterpret_cast<const u32x4*>(smem + (bo ^ (s_ << 5)));
        } else {
            af[i - NG] = *reinterpret_cast<const u32x4*>(ws_ + ((aoff0 ^ (s_ << 5)) + (i - NG) * 32 * HROW));
        }
    };

    u32x4 a0[NF], b0[NG], a1[NF], b1[NG];
#pragma unroll
    for (int i = 0; i < NR; ++i) frag(wbase, 0, 0, i, a0, b0);

    // One K step = one 32-channel chunk; u = st % 3 = ring stage of its weight tiles = its pixel buffer (compile-time after unrolling)
    auto step = [&](int st, int u) {
        const unsigned char* ws = wbase + u * (2 * WTILE);
        const unsigned char* wsn = wbase + ((u + 1) % 3) * (2 * WTILE);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase A: MFMAs of k16 #0; fragment reads of k16 #1; the pixels of chunk st + 2 into the buffer chunk st - 1 used
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma32<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);
            if (2 * i < NR) frag(ws, u, 1, 2 * i, a1, b1);
            if (2 * i + 1 < NR) frag(ws, u, 1, 2 * i + 1, a1, b1);
            if (i < NXP) dma_x(i, st + 2, (u + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile st + 1 (issued two phase-Bs ago) and chunk st + 1 (issued in the previous phase A) have landed; the newest tile and this
        // phase's pixel pieces may stay in flight
        wait_vm_lgkm0<NWP + NXP>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: MFMAs of k16 #1; the tiles of step st + 3 into the stage just freed; fragment reads of (st + 1, k16 #0)
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma32<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
            if (i == 0) dma_w(st + 3, u, 0);
            if (i == 3) dma_w(st + 3, u, 1);
            if (i == 6) dma_w(st + 3, u, 2);
            if (i >= 1 && 2 * (i - 1) < NR) frag(wsn, (u + 1) % 3, 0, 2 * (i - 1), a0, b0);
            if (i >= 1 && 2 * (i - 1) + 1 < NR) frag(wsn, (u + 1) % 3, 0, 2 * (i - 1) + 1, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c3 = 0; c3 < a.nchunk; c3 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (c3 + u < a.nchunk) step(c3 + u, u);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- BN statistics of the raw convolution (train mode): per (pixel block, channel) partial sums ---------------------------------------
    if (a.stats != nullptr) {
        float* stl = reinterpret_cast<float*>(smem + HZERO);  // [8 waves][2][HC]
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
        if (t < 2 * HC) {
            const int cbh = t / HC, c = t - cbh * HC;
            const int co = (pair * 2 + cbh) * HC + c;
            if (co < a.Cd) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    sv += stl[((cbh * 4 + m) * 2 + 0) * HC + c];
                    qv += stl[((cbh * 4 + m) * 2 + 1) * HC + c];
                }
                if (CDET_FOLD(a.fold)) {
                    const __amdgpu_buffer_rsrc_t rs_ = bnf_rsrc(a.stats);
                    bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 0) * a.Cd + co) * 4), sv);
                    bnf_stf(rs_, (unsigned)((((int64_t)pblk * 2 + 1) * a.Cd + co) * 4), qv);
                } else {
                    a.stats[((int64_t)pblk * 2 + 0) * a.Cd + co] = sv;
                    a.stats[((int64_t)pblk * 2 + 1) * a.Cd + co] = qv;
                }
            }
        }
    }

    // ---- epilogue (as conv_halo.hip) ----------------------------------------------------------------------------------------------------------
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    float* const sbl = reinterpret_cast<float*>(smem + HZERO + 10240) + cb * (2 * HC);  // [2][HC] per cout block
    if (EPI == HEPI_FULL) {
        if (t < 2 * HC) {
            const int cbh = t / HC, c_ = t - cbh * HC;
            const int cg = (pair * 2 + cbh) * HC + c_;
            const int c = cg < a.Cd ? cg : a.Cd - 1;
            float* d = reinterpret_cast<float*>(smem + HZERO + 10240) + cbh * (2 * HC);
            d[c_] = a.scale ? a.scale[c] : 1.f;
            d[HC + c_] = a.bias ? a.bias[c] : 0.f;
        }
        __syncthreads();
    }
    constexpr int RS = HC * 2 + 16;
    constexpr int CH = HC / 8;
    unsigned char* const stg = smem + HZERO + PR_STAGE_OFF + wave * (32 * RS);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int p = p0 + pw * 64 + g * 32 + l31;
        const bool pok = p < a.M;
        const int64_t rb = (int64_t)p * a.res_ld + a.res_coff;
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
                if (EPI == HEPI_FULL) {
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
                u32x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = hpack2<DT>(v[2 * r], v[2 * r + 1]);
                *reinterpret_cast<u32x4*>(stg + l31 * RS + cl * 2) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
            const int id = it * 64 + lane;
            const int px = id / CH, c = id - px * CH;
            if (id < 32 * CH) {
                const u32x4 pk = *reinterpret_cast<const u32x4*>(stg + px * RS + c * 16);
                const int po = p0 + pw * 64 + g * 32 + px;
                const int co = c0 + 8 * c;
                if (po < a.M && co < a.Cd) *reinterpret_cast<u32x4*>(yp + (int64_t)po * a.dst_ld + a.dst_coff + co) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // train form with the statistics finished in this launch (bn_fold.h): the workgroup's row share (both cout blocks: 2 HC columns) went out with
    // write-through stores above; tickets at the very end, when the staging LDS is free
    if (EPI == HEPI_RAW && a.stats != nullptr && CDET_FOLD(a.fold))
        bn_fold_finish<true>(a.fold, a.stats, pblk, pair * 2 * HC, 2 * HC, pair, reinterpret_cast<volatile int*>(smem));
}

template <int DT, int EPI, bool CAT = false>
static void launch_pair(const PairArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_pair_kernel<DT, EPI, CAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((conv_pair_kernel<DT, EPI, CAT>), dim3(nblocks), dim3(512), lds, s, a);
}

// 1x1, stride 1, 16-bit in == out, an even number (>= 2) of 160-cout blocks, enough pixel tiles to give every CU a workgroup
bool pair_plan_ok(const cdet_conv_desc* d) {
    const int mode = sw(SW_CONV_PAIR);  // 0: never (A/B timing), 2: whenever the geometry allows (tests)
    if (mode == 0) return false;
    if (!(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0)) return false;
    if (d->Cd < 320) return false;
    const int n_cblk = div_up(d->Cd, 160);
    if (n_cblk % 2 != 0) return false;
    const int64_t M = (int64_t)d->N * d->Hs * d->Ws;
    // Measured (tools/conv_tiled_bench.py, batch 32): 40 x 40 1600 -> 640 936 -> 1038 TF/s, 40 x 40 960 -> 640 799 -> 832, 640 -> 640 685 -> 714,
    // 80 x 80 960 -> 320 722 -> 741, but 80 x 80 320 -> 320 (10 K steps) 562 -> 508: sharing the pixel tile pays where the K loop is long
    // enough to amortise the lock-step of eight waves on one barrier; five pixel buffers (four steps of lead) changed nothing -- the short
    // layers are not latency-bound.
    return mode == 2 || (d->Cs >= 640 && div_up(M, HP) * (n_cblk / 2) >= 256);
}

int pair_launch(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias, const void* residual, void* y,
                float* stats, hipStream_t s, const CatSrcs* cat, const BnFold* fold) {
    PairArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_tiled; a.scale = scale; a.bias = bias; a.res = (const uint16_t*)residual;
    a.y = y; a.stats = stats; a.fold = stats ? fold : nullptr;
    a.Cd = d->Cd;
    a.M = d->N * d->Hs * d->Ws;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dst_ld = d->dst_ld; a.dst_coff = d->dst_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.nchunk = div_up(d->Cs, 32);
    a.Cs = d->Cs;
    a.n_pblk = div_up(a.M, HP);
    const int n_cblk = div_up(d->Cd, 160);
    a.n_pair = n_cblk / 2;
    a.act = d->act;
    a.x_bytes = (unsigned)((int64_t)a.M * d->src_ld * 2);
    a.w_bytes = (unsigned)((int64_t)n_cblk * a.nchunk * 160 * HROW);
    const size_t loop = (size_t)HZERO + 3 * (size_t)PR_XB + 3 * 2 * (size_t)(160 * HROW) + 1024;
    const size_t epi = (size_t)HZERO + PR_STAGE_OFF + 8 * 32 * (size_t)(160 * 2 + 16);
    const size_t lds = loop > epi ? loop : epi;
    const bool full = scale || bias || residual || d->act != CDET_ACT_NONE;
    const int nblocks = a.n_pblk * a.n_pair;
    if (cat) {
        a.cat = *cat;
        if (d->dtype == CDET_BF16) launch_pair<CDET_BF16, HEPI_FULL, true>(a, lds, nblocks, s);
        else launch_pair<CDET_F16, HEPI_FULL, true>(a, lds, nblocks, s);
        return 0;
    }
    a.cat.n = 0;
    if (d->dtype == CDET_BF16) {
        if (full) launch_pair<CDET_BF16, HEPI_FULL>(a, lds, nblocks, s);
        else launch_pair<CDET_BF16, HEPI_RAW>(a, lds, nblocks, s);
    } else {
        if (full) launch_pair<CDET_F16, HEPI_FULL>(a, lds, nblocks, s);
        else launch_pair<CDET_F16, HEPI_RAW>(a, lds, nblocks, s);
    }
    return 0;
}

}  // namespace cdet
