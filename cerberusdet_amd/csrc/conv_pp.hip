// Two-phase "ping-pong" form of the tap-resident stride-1 3x3 convolution (conv_halo.hip) for the 160-cout tile: ONE 8-wave workgroup per CU, two
// waves per SIMD that are kept in OPPOSITE phases for the whole K loop.
//
// Why (profiles/r05_halo_ng4.txt, section 1): in conv_halo_kernel the two waves of a SIMD belong to two independent 4-wave
// workgroups. Each K step of a wave interleaves its 20 MFMAs with 14 fragment reads, 3-4 LDS-DMA issues and a counted wait + barrier; nothing keeps
// one wave in its MFMA stretch while the other sits in its wait -- on 40 x 40 320 -> 320 the loop's MFMAs alone take 59 us, everything else alone 48 us,
// and the two overlap by only ~10 us (MFMA pipe busy 46 % of the launch). Here the halves of the workgroup are offset by ONE barrier in the prologue and
// every K step is split into
//   * a MEMORY phase  -- all 14 ds_read_b128 of the step (both k16 halves) into the single fragment register set, the step's LDS-DMA issues (this
//                        wave's share of the weight tile two steps ahead, at most one pixel piece of the next channel chunk), the tap offsets of the
//                        next step, one counted vmcnt wait -- and
//   * a COMPUTE phase -- the step's 20 v_mfma_f32_32x32x16 back to back at raised priority, nothing else,
// each closed by the workgroup's ordinary s_barrier. While waves 0-3 (group 0) compute, waves 4-7 (group 1) fetch, and vice versa: every SIMD always
// holds exactly one wave in its MFMA stretch. No named barriers are needed -- group 1 executes one extra s_barrier before its loop, group 0 one behind.
//
// The two groups own two NEIGHBOURING pixel tiles of the SAME cout block: the weight ring is shared (one 10 KiB tile per K step for 512 pixels instead
// of one per 256: half the weight LDS-DMA bytes, LDS writes and L2 reads per FLOP -- DESIGN.md section 8 (0)), every wave copies 1/8 of each tile.
// Per K step a wave issues 2 weight DMA instructions (1 KiB + 256 B) and at most one pixel piece, against 3 + 1 in conv_halo_kernel.
//
// Arithmetic: the same tile partition (256-pixel tiles / 16 x 16 patches, 160-cout blocks), the same K order per accumulator (chunk-outer, tap-inner,
// k16 #0 then #1) and the same epilogue as conv_halo_kernel<DT, 9, 5, EPI, 3, PATCH, 2> -- results and BatchNorm partial-sum rows are bit-identical.
//
// Schedule (n = K steps; T(j) = weight tile of step j, ring slot j % 3; B = s_barrier of all eight waves):
//   prologue : both groups fetch chunk 0 of their pixel tile and their shares of T(0), T(1); vmcnt(0); B
//   group 0  :    M(0) B C(0) B M(1) B C(1) B ... M(n-1) B C(n-1) B          | vmcnt(0) B  epilogue
//   group 1  : B  M(0) B C(0) B M(1) B ...           M(n-1) B C(n-1)         | vmcnt(0) B  epilogue
//   M(j) issues this wave's share of T(j+2) into slot (j+2) % 3 -- T(j-1), its previous content, was last read by group 1 in ITS M(j-1), one phase before
//   group 0's M(j) -- and waits (counted) for everything issued before M(j): T(j+1) is complete, for both groups, one barrier before anyone reads it.
#include <stdlib.h>

#include "halo_common.h"

namespace cdet {

struct PpArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* bias;
    const uint16_t* res;
    void* y;
    float* stats;
    int H, W, Cd;
    int M;  // N*H*W
    int src_ld, src_coff, dst_ld, dst_coff, res_ld, res_coff;
    int nchunk;  // ceil(Cs / 32)
    int Cs;      // reduction channels (a last partial chunk is zero-filled: multiples of 8)
    int n_pblk, n_cblk, n_ppair;
    int act;
    int XH;                      // halo rows per pixel buffer (multiple of 16)
    int tiles_x, tiles_per_img;  // patch mode: 16 x 16 patches per image row / per image
    unsigned x_bytes, w_bytes;
};

constexpr int PP_EREG = 51200;        // epilogue LDS region of one group: zero row, statistics scratch, scale / bias, store staging (conv_halo.hip's map)
constexpr int PP_STAGE_OFF = 6912;    // (= HEPI_STAGE_OFF)

#ifdef CDET_PROFILING
// per-workgroup record for tools/pp_timeline.py: [t_start, t_loop, t_epilogue, t_end, sum of group 0's memory-phase bodies, of its memory phases incl. the
// barrier wait, of its compute phases incl. the barrier wait, xcc id] (s_memtime clocks, wave 0)
__device__ unsigned long long* g_pp_dbg = nullptr;
#endif

// ABL (timing experiments, -DCDET_PROFILING builds only): 1 = no DMA in the loop, 2 = no fragment reads, 4 = no MFMA, 8 = no s_setprio around the compute phase
// (results stay correct), 32 = per-phase s_memtime stamps
// (tools/pp_timeline.py; the stamps themselves stretch every phase -- read them as proportions)
template <int DT, int EPI, bool PATCH, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv_pp_kernel(const PpArgs a) {
    constexpr bool TL = (ABL & 32) != 0;
    constexpr int NF = 5, NG = 2;
    constexpr int HC = NF * 32;           // couts per block
    constexpr int WTILE = HC * HROW;      // 10240 bytes per (cblk, chunk, tap) weight tile
    constexpr int WSH = WTILE / 8;        // 1280 bytes of every tile per wave: one 1 KiB piece + one 256-byte piece (lanes 0-15)
    constexpr int NM = NG * NF;           // MFMAs per k16 half
    constexpr int NR = NF + NG;           // fragment reads per k16 half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2;           // 0: computes in the odd phases, 1: in the even ones
    const int wave = wave8 & 3;           // wave inside the group
    const int t = tid & 255;              // thread inside the group
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware remap (bijective): consecutive logical ids -- the cout blocks of one pixel-tile pair, then the next pair -- run on ONE XCD
    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int cblk = L % a.n_cblk;
    const int pblk = (L / a.n_cblk) * 2 + grp;
    const bool live = pblk < a.n_pblk;    // (an odd number of pixel tiles: the last pair's second group only carries its share of the weight stream)
    const int c0 = cblk * HC;
    const int W = a.W;
    const int XHB = a.XH * HROW;
    const int nsteps = a.nchunk * 9;
    const int tpitch = PATCH ? PATCH_HPW : W;
    const int halo0 = tpitch + 1;
    const int p0 = pblk * HP;      // linear mode
    int pn = 0, py0 = 0, px0 = 0;  // patch mode: image, top-left pixel
    if (PATCH) {
        pn = pblk / a.tiles_per_img;
        const int r = pblk - pn * a.tiles_per_img;
        py0 = (r / a.tiles_x) * 16;
        px0 = (r % a.tiles_x) * PATCH_W;
    }
    const int xoff_g = HZERO + grp * 2 * XHB;              // this group's two pixel buffers (byte offset inside smem)
    unsigned char* const xbase = smem + xoff_g;
    unsigned char* const wbase = smem + HZERO + 4 * XHB;   // the shared three-stage weight ring

    if (tid < 16) reinterpret_cast<uint32_t*>(smem)[tid] = 0u;  // zero row (visible after the first barrier)
#ifdef CDET_PROFILING
    unsigned long long tp_start = 0, tp_loop = 0, tp_epi = 0, tp_mb = 0, tp_m = 0, tp_c = 0;
    unsigned long long tr_start = 0;  // s_memrealtime: 100 MHz, one counter for the whole chip (s_memtime counters differ between XCCs)
    if (g_pp_dbg != nullptr && tid == 0) {
        tp_start = __builtin_readcyclecounter();
        tr_start = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // ---- pixel DMA pieces of this wave: piece id 4*i + wave covers halo rows 16*id .. 16*id+15, 4 lanes (64 B) per row ---------------------------
    const int nxp_total = (a.XH + 15) >> 4;
    const int nxpw = (nxp_total - wave + 3) >> 2;  // wave-uniform, <= MAXXP
    unsigned xvoff[MAXXP];
#pragma unroll
    for (int i = 0; i < MAXXP; ++i) {
        const int hrow = 16 * (4 * i + wave) + (lane >> 2);
        int g;
        bool ok;
        if (PATCH) {
            const int hy = hrow / PATCH_HPW, hx = hrow - hy * PATCH_HPW;
            const int y = py0 - 1 + hy, x = px0 - 1 + hx;
            ok = live && hy < 18 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)W;  // outside the image: zeros = the padding
            g = (pn * a.H + y) * W + x;
        } else {
            g = p0 - halo0 + hrow;
            ok = live && g >= 0 && g < a.M;
        }
        const unsigned off = ((unsigned)g * (unsigned)a.src_ld + (unsigned)a.src_coff) * 2u + ((unsigned)((lane & 3) ^ ((hrow >> 2) & 3)) << 4);
        xvoff[i] = ok ? off : HSENT;
    }
    const bool partial = (a.Cs & 31) != 0;                   // wave-uniform
    const int xls = (lane & 3) ^ ((lane >> 4) & 3);          // logical 16-byte slot this lane fetches (the same for every piece)
    const unsigned wvoff = (unsigned)(wave8 * WSH + lane * 16);
    const unsigned wtile0 = (unsigned)cblk * (unsigned)nsteps * (unsigned)WTILE;

    // Steps / chunks beyond the end are requested through an EMPTY descriptor (every load returns zeros): the number of DMA instructions per phase --
    // what the counted waits rely on -- never changes.
    auto dma_w = [&](int step, int stage) {  // this wave's share of the weight tile of K step `step` -> ring stage `stage`
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, step < nsteps ? (int)a.w_bytes : 0, 0x00020000);
        unsigned char* dst = wbase + stage * WTILE + wave8 * WSH;
        const unsigned soff = wtile0 + (unsigned)step * (unsigned)WTILE;
        dma16(rs, wvoff, soff, dst);
        if (lane < 16) dma16(rs, wvoff + 1024u, soff, dst + 1024);  // the 256-byte tail of the share
    };
    auto dma_x = [&](int i, int chunk, int xb) {  // piece i of this wave, channels of `chunk` -> pixel buffer xb of the group
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, chunk < a.nchunk ? (int)a.x_bytes : 0, 0x00020000);
        unsigned char* dst = xbase + xb * XHB + (4 * i + wave) * 1024;
        unsigned v = xvoff[i] + (unsigned)chunk * 64u;
        // last partial chunk (Cs % 32 != 0): the lanes whose 8-channel slot lies beyond Cs fetch zeros (what sits there in memory is a neighbouring
        // channel slice, possibly never written, and 0-weight x NaN would still be NaN)
        if (partial && chunk * 32 + 8 * xls >= a.Cs) v = HSENT;
        dma16<CDET_HALO_X_AUX>(rs, v, 0u, dst);
    };

    // ---- prologue DMA: chunk 0 of the group's pixels, this wave's shares of T(0) and T(1). Issued HERE, in front of the tap-mask divisions and the
    //      accumulator initialisation below (~700 VALU cycles that now run under the fetch latency); waited for right before the first barrier
#pragma unroll
    for (int i = 0; i < MAXXP; ++i)
        if (i < nxpw) dma_x(i, 0, 0);
    dma_w(0, 0);
    dma_w(1, 1);

    // ---- fragment read offsets -----------------------------------------------------------------------------------------------------------------
    const int aoff0 = l31 * HROW + ((h ^ ((l31 >> 2) & 3)) << 4);  // A (weights): row f*32 + l31 of the stage, k-slot 2*s + h
    int pixh[NG], pout[NG];
    unsigned vmask[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = wave * 64 + g * 32 + l31;  // tile pixel
        unsigned m = 0u;
        if (PATCH) {
            const int iy = i / PATCH_W, ix = i % PATCH_W;
            pixh[g] = iy * PATCH_HPW + ix + halo0;
            pout[g] = (pn * a.H + py0 + iy) * W + px0 + ix;
            m = 0x1ffu;  // every halo row of a patch is the right neighbour or DMA-filled padding
        } else {
            pixh[g] = i + halo0;
            const int p = p0 + i;
            pout[g] = p;
            if (p < a.M) {
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

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prologue's DMA (issued above, in front of the index arithmetic) has landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // B-fragment byte offsets (relative to smem) of a K step: halo row of each of the lane's two pixels for the step's tap
    auto b_offsets = [&](int xoff, int tap_, int (&bo)[NG]) {
        const int dy_ = tap_ / 3 - 1, dx_ = tap_ % 3 - 1;
        // opaque copies: without them the compiler hoists the nine per-tap offset pairs (and their scalar parts) out of the chunk loop as loop
        // invariants -- 18 VGPRs + ~30 SGPRs the 256-register budget does not have
        int tp = tpitch, p_[NG];
        asm volatile("" : "+s"(tp));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            p_[g] = pixh[g];
            asm volatile("" : "+v"(p_[g]));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int hrow = p_[g] + dy_ * tp + dx_;
            const int off = hrow * HROW + ((h ^ ((hrow >> 2) & 3)) << 4);
            const bool ok = PATCH || ((vmask[g] >> tap_) & 1u);
            bo[g] = ok ? off + xoff : 0;  // invalid tap / pixel: the zero row
        }
    };
    // fragment i of a k16 half: i < NG -> pixel rows (B operand), else weight rows (A operand); read order = use order
    auto frag = [&](const unsigned char* ws_, const int (&bo)[NG], int s_, int i, u32x4 (&af)[NF], u32x4 (&bf)[NG]) {
        if (ABL & 2) {
            if (i < NG) bf[i] = u32x4{(unsigned)lane, 1u, 2u, 3u};
            else af[i - NG] = u32x4{(unsigned)lane, 1u, 2u, 3u};
        } else if (i < NG) bf[i] = *reinterpret_cast<const u32x4*>(smem + (bo[i] ^ (s_ << 5)));
        else af[i - NG] = *reinterpret_cast<const u32x4*>(ws_ + ((aoff0 ^ (s_ << 5)) + (i - NG) * 32 * HROW));
    };

    int bo_cur[NG], bo_nxt[NG];
    u32x4 a0[NF], b0[NG], a1[NF], b1[NG];
    b_offsets(xoff_g, 0, bo_cur);

#ifdef CDET_PROFILING
    if (g_pp_dbg != nullptr && tid == 0) tp_loop = __builtin_readcyclecounter();
    unsigned long long tp_prev = tp_loop;
#endif
    if (grp == 1) {  // half a step behind group 0 from here to the end of the loop
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // One K step of this wave: memory phase, barrier, compute phase, barrier. `u` (the tap) is a compile-time constant after unrolling, so the tap
    // offsets, the ring stages and the pixel-piece schedule fold away; `chunk` is the step's channel chunk, `st` its index.
    auto step = [&](int st, int chunk, int u, bool last) {
        const unsigned char* ws = wbase + (u % 3) * WTILE;  // st % 3 == u % 3 (nine taps = three turns of the ring)
        const int tapn = (u + 1) % 9;
        const int xbn = (chunk + (u == 8 ? 1 : 0)) & 1;     // pixel buffer of step st+1
        const bool xa = u < MAXXP && u < nxpw;              // a pixel piece of the next chunk goes out in this phase (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
        // ---- memory phase ----
#pragma unroll
        for (int i = 0; i < NR; ++i) frag(ws, bo_cur, 0, i, a0, b0);
#pragma unroll
        for (int i = 0; i < NR; ++i) frag(ws, bo_cur, 1, i, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) {
            if (u < MAXXP) {
                if (xa) dma_x(u, chunk + 1, (chunk + 1) & 1);
            }
            dma_w(st + 2, (u + 2) % 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        b_offsets(HZERO + grp * 2 * XHB + xbn * XHB, tapn, bo_nxt);
        __builtin_amdgcn_sched_barrier(0);
        // everything issued before this phase has landed: T(st+1) (both groups wait for their shares one barrier before anyone reads the tile) and the
        // pixel pieces of earlier phases; this phase's two weight pieces and its pixel piece stay in flight
        if (!(ABL & 1)) {
            if (xa) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
#ifdef CDET_PROFILING
        unsigned long long tq = 0;
        if (TL && g_pp_dbg != nullptr && tid == 0) {
            tq = __builtin_readcyclecounter();
            tp_mb += tq - tp_prev;
        }
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef CDET_PROFILING
        if (TL && g_pp_dbg != nullptr && tid == 0) {
            tq = __builtin_readcyclecounter();
            tp_m += tq - tp_prev;
            tp_prev = tq;
        }
#endif
#ifdef CDET_PROFILING
        if ((ABL & 64) && g_pp_dbg != nullptr && blockIdx.x == 0 && (tid == 0 || tid == 256) && st < 400) {
            // ABL 64: one s_memtime stamp per K step at the start of the compute phase (wave 0 of each group of workgroup 0 only), parked in LDS behind the
            // loop's buffers (an LDS store does not touch vmcnt) and copied out at the end
            const unsigned long long tq2 = __builtin_readcyclecounter();
            reinterpret_cast<unsigned long long*>(smem + HZERO + 4 * XHB + 3 * WTILE)[grp * 400 + st] = tq2;
        }
#endif
        // ---- compute phase ----
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NM; ++i) mfma32<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);
#pragma unroll
            for (int i = 0; i < NM; ++i) mfma32<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
        } else {  // keep the fragments alive (and their waits in place)
#pragma unroll
            for (int i = 0; i < NF; ++i) asm volatile("" ::"v"(a0[i]), "v"(a1[i]));
#pragma unroll
            for (int i = 0; i < NG; ++i) asm volatile("" ::"v"(b0[i]), "v"(b1[i]));
        }
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef CDET_PROFILING
        if (TL && g_pp_dbg != nullptr && tid == 0) {
            tq = __builtin_readcyclecounter();
            tp_c += tq - tp_prev;
            tp_prev = tq;
        }
#endif
#pragma unroll
        for (int g = 0; g < NG; ++g) bo_cur[g] = bo_nxt[g];
    };

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
#pragma unroll
        for (int u = 0; u < 9; ++u) step(chunk * 9 + u, chunk, u, u == 8 && chunk == a.nchunk - 1);
    }
    // group 0 releases group 1 into its last compute phase; then every wave drains its (dead) trailing DMA before anyone reuses the LDS
    if (grp == 0) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");  // asm MFMAs are opaque to the hazard recogniser
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef CDET_PROFILING
    if (g_pp_dbg != nullptr && tid == 0) tp_epi = __builtin_readcyclecounter();
    if ((ABL & 64) && g_pp_dbg != nullptr && blockIdx.x == 0) {  // the per-step stamps -> the record buffer behind the per-workgroup records
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + HZERO + 4 * XHB + 3 * WTILE);
        unsigned long long* dst = g_pp_dbg + (size_t)gridDim.x * 8;
        for (int i = tid; i < 800; i += 512) dst[i] = src[i];
        __syncthreads();
    }
#endif

    unsigned char* const esm = smem + grp * PP_EREG;  // the group's epilogue region (the loop's buffers are dead)
    // ---- BN statistics of the raw convolution (train mode): per (pixel block, channel) partial sums ---------------------------------------------
    if (a.stats != nullptr) {
        float* stl = reinterpret_cast<float*>(esm + HZERO);  // [4 waves][2][HC]
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
    }

    // ---- epilogue (conv_halo_kernel's): v_permlane32_swap pairs the two half waves' runs into 8 consecutive couts per lane; scale / bias / SiLU /
    //      residual on 16-byte vectors; rows staged through LDS and stored as whole NHWC pixels
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    float* const sbl = reinterpret_cast<float*>(esm + HZERO + 5120);  // [2][HC], behind the statistics scratch
    if (EPI != HEPI_RAW) {
        if (t < HC) {
            const int c = c0 + t < a.Cd ? c0 + t : a.Cd - 1;
            sbl[t] = a.scale ? a.scale[c] : 1.f;
            sbl[HC + t] = a.bias ? a.bias[c] : 0.f;
        }
        __syncthreads();
    }
    constexpr int RS = HC * 2 + 16;               // staging row stride (bytes): +16 spreads the 8-lane ds_write_b128 groups over all banks
    constexpr int CH = HC / 8;                    // 16-byte chunks per pixel row
    unsigned char* const stg = esm + HZERO + PP_STAGE_OFF + wave * (32 * RS);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int p = pout[g];
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
                u32x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = hpack2<DT>(v[2 * r], v[2 * r + 1]);
                *reinterpret_cast<u32x4*>(stg + l31 * RS + cl * 2) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // whole rows out: chunk id -> (pixel of this fragment, 16-byte chunk of its row)
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
            const int id = it * 64 + lane;
            const int px = id / CH, c = id - px * CH;
            if (id < 32 * CH) {
                const u32x4 pk = *reinterpret_cast<const u32x4*>(stg + px * RS + c * 16);
                const int i = wave * 64 + g * 32 + px;
                int po;
                if (PATCH) po = (pn * a.H + py0 + i / PATCH_W) * W + px0 + i % PATCH_W;
                else po = p0 + i;
                const int co = c0 + 8 * c;
                if (po < a.M && co < a.Cd) *reinterpret_cast<u32x4*>(yp + (int64_t)po * a.dst_ld + a.dst_coff + co) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // the next fragment overwrites the staging tile
    }
    if (a.stats != nullptr) {
        const float* stl = reinterpret_cast<const float*>(esm + HZERO);
        __syncthreads();
        if (live && t < HC && c0 + t < a.Cd) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sv += stl[(m * 2 + 0) * HC + t];
                qv += stl[(m * 2 + 1) * HC + t];
            }
            a.stats[((int64_t)pblk * 2 + 0) * a.Cd + c0 + t] = sv;
            a.stats[((int64_t)pblk * 2 + 1) * a.Cd + c0 + t] = qv;
        }
    }
#ifdef CDET_PROFILING
    if (g_pp_dbg != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = g_pp_dbg + (size_t)blockIdx.x * 8;
        o[0] = tp_start; o[1] = tp_loop; o[2] = tp_epi; o[3] = __builtin_readcyclecounter();
        if (TL) {
            o[4] = tp_mb; o[5] = tp_m; o[6] = tp_c;
        } else {
            o[4] = tr_start; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = 0;
        }
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
#endif
}

template <int DT, int EPI, bool PATCH>
static void launch_pp(const PpArgs& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_pp_kernel<DT, EPI, PATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
#ifdef CDET_PROFILING
    static int abl = -1;
    if (abl < 0) {
        const char* e = getenv("CDET_PP_ABLATE");
        abl = e ? atoi(e) : 0;
        if (abl & ~32) fprintf(stderr, "[cdet] CDET_PP_ABLATE=%d: conv results are WRONG by design (timing experiment)\n", abl);
    }
    if constexpr (DT == CDET_BF16 && EPI == HEPI_FULL && !PATCH) {
#define CDET_PPABL(N)                                                                                                                       \
    case N:                                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)conv_pp_kernel<DT, EPI, PATCH, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        hipLaunchKernelGGL((conv_pp_kernel<DT, EPI, PATCH, N>), dim3(nblocks), dim3(512), lds, s, a);                                        \
        return;
        switch (abl) {
            CDET_PPABL(1) CDET_PPABL(2) CDET_PPABL(3) CDET_PPABL(4) CDET_PPABL(5) CDET_PPABL(6) CDET_PPABL(7) CDET_PPABL(8) CDET_PPABL(32) CDET_PPABL(64)
            default: break;
        }
#undef CDET_PPABL
    }
#endif
    hipLaunchKernelGGL((conv_pp_kernel<DT, EPI, PATCH>), dim3(nblocks), dim3(512), lds, s, a);
}

// stride-1 3x3, 160-cout blocks, 256-pixel tiles (linear halo or 16 x 16 patches), 16-bit in == out. `XH` / `patch`: conv_halo.hip's halo_plan
bool pp_plan_ok(const cdet_conv_desc* d, int nf, int ng, bool patch, int XH) {
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1)) return false;
    if (nf != 5 || ng != 2 || d->out_dtype != d->dtype || d->accumulate) return false;
    if (XH > 16 * 4 * MAXXP) return false;
    const size_t loop = (size_t)HZERO + 4 * (size_t)XH * HROW + 3 * (size_t)(160 * HROW);
    return loop <= 160 * 1024;
}

int pp_launch(const cdet_conv_desc* d, bool patch, int XH, int n_pblk, const void* x, const void* w_tiled, const float* scale, const float* bias,
              const void* residual, void* y, float* stats, hipStream_t s) {
    PpArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_tiled; a.scale = scale; a.bias = bias; a.res = (const uint16_t*)residual;
    a.y = y; a.stats = stats;
    a.H = d->Hs; a.W = d->Ws; a.Cd = d->Cd;
    a.M = d->N * d->Hs * d->Ws;
    a.src_ld = d->src_ld; a.src_coff = d->src_coff; a.dst_ld = d->dst_ld; a.dst_coff = d->dst_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.nchunk = div_up(d->Cs, 32);
    a.Cs = d->Cs;
    a.n_pblk = n_pblk;
    a.n_cblk = div_up(d->Cd, 160);
    a.n_ppair = div_up(n_pblk, 2);
    a.act = d->act;
    a.XH = XH;
    a.tiles_x = d->Ws / PATCH_W;
    a.tiles_per_img = (d->Hs / 16) * (d->Ws / PATCH_W);
    a.x_bytes = (unsigned)((int64_t)a.M * d->src_ld * 2);
    a.w_bytes = (unsigned)((int64_t)a.n_cblk * a.nchunk * 9 * 160 * HROW);
    size_t lds = (size_t)HZERO + 4 * (size_t)XH * HROW + 3 * (size_t)(160 * HROW);
    if (lds < 2 * (size_t)PP_EREG) lds = 2 * (size_t)PP_EREG;
#ifdef CDET_PROFILING
    lds = (size_t)HZERO + 4 * (size_t)XH * HROW + 3 * (size_t)(160 * HROW) + 8192 > lds ? (size_t)HZERO + 4 * (size_t)XH * HROW + 3 * (size_t)(160 * HROW) + 8192 : lds;  // (ABL 64: per-step stamps)
#endif
    const bool full = scale || bias || residual || d->act != CDET_ACT_NONE;
    const int nblocks = a.n_ppair * a.n_cblk;
#define CDET_PP_GO(DT_)                                                                      \
    do {                                                                                     \
        if (patch) {                                                                         \
            if (full) launch_pp<DT_, HEPI_FULL, true>(a, lds, nblocks, s);                   \
            else launch_pp<DT_, HEPI_RAW, true>(a, lds, nblocks, s);                         \
        } else {                                                                             \
            if (full) launch_pp<DT_, HEPI_FULL, false>(a, lds, nblocks, s);                  \
            else launch_pp<DT_, HEPI_RAW, false>(a, lds, nblocks, s);                        \
        }                                                                                    \
    } while (0)
    if (d->dtype == CDET_BF16) CDET_PP_GO(CDET_BF16);
    else CDET_PP_GO(CDET_F16);
#undef CDET_PP_GO
    return 0;
}

}  // namespace cdet

#ifdef CDET_PROFILING
// profiling builds only (not part of the C-ABI): per-workgroup record buffer of 8 u64 per workgroup, or NULL to switch it off
extern "C" int cdet_debug_pp_timeline(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(cdet::g_pp_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
