// Eval-form fusion of the first two backbone rows (gfx950): the Cin = 3 stem  Conv(3, c1, 3, 2)  and the stride-2  Conv(c1, c2, 3, 2)
// behind it (models/common.py:57-68 via models/yolo.py:172-203, rows 0 and 1 of every YOLOv8 backbone), BatchNorm folded into scale / bias.
//
// Unfused, the stem writes its 320 x 320 x 80 map (524 MB at batch 32 @640, the largest tensor of the network) and the next row reads
// it back through four parity-plane gathers: 0.20 + 0.29 ms of a 12.5 ms forward for 0.9 % of its FLOPs, neither kernel near a roof
// (the stem's time is its SiLU epilogue, 262 M exponentials + reciprocals on the transcendental pipe; the stride-2 row moves 1.1 GB).
// Here a workgroup owns a 16 x 16 tile of the SECOND row's output (x all its <= 160 couts) and never lets the stem's map leave the CU:
//   * the 67 x 67 x 3 image patch under the tile is loaded once into LDS (aligned dwords of the NCHW image, uint8 / 16-bit / float);
//   * per 32-channel chunk of the stem's output and per parity plane P_pq(y', x') = S(2y' + p, 2x' + q) of it (conv_vt.hip's view of a
//     stride-2 input), the 17 x 17 stem outputs the plane's taps read are COMPUTED into the pixel buffer -- stem_mfma.hip's GEMM
//     (k = 4 (3c + kh) + kw, K = 48, B fragments = two runs of four consecutive image columns straight from the patch) and its
//     scale / bias / SiLU epilogue, written as the swizzled 64-byte rows the tap loop reads (positions outside the stem's map are the
//     second row's zero padding and are written as zeros);
//   * the plane's 4 / 2 / 2 / 1 tap steps then run as in conv_vt.hip: 10 KiB weight tiles through a 3-stage LDS-DMA ring,
//     v_mfma_f32_32x32x16, 160 fp32 accumulators per lane, one barrier per step, the same (chunk, plane, tap, k16) summation order --
//     so the result carries the same bits as the two-kernel path (tests/test_gpu_kernels.py);
//   * epilogue as conv_halo.hip / conv_vt.hip (scale, bias, SiLU, LDS-staged whole-row NHWC stores).
// The stem is recomputed on a 33 x 33 halo per 32 x 32 outputs it feeds (+6 %), its exponentials run under the other resident
// workgroup's MFMAs, and HBM sees the image and the 160 x 160 output only.
#include <stdlib.h>

#include "halo_common.h"

#include <type_traits>

namespace cdet {

struct Sc1Args {
    const void* img;          // NCHW
    const u32x4* ws;          // stem weights as MFMA A fragments: [chunk][k16 step 0..2][lane] (cdet_stem_conv1_pack)
    const float* sscale;      // stem folded-BN scale / bias [c1] (NULL: 1 / 0)
    const float* sbias;
    const uint16_t* w;        // second row: forward operand of cdet_pack_weights_tiled ([1 cout block][chunk][tap][RB][32])
    const float* scale;       // second row folded-BN scale / bias [c2]
    const float* bias;
    void* y;
    int img_dtype, N, H, W;   // image
    int Hs, Ws;               // stem output = H/2 x W/2
    int Ho, Wo;               // output = H/4 x W/4
    int c1, c2, nchunk, halfk;
    int dst_ld, dst_coff, act;
    int tiles_x, tiles_y;
    unsigned w_bytes;
    int abl;  // -DCDET_PROFILING timing experiments (CDET_SC1_ABLATE): 1 = no SiLU in the stem phase, 2 = no tap steps, 4 = no stem phases, 8 = no patch load
};

#ifdef CDET_PROFILING
#define SC1_ABL(bit) (a.abl & (bit))
#else
#define SC1_ABL(bit) 0
#endif

constexpr int SC_PR = 67;                       // image rows (and columns) under a 16 x 16 output tile: 4 * 16 + 3
constexpr int SC_PITCH = 136;                   // bytes per patch row: entries e = 0 .. 67 <-> image column 4 * ox0 - 3 + e (68 x 2 bytes)
constexpr int SC_PATCH = 3 * SC_PR * SC_PITCH;  // 27336
constexpr int SC_PATCH_PAD = (SC_PATCH + 255) / 256 * 256;
constexpr int SC_HPW = PATCH_W + 1;             // plane halo pitch 17 (conv_vt.hip patch mode)
constexpr int SC_NPOS = SC_HPW * SC_HPW;        // 289 plane positions
constexpr int SC_XROWS = 290;                   // rows of the plane buffer: 289 positions + a dump row for the lanes of a last, partial fragment
constexpr int SC_AF = 3 * 64 * 16;                // one chunk's stem A fragments in LDS
constexpr int SC_SB = 2 * 96 * 4;               // stem scale / bias in LDS
constexpr int SC_EPI_STAGE_OFF = 6912;

// LDS map: [zero row 256][stem scale/bias 768][patch][plane buffer 290 x 64][weight ring 3 x WTILE][dump 1 KiB][stem A fragments 3 KiB]
template <int DT, int NF, int A>
__global__ __launch_bounds__(256, 2) void stem_conv1_kernel(const Sc1Args a) {
    constexpr int NG = 2;
    constexpr int HC = NF * 32;
    constexpr int WTILE = HC * HROW;
    constexpr int WPC = WTILE / 1024;
    constexpr int NWP = (WPC + 3) / 4;
    constexpr int NM = NG * NF;
    constexpr int XOFF = HZERO + SC_SB + SC_PATCH_PAD;       // plane buffer
    constexpr int WOFF = XOFF + SC_XROWS * HROW;             // weight ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7, j = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tpi = a.tiles_x * a.tiles_y;
    const int pn = L / tpi;
    const int tr = L - pn * tpi;
    const int py0 = (tr / a.tiles_x) * PATCH_W, px0 = (tr % a.tiles_x) * PATCH_W;

    unsigned char* const patch = smem + HZERO + SC_SB;
    float* const ssb = reinterpret_cast<float*>(smem + HZERO);
    unsigned char* const xbuf = smem + XOFF;
    unsigned char* const wbase = smem + WOFF;
    unsigned char* const wdump = wbase + 3 * WTILE;
    u32x4* const afl = reinterpret_cast<u32x4*>(wdump + 1024);  // [k16 step][lane]: the current chunk's stem weights

    if (t < 16) reinterpret_cast<uint32_t*>(smem)[t] = 0u;  // zero row
    if (t < 96) {
        ssb[t] = t < a.c1 ? (a.sscale ? a.sscale[t] : 1.f) : 0.f;   // channels beyond c1: SiLU(0 * 0 + 0) = 0 fills the chunk's tail
        ssb[96 + t] = t < a.c1 ? (a.sbias ? a.sbias[t] : 0.f) : 0.f;
    }

    // ---- weight ring DMA (as conv_vt.hip: piece 4j + wave of the tile, surplus ids through an empty descriptor into the dump) -------
    auto dma_w1 = [&](int step, int stage, int j) __attribute__((always_inline)) {  // step = chunk * 9 + tap id in the packed order
        const int id = 4 * j + wave;
        const bool real = id < WPC;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (real && step < a.nchunk * 9) ? (int)a.w_bytes : 0, 0x00020000);
        unsigned char* dst = real ? wbase + stage * WTILE + id * 1024 : wdump;
        dma16(rs, (unsigned)((real ? id : 0) * 1024 + lane * 16), (unsigned)step * (unsigned)WTILE, dst);
    };
    // packed tap id (ky * 3 + kx) of step position u: P11 (0,0) (0,2) (2,0) (2,2) | P10 (0,1) (2,1) | P01 (1,0) (1,2) | P00 (1,1)
    auto tap_of = [](int u) -> int {
        constexpr int T[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
        return T[u];
    };
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int j = 0; j < NWP; ++j) dma_w1(tap_of(s_), s_, j);

    // ---- image patch -> LDS (16-bit entries, zeros outside the image). Every thread issues ALL its aligned dword loads before it
    //      converts the first one (one memory latency per tile instead of one per batch of eight: 10 us -> 4 us per workgroup)
    if (!SC1_ABL(8)) {
        constexpr int ND = (68 + A) / A;               // aligned dwords per patch row from column 4 * px0 - 4 on (entry e = -1 + d * A + j)
        constexpr int NITEM = 3 * SC_PR * ND;
        constexpr int NPT = (NITEM + 255) / 256;       // 15 (uint8) / 28 (16-bit) / 55 (float) dwords per thread
        constexpr int NB = NPT > 28 ? 28 : NPT;        // per batch (the float image takes two)
        constexpr int ES = A == 4 ? 1 : (A == 2 ? 2 : 4);
        const int iy0 = 4 * py0 - 3, col00 = 4 * px0 - 4;
        uint16_t* const pl = reinterpret_cast<uint16_t*>(patch);
        // (a 16-bit image has the compute type -- cdet_stem_conv1_ok -- and its entries are copied bit for bit)
        for (int base = 0; base < NITEM; base += 256 * NB) {
            uint32_t v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = base + i * 256 + t;
                v[i] = 0u;
                if (idx < NITEM) {
                    const int row = idx / ND, d = idx - row * ND;
                    const int c = row / SC_PR, r = row - c * SC_PR;
                    const int iy = iy0 + r, col0 = col00 + d * A;
                    if ((unsigned)iy < (unsigned)a.H && col0 >= 0 && col0 < a.W) {
                        const int64_t e0 = (((int64_t)pn * 3 + c) * a.H + iy) * a.W + col0;
                        v[i] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(a.img) + e0 * ES);  // W % A == 0 (host check)
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = base + i * 256 + t;
                if (idx < NITEM) {
                    const int row = idx / ND, d = idx - row * ND;
#pragma unroll
                    for (int e = 0; e < A; ++e) {
                        const int cp = -1 + d * A + e;
                        uint16_t bits;
                        if (A == 2) bits = (uint16_t)(v[i] >> (16 * e));
                        else if (A == 4) bits = Elem<DT>::from_f32((float)((v[i] >> (8 * e)) & 0xffu) * (1.0f / 255.0f));
                        else bits = Elem<DT>::from_f32(__uint_as_float(v[i]));
                        if (cp >= 0 && cp < 68) pl[row * (SC_PITCH / 2) + cp] = bits;
                    }
                }
            }
        }
    }

    // ---- tap-loop fragment offsets (conv_vt.hip, forward, patch mode) -----------------------------------------------------------------
    const int aoff0 = l31 * HROW + ((h ^ ((l31 >> 2) & 3)) << 4);
    int pixh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = wave * (32 * NG) + g * 32 + l31;
        pixh[g] = (i / PATCH_W + 1) * SC_HPW + i % PATCH_W + 1;
    }

    f32x16 acc[NF][NG];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][g][r] = 0.f;

    // ---- stem plane: the stem outputs S(2 (py0 + hy - 1) + p, 2 (px0 + hx - 1) + q) the plane's taps read -- hy from 1 - p, hx from 1 - q
    //      (the p = 0 / q = 0 planes have no row / column -1) up to 16 -- channels of `chunk`, into row hy * 17 + hx of the plane buffer.
    //      P11 289 positions = 10 fragments of 32, P10 / P01 272 = 9, P00 256 = 8: wave w takes fragments w, w + 4, w + 8.
    //      af: the chunk's three A fragments (loaded once per chunk). half: only the first 16 channels of the chunk exist (c1 % 32 <= 16).
    auto stem_plane = [&](int chunk, int p, int q, auto HK) __attribute__((always_inline)) {
        constexpr bool half = decltype(HK)::value;
        if (SC1_ABL(4)) return;
        u32x4 af[3];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) af[s_] = afl[s_ * 64 + lane];
        const int CN = 16 + q, RN = 16 + p;            // columns / rows of the plane that are read
        const int npos = CN * RN, nfr = (npos + 31) / 32;
        const int poff = (2 * p - 2) * SC_PITCH + (2 * q - 2) * 2;  // plane offset inside the patch (rows, 2-byte entries)
        const float* const sc = ssb + chunk * 32;
        const float* const bi = ssb + 96 + chunk * 32;
        // opaque copies: every position-dependent value below is invariant across the chunk loop, and hoisted out of it they cost ~50 VGPRs
        // (spilled to scratch and reloaded inside the phases); recomputing them per phase is a dozen integer instructions per fragment
        int l31o = l31, ho = h;
        asm volatile("" : "+v"(l31o), "+v"(ho));
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
            const int fr = wave + 4 * rd;              // wave-uniform
            if (fr < nfr) {
                const int j = fr * 32 + l31o;
                const int jy = j / CN, jx = j - jy * CN;
                const int hy = jy + 1 - p, hx = jx + 1 - q;
                const bool live = j < npos;
                const int base = live ? 4 * hy * SC_PITCH + 8 * hx + poff : 0;
                f32x16 sa;
#pragma unroll
                for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    // k16 step s, run jr reads patch row R(r), r = min(4s + 2h + jr, 8), R(r) = (r / 3) * 67 + r % 3
                    const int r0 = 4 * s_, r1 = 4 * s_ + 2;
                    const int ra0 = r0 > 8 ? 8 : r0, ra1 = r1 > 8 ? 8 : r1, rb0 = r0 + 1 > 8 ? 8 : r0 + 1, rb1 = r1 + 1 > 8 ? 8 : r1 + 1;
                    const int o0 = (ho ? (ra1 / 3) * SC_PR + ra1 % 3 : (ra0 / 3) * SC_PR + ra0 % 3) * SC_PITCH;
                    const int o1 = (ho ? (rb1 / 3) * SC_PR + rb1 % 3 : (rb0 / 3) * SC_PR + rb0 % 3) * SC_PITCH;
                    const uint32_t* p0 = reinterpret_cast<const uint32_t*>(patch + base + o0);
                    const uint32_t* p1 = reinterpret_cast<const uint32_t*>(patch + base + o1);
                    const u32x4 bf = u32x4{p0[0], p0[1], p1[0], p1[1]};
                    // (the builtin, not the asm wrapper: the compiler then places the wait states between the MFMAs and the VALU epilogue below)
                    if (DT == CDET_BF16) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s_]), __builtin_bit_cast(bf16x8, bf), sa, 0, 0, 0);
                    else sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[s_]), __builtin_bit_cast(f16x8, bf), sa, 0, 0, 0);
                }
                const int Y = 2 * (py0 + hy - 1) + p, X = 2 * (px0 + hx - 1) + q;
                const bool inside = live && (unsigned)Y < (unsigned)a.Hs && (unsigned)X < (unsigned)a.Ws;
                const int i = live ? hy * SC_HPW + hx : SC_NPOS;  // plane-buffer row (dead lanes of a partial fragment: the dump row)
#pragma unroll
                for (int qq = 0; qq < 4; qq += 2) {
                    if (half && qq == 2) continue;     // channels 16 .. 31 of the last chunk do not exist: the tap steps never read them
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float lo = sa[4 * qq + r], hi = sa[4 * qq + 4 + r];
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                        const unsigned s0 = sw[0], s1 = sw[1];
                        v[r] = __uint_as_float(s0);
                        v[4 + r] = __uint_as_float(s1);
                    }
                    const int cl = 8 * (qq + ho);
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + cl), s1 = *reinterpret_cast<const f32x4*>(sc + cl + 4);
                    const f32x4 b0v = *reinterpret_cast<const f32x4*>(bi + cl), b1v = *reinterpret_cast<const f32x4*>(bi + cl + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = v[r] * s0[r] + b0v[r];
                        v[4 + r] = v[4 + r] * s1[r] + b1v[r];
                    }
                    if (a.act == CDET_ACT_SILU && !SC1_ABL(1)) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                    }
                    u32x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = inside ? hpack2<DT>(v[2 * r], v[2 * r + 1]) : 0u;
                    *reinterpret_cast<u32x4*>(xbuf + i * HROW + (((qq + ho) ^ ((i >> 2) & 3)) << 4)) = pk;
                }
            }
        }
    };

    // B-fragment byte offsets (relative to smem) of the tap (ky, kx)
    auto b_offsets = [&](int ky, int kx, int (&bo)[NG]) __attribute__((always_inline)) {
        const int d = (ky == 0 ? SC_HPW : 0) + (kx == 0 ? 1 : 0);
        // opaque copies (conv_vt.hip): otherwise the nine taps' offset pairs are hoisted out of the chunk loop as invariants -- 18 VGPRs
        // the 256-register budget does not have (they spilled to scratch)
        int p_[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            p_[g] = pixh[g];
            asm volatile("" : "+v"(p_[g]));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int hrow = p_[g] - d;
            bo[g] = XOFF + hrow * HROW + ((h ^ ((hrow >> 2) & 3)) << 4);
        }
    };
    auto frag_b = [&](const int (&bo)[NG], int s_, int g) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (bo[g] ^ (s_ << 5))); };
    auto frag_a = [&](const unsigned char* ws_, int s_, int f) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(ws_ + ((aoff0 ^ (s_ << 5)) + f * 32 * HROW));
    };

    // One tap step: st = chunk * 9 + u (ring stage u % 3), two-phase as in conv_vt.hip. Tile st has landed and is visible when the step
    // starts (the mid-step barrier of the previous step / the barrier behind the stem phase).
    //   phase A: MFMAs of k16 #0 with the fragment reads of k16 #1 in their shadow; counted wait (tile st + 1, issued two steps ago, has
    //            landed) + barrier: nobody reads stage u % 3 or -- at a plane's last step -- the plane buffer any more;
    //   phase B: DMA of tile st + 3 into the freed stage, MFMAs of k16 #1, and inside a plane the k16 #0 fragments of step st + 1.
    // first / last: the step opens / closes its parity plane (the plane buffer is rewritten between planes, so the fragment
    // prefetch stops at the boundary and the first step reads its own).
    u32x4 a0[NF], b0[NG], a1[NF], b1[NG];
    int bo_cur[NG], bo_nxt[NG];
    auto run_step = [&](int chunk, int u, auto HK, bool first, bool last) __attribute__((always_inline)) {
        constexpr bool half = decltype(HK)::value;
        const int stage = u % 3, stn = (u + 1) % 3;
        const unsigned char* ws = wbase + stage * WTILE;
        const unsigned char* wsn = wbase + stn * WTILE;
        constexpr int KY[9] = {0, 0, 2, 2, 0, 2, 1, 1, 1}, KX[9] = {0, 2, 0, 2, 1, 1, 0, 2, 1};
        if (first) {
            b_offsets(KY[u], KX[u], bo_cur);
#pragma unroll
            for (int g = 0; g < NG; ++g) b0[g] = frag_b(bo_cur, 0, g);
#pragma unroll
            for (int f = 0; f < NF; ++f) a0[f] = frag_a(ws, 0, f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (!SC1_ABL(2)) mfma32<DT>(a0[i / NG], b0[i % NG], acc[i / NG][i % NG]);
            if (!half) {
                if (i < NG) b1[i] = frag_b(bo_cur, 1, i);
                else if (i < NG + NF) a1[i - NG] = frag_a(ws, 1, i - NG);
            }
            if (i == NM - 1 && !last) b_offsets(KY[(u + 1) % 9], KX[(u + 1) % 9], bo_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm_lgkm0<NWP>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            const int u3 = (u + 3) % 9, c3 = chunk + (u + 3) / 9;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (!half && !SC1_ABL(2)) mfma32<DT>(a1[i / NG], b1[i % NG], acc[i / NG][i % NG]);
                if (i < NWP) dma_w1(c3 * 9 + tap_of(u3), stage, i);
                if (!last) {
                    if (i >= 1 && i - 1 < NG) b0[i - 1] = frag_b(bo_nxt, 0, i - 1);
                    else if (i >= 1 && i - 1 < NG + NF) a0[i - 1 - NG] = frag_a(wsn, 0, i - 1 - NG);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!last) {
#pragma unroll
            for (int g = 0; g < NG; ++g) bo_cur[g] = bo_nxt[g];
        }
    };

    // one 32-channel chunk of the stem's output: its A fragments into LDS, then per parity plane the stem phase and the plane's tap steps
    auto run_chunk = [&](int chunk, auto HK) __attribute__((always_inline)) {
        if (t < 192) afl[t] = a.ws[chunk * 192 + t];  // (the previous chunk's last read of this area lies four barriers back)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // P11: taps (0,0) (0,2) (2,0) (2,2)
        stem_plane(chunk, 1, 1, HK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        run_step(chunk, 0, HK, true, false);
        run_step(chunk, 1, HK, false, false);
        run_step(chunk, 2, HK, false, false);
        run_step(chunk, 3, HK, false, true);
        // P10: taps (0,1) (2,1)
        stem_plane(chunk, 1, 0, HK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        run_step(chunk, 4, HK, true, false);
        run_step(chunk, 5, HK, false, true);
        // P01: taps (1,0) (1,2)
        stem_plane(chunk, 0, 1, HK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        run_step(chunk, 6, HK, true, false);
        run_step(chunk, 7, HK, false, true);
        // P00: tap (1,1)
        stem_plane(chunk, 0, 0, HK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        run_step(chunk, 8, HK, true, true);
    };
    // (the barrier inside run_chunk also publishes the patch, the zero row and the scale / bias table)
    const int nfull = a.halfk ? a.nchunk - 1 : a.nchunk;
    for (int chunk = 0; chunk < nfull; ++chunk) run_chunk(chunk, std::false_type{});
    if (a.halfk) run_chunk(nfull, std::true_type{});  // the last chunk holds at most 16 channels: its second k16 half does not exist
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (conv_vt.hip): 8 consecutive couts per lane, scale / bias / SiLU, LDS-staged whole-row stores -----------------------------
    uint16_t* const yp = reinterpret_cast<uint16_t*>(a.y);
    float* const sbl = reinterpret_cast<float*>(smem + HZERO + 5120);
    if (t < HC) {
        const int c = t < a.c2 ? t : a.c2 - 1;
        sbl[t] = a.scale ? a.scale[c] : 1.f;
        sbl[HC + t] = a.bias ? a.bias[c] : 0.f;
    }
    __syncthreads();
    constexpr int RS = HC * 2 + 16;
    constexpr int CH = HC / 8;
    unsigned char* const stg = smem + HZERO + SC_EPI_STAGE_OFF + wave * (32 * RS);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
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
                const int i = wave * (32 * NG) + g * 32 + px;
                const int oy = py0 + i / PATCH_W, ox = px0 + i % PATCH_W;
                if (oy < a.Ho && ox < a.Wo && 8 * c < a.c2)
                    *reinterpret_cast<u32x4*>(yp + (((int64_t)pn * a.Ho + oy) * a.Wo + ox) * a.dst_ld + a.dst_coff + 8 * c) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// stem weights fp32 OIHW [c1, 3, 3, 3] -> MFMA A fragments [chunk][k16 step][lane] (row = chunk * 32 + (lane & 31), k = 16 s + 8 (lane >> 5) + j
// <-> (3c + kh) = k / 4, kw = k % 4; kw = 3, k / 4 >= 9 and rows >= c1 are zero): stem_mfma.hip's operand, one chunk per 32 couts
__global__ void stem_conv1_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out, int c1, int dtype) {
    const int lane = threadIdx.x & 63, s = (threadIdx.x >> 6) % 3, chunk = blockIdx.x;
    const int co = chunk * 32 + (lane & 31), h = lane >> 5;
    uint16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = s * 16 + h * 8 + j;
        const int r = k >> 2, kw = k & 3;
        const float f = (co < c1 && r < 9 && kw < 3) ? w[co * 27 + r * 3 + kw] : 0.f;
        v[j] = dtype == CDET_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
    }
    out[(chunk * 3 + s) * 64 + lane] = u32x4{(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                                             (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
}

template <int DT, int NF, int A>
static void launch_sc1(const Sc1Args& a, size_t lds, int nblocks, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)stem_conv1_kernel<DT, NF, A>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((stem_conv1_kernel<DT, NF, A>), dim3(nblocks), dim3(256), lds, s, a);
}

template <int DT, int NF>
static void dispatch_sc1(const Sc1Args& a, int per, size_t lds, int nblocks, hipStream_t s) {
    if (per == 4) launch_sc1<DT, NF, 4>(a, lds, nblocks, s);
    else if (per == 2) launch_sc1<DT, NF, 2>(a, lds, nblocks, s);
    else launch_sc1<DT, NF, 1>(a, lds, nblocks, s);
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_stem_conv1_ok(int32_t N, int32_t H, int32_t W, int32_t c1, int32_t c2, int32_t img_dtype, int32_t dtype, int32_t dst_ld,
                                  int32_t dst_coff) {
    if (N <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 4 != 0) return 0;
    if (c1 % 8 != 0 || c1 < 8 || c1 > 96 || c2 % 8 != 0 || c2 < 8 || c2 > 160) return 0;
    if (!(dtype == CDET_BF16 || dtype == CDET_F16)) return 0;
    if (!(img_dtype == CDET_U8 || img_dtype == CDET_F32 || img_dtype == dtype)) return 0;  // a 16-bit image must have the compute type
    if (dst_ld % 8 != 0 || dst_coff % 8 != 0) return 0;
    if ((int64_t)N * (H / 4) * (W / 4) * dst_ld >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int64_t cdet_stem_conv1_pack_elems(int32_t c1) { return (int64_t)div_up(c1, 32) * 3 * 64 * 8; }  // 16-bit elements

extern "C" int cdet_stem_conv1_pack(const float* w_stem, void* out, int32_t c1, int32_t dtype, void* stream) {
    CDET_CHECK_ARG(w_stem && out && c1 > 0 && c1 <= 96, "cdet_stem_conv1_pack: bad arguments");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_stem_conv1_pack: dtype must be bf16/f16");
    hipLaunchKernelGGL(stem_conv1_pack_kernel, dim3(div_up(c1, 32)), dim3(192), 0, (hipStream_t)stream, w_stem, (u32x4*)out, c1, dtype);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_stem_conv1(const void* img, int32_t img_dtype, const void* w_stem_packed, const float* stem_scale, const float* stem_bias,
                               const void* w1_tiled, const float* scale, const float* bias, void* y, int32_t N, int32_t H, int32_t W, int32_t c1,
                               int32_t c2, int32_t dtype, int32_t dst_ld, int32_t dst_coff, int32_t act, void* stream) {
    CDET_CHECK_ARG(img && w_stem_packed && w1_tiled && y, "cdet_stem_conv1: null pointer");
    CDET_CHECK_ARG(cdet_stem_conv1_ok(N, H, W, c1, c2, img_dtype, dtype, dst_ld, dst_coff),
                   "cdet_stem_conv1: unsupported geometry (H, W multiples of 4; stem couts <= 96, second-row couts <= 160, both multiples of 8; "
                   "16-bit activations; image uint8, float or of the activation type)");
    const int per = img_dtype == CDET_U8 ? 4 : (img_dtype == CDET_F32 ? 1 : 2);
    CDET_CHECK_ARG(W % per == 0, "cdet_stem_conv1: image rows must be whole dwords");
    Sc1Args a;
    a.img = img; a.ws = (const u32x4*)w_stem_packed; a.sscale = stem_scale; a.sbias = stem_bias;
    a.w = (const uint16_t*)w1_tiled; a.scale = scale; a.bias = bias; a.y = y;
    a.img_dtype = img_dtype; a.N = N; a.H = H; a.W = W;
    a.Hs = H / 2; a.Ws = W / 2; a.Ho = H / 4; a.Wo = W / 4;
    a.c1 = c1; a.c2 = c2;
    a.nchunk = div_up(c1, 32);
    a.halfk = (c1 % 32 != 0 && c1 % 32 <= 16) ? 1 : 0;
    a.dst_ld = dst_ld; a.dst_coff = dst_coff; a.act = act;
    a.tiles_x = div_up(a.Wo, PATCH_W);
    a.tiles_y = div_up(a.Ho, PATCH_W);
    const int nf = c2 <= 96 ? 3 : 5;
    const int rb = nf * 32;
    a.w_bytes = (unsigned)((int64_t)a.nchunk * 9 * rb * HROW);
    a.abl = tune_env("CDET_SC1_ABLATE", 0);
    size_t lds = (size_t)HZERO + SC_SB + SC_PATCH_PAD + (size_t)SC_XROWS * HROW + 3 * (size_t)rb * HROW + 1024 + SC_AF;
    const size_t epi = (size_t)HZERO + SC_EPI_STAGE_OFF + 4 * 32 * (size_t)(rb * 2 + 16);
    if (lds < epi) lds = epi;
    const int nblocks = N * a.tiles_x * a.tiles_y;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CDET_BF16) {
        if (nf == 5) dispatch_sc1<CDET_BF16, 5>(a, per, lds, nblocks, s);
        else dispatch_sc1<CDET_BF16, 3>(a, per, lds, nblocks, s);
    } else {
        if (nf == 5) dispatch_sc1<CDET_F16, 5>(a, per, lds, nblocks, s);
        else dispatch_sc1<CDET_F16, 3>(a, per, lds, nblocks, s);
    }
    CDET_LAUNCH_CHECK();
    return 0;
}
