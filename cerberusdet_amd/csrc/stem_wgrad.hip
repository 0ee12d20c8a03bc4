// Weight gradient of the Cin = 3 stem (3x3, stride 2, pad 1) straight from the NCHW image (gfx950, 16-bit dz, fp32 accumulate):
//
//   dW[co][c][kh][kw] = sum_{n,oh,ow} dz[n,oh,ow,co] * img[n, c, 2*oh + kh - 1, 2*ow + kw - 1] (* 1/255 for uint8 images)
//                                                               (autograd's convolution_backward(weight) of the first Conv, models/common.py:57)
//
// A GEMM with a tiny output (Cout x 27) and a huge reduction (N*Ho*Wo pixels): it is bound by reading dz once (524 MB at batch 32 @640).
// Round 2 ran it on the generic im2col kernel over an NHWC8 copy of the image (K = 72 padded to 128: 0.23 ms + 0.08 ms for the copy).
// Here a workgroup (4 waves, 4 per CU -- 36 KB of LDS each: the phases of a tile are serial inside a workgroup, the others' DMA is what
// keeps HBM busy meanwhile; persistent) walks tiles of 4 x 32 output pixels:
//   * dz tile [128 px][Cout <= 80] by LDS-DMA into the transpose-read layout of conv_wgrad_halo.hip (1 KiB images
//     [16-pixel block][cout pair][2 cout groups][16 px][32 B]);
//   * the image patch (3 x 9 x 65 values) is decoded once into LDS (x 1/255, rounded to the compute dtype exactly like the forward's
//     operand), two threads then write the im2col row of one output pixel: 27 values + 5 zeros = two 32-byte plane entries
//     (k = (c*3 + kh)*3 + kw, the OIHW order);
//   * per tile a wave issues 2 x 5 v_mfma_f32_16x16x32 for its 32 pixels (im2col planes x cout groups), operands by ds_read_b64_tr_b16.
// Every workgroup leaves one fp32 partial [Cout][32]; stem_wgrad_finish_kernel sums them in a fixed order (deterministic, no atomics).
#include "common.h"
#include "halo_common.h"
#include "wgrad_tr.h"

namespace cdet {

struct StemWgArgs {
    const void* img;
    const uint16_t* dy;
    float* ws;
    int img_dtype;
    int N, H, W, Ho, Wo, Cout, dy_ld;
    int tx, ty, NT;  // tiles per image row / column, tiles in all
    unsigned dy_bytes;
};

constexpr int SG_TR = 4;                          // output rows of a tile (x 32 columns)
constexpr int SG_NPB = SG_TR * 2;                 // 16-pixel blocks of a tile
constexpr int SG_DZ = SG_NPB * 3 * 1024;          // pixel blocks x 3 cout pairs x 1 KiB
constexpr int SG_IMP = SG_NPB * 512;              // one im2col plane: [pixel block][16 px][32 B]
constexpr int SG_IM = 2 * SG_IMP;
constexpr int SG_PR = 2 * SG_TR + 1;              // patch rows
constexpr int SG_PW = 66;                         // patch row pitch (65 columns)
constexpr int SG_PATCH = 3 * SG_PR * SG_PW;       // 16-bit values
constexpr int SG_TILE_LDS = SG_DZ + SG_IM + (SG_PATCH * 2 + 15) / 16 * 16;
constexpr int SG_LDS = SG_TILE_LDS > 4 * 10 * 1024 ? SG_TILE_LDS : 4 * 10 * 1024;  // the end-of-kernel wave sum (4 x 10 KiB) reuses the tile memory
constexpr int SG_WS = 96 * 32;                    // floats per workgroup partial
static_assert(4 * SG_LDS <= 160 * 1024, "four workgroups per CU");

template <int DT>
__global__ __launch_bounds__(256, 4) void stem_wgrad_mfma_kernel(const StemWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* patch = reinterpret_cast<uint16_t*>(smem + SG_DZ + SG_IM);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const int ldyB = a.dy_ld * 2;
    // DMA lane -> (cout group lane>>5, pixel (lane>>1)&15 of the block, 16-byte half lane&1)
    const int yco_l = (lane >> 5) * 16 + (lane & 1) * 8;
    const int ypx = (lane >> 1) & 15;
    // im2col: threads t and t + 128 write the two 32-byte plane entries (k 0..15, k 16..31) of output pixel t & 127 of the tile
    const int pix = t & 127, plane = t >> 7;
    const int pr = pix >> 5, pc = pix & 31;
    unsigned char* im_dst = smem + SG_DZ + plane * SG_IMP + ((pr * 2 + (pc >> 4)) * 16 + (pc & 15)) * 32;
    const uint16_t* pbase = patch + (2 * pr) * SG_PW + 2 * pc;

    f32x4 acc[2][5];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int vy = wave * 2 * 3072 + lane * 8;              // dz images of pixel blocks 2*wave, 2*wave + 1
    const int vx = SG_DZ + wave * 2 * 512 + lane * 8;       // im2col plane 0, the same pixel blocks

    const int per = a.ty * a.tx;
    for (int tile = blockIdx.x; tile < a.NT; tile += gridDim.x) {
        const int n = tile / per, r0 = tile - n * per;
        const int tyi = r0 / a.tx, txi = r0 - tyi * a.tx;
        const int oh0 = tyi * SG_TR, ow0 = txi * 32;
        // ---- dz tile: 6 DMA instructions per wave
#pragma unroll
        for (int i = 0; i < SG_NPB * 3 / 4; ++i) {
            const int di = wave * (SG_NPB * 3 / 4) + i;  // wave-uniform: (pixel block di/3, cout pair di%3)
            const int pb = di / 3, cp = di - pb * 3;
            const int oh = oh0 + (pb >> 1), ow = ow0 + (pb & 1) * 16 + ypx;
            const bool ok = oh < Ho && ow < Wo && cp * 32 + yco_l < a.Cout;
            const unsigned v = (unsigned)(((n * Ho + oh) * Wo + ow) * ldyB + (cp * 32 + yco_l) * 2);
            wh_dma16(rs_y, ok ? v : WH_SENT, smem + di * 1024);
        }
        // ---- image patch -> LDS (rows 2*oh0-1 .. +2*TR, columns 2*ow0-1 .. +64), zero outside the image
        if (a.img_dtype == CDET_U8 && (W & 3) == 0) {
            // uint8 rows are read as the 17 aligned dwords [2*ow0 - 4, 2*ow0 + 64): byte b of dword d is patch column 4*d + b - 3
            for (int i = t; i < 3 * SG_PR * 17; i += 256) {
                const int row = i / 17, d = i - row * 17;  // row = c * SG_PR + rr
                const int c = row / SG_PR, rr = row - c * SG_PR;
                const int iy = 2 * oh0 - 1 + rr, ix0 = 2 * ow0 - 4 + 4 * d;
                uint32_t q = 0;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix0 < (unsigned)W)  // (ix0 = -4 for the first dword of a left-border tile)
                    q = *reinterpret_cast<const uint32_t*>((const uint8_t*)a.img + (((int64_t)n * 3 + c) * H + iy) * W + ix0);
                const uint32_t p01 = hpack2<DT>((float)(q & 0xffu) * (1.0f / 255.0f), (float)((q >> 8) & 0xffu) * (1.0f / 255.0f));
                const uint32_t p23 = hpack2<DT>((float)((q >> 16) & 0xffu) * (1.0f / 255.0f), (float)(q >> 24) * (1.0f / 255.0f));
                uint16_t* dst = patch + row * SG_PW + 4 * d - 3;
                if (d > 0) {
                    dst[0] = (uint16_t)p01;
                    dst[1] = (uint16_t)(p01 >> 16);
                    dst[2] = (uint16_t)p23;
                }
                dst[3] = (uint16_t)(p23 >> 16);
            }
        } else {
            for (int i = t; i < 3 * SG_PR * 65; i += 256) {
                const int c = i / (SG_PR * 65), rem = i - c * (SG_PR * 65);
                const int rr = rem / 65, cc = rem - rr * 65;
                const int iy = 2 * oh0 - 1 + rr, ix = 2 * ow0 - 1 + cc;
                float v = 0.f;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int64_t gi = (((int64_t)n * 3 + c) * H + iy) * W + ix;
                    v = a.img_dtype == CDET_U8 ? (float)((const uint8_t*)a.img)[gi] * (1.0f / 255.0f) : load_elem(a.img, gi, a.img_dtype);
                }
                patch[(c * SG_PR + rr) * SG_PW + cc] = Elem<DT>::from_f32(v);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- this thread's half of an im2col row: k = (c*3 + kh)*3 + kw, 16 values
        {
            uint32_t w[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {  // (both planes' offsets are compile-time; the plane picks one)
                    const int k0 = 16 * pl + 2 * k2, k1 = k0 + 1;
                    if (pl == plane) {
                        if (k0 < 27) lo = pbase[((k0 / 9) * SG_PR + (k0 % 9) / 3) * SG_PW + k0 % 3];
                        if (k1 < 27) hi = pbase[((k1 / 9) * SG_PR + (k1 % 9) / 3) * SG_PW + k1 % 3];
                    }
                }
                w[k2] = lo | (hi << 16);
            }
            *reinterpret_cast<u32x4*>(im_dst) = u32x4{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<u32x4*>(im_dst + 16) = u32x4{w[4], w[5], w[6], w[7]};
        }
        __syncthreads();
        // ---- one reduction step of 32 pixels per wave: pixel blocks 2*wave, 2*wave + 1
        {
            u32x2 blo[5], bhi[5], alo[2], ahi[2];
            wh_static_for(std::make_integer_sequence<int, 5>{}, [&](auto J) {
                constexpr int j = decltype(J)::value;
                blo[j] = wh_tr<j * 512>(vy);
                bhi[j] = wh_tr<3072 + j * 512>(vy);
            });
            alo[0] = wh_tr<0>(vx);
            ahi[0] = wh_tr<512>(vx);
            alo[1] = wh_tr<SG_IMP>(vx);
            ahi[1] = wh_tr<SG_IMP + 512>(vx);
            wh_wait_b<0>(blo, bhi, alo[0], ahi[0]);
            wh_wait<0>(alo[1], ahi[1]);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const u32x4 av{alo[p][0], alo[p][1], ahi[p][0], ahi[p][1]};
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const u32x4 bv{blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                    wh_mfma<DT>(av, bv, acc[p][j]);
                }
            }
        }
        __syncthreads();  // all reads of this tile's LDS are done before the next tile's fill
    }

    // ---- sum the four waves' tiles through LDS; partial[co][k] of this workgroup (C[k][co] tiles: lane (q, li) holds k = 16p + 4q + r, co = 16j + li)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4*>(smem + ((wave * 10 + p * 5 + j) * 64 + lane) * 16) = acc[p][j];
    __syncthreads();
    const float* sf = reinterpret_cast<const float*>(smem);
    float* wsp = a.ws + (int64_t)blockIdx.x * SG_WS;
    for (int e = t; e < 10 * 256; e += 256) {
        const int tl = e >> 8, le = e & 255;
        const float s = (sf[(0 * 10 + tl) * 256 + le] + sf[(1 * 10 + tl) * 256 + le]) + (sf[(2 * 10 + tl) * 256 + le] + sf[(3 * 10 + tl) * 256 + le]);
        const int p = tl / 5, j = tl - p * 5;
        const int ln = le >> 2, r = le & 3;
        const int k = 16 * p + 4 * (ln >> 4) + r, co = 16 * j + (ln & 15);
        wsp[co * 32 + k] = s;
    }
}

// dw[co][k] (+)= sum over workgroups, in a fixed order: 16 threads per output, thread j sums the partials g = j, j + 16, ... (two interleaved chains), the 16
// sums are added up in lane order through LDS. (One thread per output walked the 1024 partials alone: 88 us on the tail of every backward pass.)
__global__ __launch_bounds__(256) void stem_wgrad_finish_kernel(const float* __restrict__ ws, int nwg, int Cout, float* __restrict__ dw, int accumulate) {
    __shared__ float sh[16][17];
    const int o = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + o;
    float s0 = 0.f, s1 = 0.f;
    if (i < Cout * 27) {
        const int co = i / 27, k = i - co * 27;
        const float* p = ws + co * 32 + k;
        int g = j;
        for (; g + 16 < nwg; g += 32) {
            s0 += p[(int64_t)g * SG_WS];
            s1 += p[(int64_t)(g + 16) * SG_WS];
        }
        if (g < nwg) s0 += p[(int64_t)g * SG_WS];
    }
    sh[o][j] = s0 + s1;
    __syncthreads();
    if (j == 0 && i < Cout * 27) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) s += sh[o][m];
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

static int stem_wgrad_grid(int N, int H, int W) {
    const int NT = N * div_up(H / 2, SG_TR) * div_up(W / 2, 32);
    return NT < 1024 ? NT : 1024;
}

}  // namespace cdet

using namespace cdet;

extern "C" int64_t cdet_stem_conv_wgrad_ws_elems(int32_t N, int32_t H, int32_t W) {
    if (N < 1 || H < 2 || W < 2) return -1;
    return (int64_t)stem_wgrad_grid(N, H, W) * SG_WS;
}

extern "C" int cdet_stem_conv_wgrad(const void* img, int32_t img_dtype, const void* dy, int32_t dy_ld, int32_t dtype, float* dw, int32_t N, int32_t H,
                                    int32_t W, int32_t Cout, int32_t accumulate, float* ws, void* stream) {
    CDET_CHECK_ARG(img && dy && dw && ws, "cdet_stem_conv_wgrad: null pointer");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_stem_conv_wgrad: dy must be bf16/f16");
    CDET_CHECK_ARG(img_dtype == CDET_U8 || img_dtype == CDET_F32 || img_dtype == CDET_BF16 || img_dtype == CDET_F16, "cdet_stem_conv_wgrad: image dtype");
    CDET_CHECK_ARG(Cout >= 8 && Cout <= 80 && Cout % 8 == 0 && dy_ld >= Cout && dy_ld % 8 == 0, "cdet_stem_conv_wgrad: Cout must be a multiple of 8 up to 80 (got %d)", Cout);
    CDET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && N >= 1, "cdet_stem_conv_wgrad: H and W must be even");
    const int64_t dyb = (int64_t)N * (H / 2) * (W / 2) * dy_ld * 2;
    CDET_CHECK_ARG(dyb < 0xC0000000ll, "cdet_stem_conv_wgrad: dy too large for 32-bit DMA offsets");
    StemWgArgs a;
    a.img = img; a.dy = (const uint16_t*)dy; a.ws = ws; a.img_dtype = img_dtype;
    a.N = N; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Cout = Cout; a.dy_ld = dy_ld;
    a.ty = div_up(a.Ho, SG_TR); a.tx = div_up(a.Wo, 32); a.NT = N * a.ty * a.tx;
    a.dy_bytes = (unsigned)dyb;
    const int grid = stem_wgrad_grid(N, H, W);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CDET_BF16) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)stem_wgrad_mfma_kernel<CDET_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS);
            attr = true;
        }
        hipLaunchKernelGGL((stem_wgrad_mfma_kernel<CDET_BF16>), dim3(grid), dim3(256), SG_LDS, s, a);
    } else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)stem_wgrad_mfma_kernel<CDET_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS);
            attr = true;
        }
        hipLaunchKernelGGL((stem_wgrad_mfma_kernel<CDET_F16>), dim3(grid), dim3(256), SG_LDS, s, a);
    }
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3(div_up(Cout * 27, 16)), dim3(256), 0, s, ws, grid, Cout, dw, accumulate);
    CDET_LAUNCH_CHECK();
    return 0;
}
