// Device helpers shared by the tap-resident convolution kernels (conv_halo.hip: stride-1 3x3 / 1x1; conv_vt.hip: the stride-2
// forward and the stride-2 data-gradient classes): MFMA 32x32x16 wrapper, LDS-DMA piece, counted waits, packing, half-wave sums.
#pragma once
#include "common.h"

namespace cdet {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HP = 256;             // pixels per block (128 in the half-tile form, template parameter NG = 1)
constexpr int HROW = 64;            // bytes per LDS row (32 channels)
constexpr int HZERO = 256;          // LDS bytes reserved in front (zero row)
constexpr int MAXXP = 7;            // X DMA pieces (16 rows each) per wave per chunk: XH <= 448
constexpr unsigned HSENT = 0xE0000000u;  // byte offset beyond every buffer: the DMA returns zeros
constexpr int PATCH_W = 16, PATCH_HPW = PATCH_W + 2;  // patch mode: 16 x 16 pixel tiles, halo pitch 18
constexpr int HEPI_RAW = 0, HEPI_FULL = 1, HEPI_F32 = 2;  // epilogue: raw 16-bit output (+ BN partial sums) / scale, bias, SiLU, residual / the same
                                                      // arithmetic into an fp32 destination, optionally accumulating onto it

// Virtual Concat (+ nearest 2x Upsample) as the source of a 1x1 convolution (models/common.py:288-295 in front of every neck C2f's cv1): up to three
// channel segments, each a slice of its own NHWC buffer, visited in order by the K loop; segment s covers K chunks [c0[s], c0[s + 1]). `up`: the
// segment is read through a nearest-neighbour 2x upsample (source pixel (y / 2, x / 2) of an H/2 x W/2 map) -- neither the upsampled map nor the
// concatenated buffer is ever written. All but the last segment hold a multiple of 32 channels.
struct CatSrcs {  // (scalar fields on purpose: arrays inside a by-value kernel argument made hipcc keep a private-memory copy of the whole argument)
    const uint16_t *x0, *x1, *x2;
    unsigned b0, b1, b2;        // buffer extents (bytes)
    int ld0, ld1, ld2, co0, co1, co2, up0, up1, up2;
    int c1, c2;                 // first K chunk of segments 1 and 2 (0x7fffffff: absent)
    int n, H, W;                // segments; pixel geometry of the convolution (for the upsampled segments)
};

// byte offset (before the chunk offset and the 16-byte slot) of pixel g of the N x H x W pixel space inside a segment (scalars, not the struct:
// a reference to the by-value kernel argument makes the compiler keep a private-memory copy of it, and its loads share the DMA's counter)
__device__ __forceinline__ unsigned cat_pixel_off(int up, int ld, int coff, int H, int W, int g) {
    int gs = g;
    if (up) {
        const int hw = H * W;
        const int n = g / hw, r = g - n * hw;
        const int y = r / W, x = r - y * W;
        gs = (n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    }
    return ((unsigned)gs * (unsigned)ld + (unsigned)coff) * 2u;
}

template <int DT>
__device__ __forceinline__ void mfma32(const u32x4& a, const u32x4& b, f32x16& c) {
    if (DT == CDET_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// the same MFMA with the accumulator in AccVGPRs (the 512-pixel tile form, conv_halo.hip NG = 4: 20 accumulator tiles = 320 registers per lane, of
// which 16 tiles live in the wave's 256 AccVGPRs and 4 in ArchVGPRs; one wave per SIMD owns all 512 registers of a lane)
template <int DT>
__device__ __forceinline__ void mfma32a(const u32x4& a, const u32x4& b, f32x16& c) {
    if (DT == CDET_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int AUX = 0>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, AUX);
}
#ifndef CDET_HALO_X_AUX
#define CDET_HALO_X_AUX 0  // default cache policy. Measured: non-temporal (2) on the pixel stream is SLOWER (40x40 320->320: 0.103 -> 0.112 ms) -- a halo row
                           // is fetched by both cout blocks of its tile and by the neighbouring tiles, and those re-reads want the L2 copy
#endif

// wave-uniform counted wait
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
}
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

typedef __attribute__((ext_vector_type(2))) float hf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 hf16x2;
template <int DT>
__device__ __forceinline__ uint32_t hpack2(float a, float b) {
    if (DT == CDET_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(hf32x2{a, b}, hbf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(hf32x2{a, b}, hf16x2));
}

// sum over the 32 lanes of a half wave (every lane ends up with the total)
__device__ __forceinline__ float half_sum32(float v) {
#define CDET_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CDET_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    CDET_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    CDET_DPP_ADD(0x141);  // row_half_mirror
    CDET_DPP_ADD(0x140);  // row_mirror
#undef CDET_DPP_ADD
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // lane ^ 16
    return v;
}

// BatchNorm partial sums of one 32 x 32 accumulator tile over its 32 pixels (the lanes of a half wave). s[r] / q[r]: the lane's contribution to
// sum / sum of squares of register r = cout 8 (r >> 2) + 4 h + (r & 3) of the tile. v_permlane16_swap pairs registers r and r + 8: after the swap
// + add the even 16-lane rows carry register r and the odd rows register r + 8, both already summed over lane ^ 16 -- eight values per quantity
// instead of sixteen go through the four DPP levels inside a row, and no LDS swizzle is involved (the former per-register form, 4 DPP adds +
// ds_swizzle + a predicated LDS store each, serialised on lgkmcnt: 7 us per workgroup with five tiles per wave). Lanes 0 / 16 / 32 / 48 store
// the 32 totals: ds[c], dq[c] for the tile-local cout c (16-byte aligned float rows).
__device__ __forceinline__ float row_sum16(float v) {
#define CDET_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CDET_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    CDET_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    CDET_DPP_ADD(0x141);  // row_half_mirror
    CDET_DPP_ADD(0x140);  // row_mirror
#undef CDET_DPP_ADD
    return v;
}

__device__ __forceinline__ void tile_stats32(const float (&s)[16], const float (&q)[16], float* ds, float* dq, int lane) {
    float ts[8], tq[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[r]), __float_as_uint(s[r + 8]), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[r]), __float_as_uint(q[r + 8]), false, false);
        const unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        ts[r] = row_sum16(__uint_as_float(a0) + __uint_as_float(a1));
        tq[r] = row_sum16(__uint_as_float(b0) + __uint_as_float(b1));
    }
    if ((lane & 15) == 0) {
        const int base = 4 * (lane >> 5) + 16 * ((lane >> 4) & 1);  // h, and which register of the pair this row carries
        *reinterpret_cast<f32x4*>(ds + base) = f32x4{ts[0], ts[1], ts[2], ts[3]};
        *reinterpret_cast<f32x4*>(ds + base + 8) = f32x4{ts[4], ts[5], ts[6], ts[7]};
        *reinterpret_cast<f32x4*>(dq + base) = f32x4{tq[0], tq[1], tq[2], tq[3]};
        *reinterpret_cast<f32x4*>(dq + base + 8) = f32x4{tq[4], tq[5], tq[6], tq[7]};
    }
}

}  // namespace cdet
