// Shared device helpers of the transpose-read weight-gradient kernels (conv_wgrad_halo.hip, conv_wgrad_s2.hip): asynchronous
// ds_read_b64_tr_b16 with explicit lgkmcnt accounting, the 16x16x32 MFMA, the 1 KiB LDS-DMA piece.
#pragma once
#include "common.h"

#include <utility>

namespace cdet {

constexpr unsigned WH_SENT = 0xE0000000u;  // voffset beyond every buffer of the path: the DMA piece fetches zeros

template <int... I, class F>
__device__ __forceinline__ void wh_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int OFF>
__device__ __forceinline__ u32x2 wh_tr(int addr) {  // asynchronous: the result is valid after wh_wait<>() on it
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// the waits take the registers they make valid as in/out operands: every use the compiler schedules comes after the wait
template <int N>
__device__ __forceinline__ void wh_wait(u32x2& a, u32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wh_wait_b(u32x2 (&lo)[5], u32x2 (&hi)[5], u32x2& a, u32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]),
                   "+v"(hi[4]), "+v"(a), "+v"(b)
                 : "n"(N));
}
__device__ __forceinline__ int wh_sel(uint64_t m, int a_valid, int a_zero) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a_zero), "v"(a_valid), "s"(m));
    return r;
}

template <int DT>
__device__ __forceinline__ void wh_mfma(const u32x4& a, const u32x4& b, f32x4& c) {
    if (DT == CDET_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ void wh_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

}  // namespace cdet
