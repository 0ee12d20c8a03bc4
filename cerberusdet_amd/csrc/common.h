// Shared helpers for the gfx950 kernels of libcerberus_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cerberus_hip.h"
#include "switches.h"

namespace cdet {

void set_error(const char* fmt, ...);

#define CDET_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            cdet::set_error(__VA_ARGS__);         \
            return -1001;                         \
        }                                         \
    } while (0)

#define CDET_LAUNCH_CHECK()                                                             \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            cdet::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return -(int)e__;                                                           \
        }                                                                               \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- scalar conversions (round-to-nearest-even) -------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

template <int DT> struct Elem;  // 16-bit storage types
template <> struct Elem<CDET_BF16> {
    static __device__ __forceinline__ float to_f32(uint16_t v) { return bf16_bits_to_f32(v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_bf16_bits(f); }
};
template <> struct Elem<CDET_F16> {
    static __device__ __forceinline__ float to_f32(uint16_t v) { return f16_bits_to_f32(v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_f16_bits(f); }
};

__device__ __forceinline__ float load_elem(const void* p, int64_t i, int dtype) {
    if (dtype == CDET_F32) return ((const float*)p)[i];
    uint16_t v = ((const uint16_t*)p)[i];
    return dtype == CDET_BF16 ? bf16_bits_to_f32(v) : f16_bits_to_f32(v);
}
__device__ __forceinline__ void store_elem(void* p, int64_t i, float f, int dtype) {
    if (dtype == CDET_F32) ((float*)p)[i] = f;
    else ((uint16_t*)p)[i] = dtype == CDET_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
}

__device__ __forceinline__ float silu_f(float a) { return a / (1.0f + __expf(-a)); }
__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + __expf(-a)); }

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// tap-resident weight gradient (conv_wgrad_halo.hip), used by cdet_conv2d_wgrad (conv_wgrad.hip) for the stride-1 3x3 layers
struct WgradHaloPlan {
    int S, chunk, Kp, Cd_pad, n_cblk, n_iblk, XH, nci, narrow, patch;
    size_t lds;
};
bool wgrad_halo_plan(const cdet_conv_desc* d, WgradHaloPlan* out, int n_items = 1);
// items == nullptr: one layer (x, dy); else n_items layers of the same geometry, tensors from the device table
int wgrad_halo_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s,
                      const cdet_wgrad_item* items_dev = nullptr, int n_items = 1);
// Tuning / ablation switches exist in profiling builds only (make EXTRA=-DCDET_PROFILING): several of them change results (dropped
// statistics, skipped MFMAs), so the shipped library ignores the environment and always takes the default.
static inline int tune_env(const char* name, int dflt) {
#ifdef CDET_PROFILING
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
// 1x1 layers (conv_wgrad_halo.hip, wgrad_gemm_kernel): same plan struct, slab rows of n_iblk * 288 floats
bool wgrad_gemm_plan(const cdet_conv_desc* d, WgradHaloPlan* out);
int wgrad_gemm_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s);
// stride-2 3x3 layers (conv_wgrad_s2.hip): same plan struct and slab layout as the stride-1 form
bool wgrad_s2_plan(const cdet_conv_desc* d, WgradHaloPlan* out);
int wgrad_s2_launch(const cdet_conv_desc* d, const WgradHaloPlan& p, const void* x, const void* dy, float* ws, hipStream_t s);
static inline int elem_size(int dtype) { return dtype == CDET_F32 ? 4 : (dtype == CDET_U8 ? 1 : 2); }

}  // namespace cdet
