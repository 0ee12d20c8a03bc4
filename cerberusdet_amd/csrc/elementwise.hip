// HBM-bound NHWC kernels around the convolutions: train-mode BatchNorm+SiLU (fwd / bwd), channel-slice copies
// (Concat), nearest 2x upsample (+bwd), SPPF max-pool chain (+bwd). All of them move 16-byte (8-channel)
// vectors per lane; a thread keeps ONE channel vector for its whole life so that the per-channel parameters
// (mean, invstd, gamma, beta) stay in registers while it walks down the pixel rows.
#include <stdlib.h>

#include "common.h"
#include "bn_fold.h"

#include <type_traits>

namespace cdet {

struct Vec8 {
    float v[8];
};

template <int DT>
__device__ __forceinline__ Vec8 load8(const uint16_t* p) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
    Vec8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o.v[2 * i] = Elem<DT>::to_f32((uint16_t)(r[i] & 0xffff));
        o.v[2 * i + 1] = Elem<DT>::to_f32((uint16_t)(r[i] >> 16));
    }
    return o;
}
typedef __attribute__((ext_vector_type(2))) float f32x2_e;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_e;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_e;
template <int DT>
__device__ __forceinline__ uint32_t pack2e(float a, float b) {  // v_cvt_pk_bf16_f32 / cvt_f16 (round to nearest even)
    if (DT == CDET_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_e{a, b}, bf16x2_e));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_e{a, b}, f16x2_e));
}
template <int DT>
__device__ __forceinline__ void store8(uint16_t* p, const Vec8& x) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2e<DT>(x.v[2 * i], x.v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = r;
}
__device__ __forceinline__ Vec8 loadf8(const float* p) {
    Vec8 o;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o.v[i] = a[i];
        o.v[4 + i] = b[i];
    }
    return o;
}

// thread -> (channel vector cv, row lane rl); rows advance by rows_per_pass
struct ColMap {
    int cv, rl, rows_per_pass;
    bool active;
};
__device__ __forceinline__ ColMap col_map(int CV) {
    ColMap m;
    const int t = threadIdx.x;
    m.rows_per_pass = blockDim.x / CV;
    m.cv = t % CV;
    m.rl = t / CV;
    m.active = m.rl < m.rows_per_pass;
    return m;
}

// ------------------------------------------------------------------------------------------------
// BN finalize: partial sums [nblk][2][C] -> mean / invstd (+ running stats)
// ------------------------------------------------------------------------------------------------
// shared by bn_finalize_kernel / bn_bwd_sums_kernel: 16 channels x NG partial-row groups per workgroup (NG = blockDim.x / 16). The
// [nblk][2][C] partials are read in 64-byte runs; every group walks its rows with two independent fp32 chains that are spilled into
// double accumulators every 32 iterations; the NG group sums are combined in LDS in a fixed order (groups of 16, then the 4..NG/16
// group-of-group sums). These two kernels sit between every convolution and its BatchNorm pass (776 launches per iteration) with a
// handful of workgroups each, i.e. their LATENCY is on the critical path: 1024 threads (64 groups) instead of 256 cut the serial
// loop 4x (measured: skipping them entirely saves 7.1 ms of a 99 ms iteration).
constexpr int BN_RED_THREADS = 1024;
// Round 5: for launches of at most BNF_CL * BNF_MAX_CL = 4096 partial rows the sums follow the fixed TREE of bn_fold.h -- clusters of 32 consecutive
// rows added ascending in double, then the cluster sums added ascending in double -- which is the order the producers use when they reduce their
// own partials in-launch: the two forms give the same bits. (Longer lists -- the round-1 generic kernel's rows -- keep the strided two-chain order.)
__device__ __forceinline__ bool bn_reduce_partials(const float* __restrict__ part, int nblk, int C, int c, double& s, double& q) {
    constexpr int NG = BN_RED_THREADS / 16;
    __shared__ double sh[2][BNF_MAX_CL][16];
    static_assert(BNF_MAX_CL >= NG, "the strided form keeps one LDS row per thread group");
    const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
    s = 0.0;
    q = 0.0;
    const int ncl = (nblk + BNF_CL - 1) / BNF_CL;
    if (ncl <= BNF_MAX_CL) {
        for (int cl = g; cl < ncl; cl += NG) {
            double a = 0.0, b = 0.0;
            if (c < C) {
                const int r0 = cl * BNF_CL, n = min(BNF_CL, nblk - r0);
                const float* p0 = part + ((int64_t)r0 * 2) * C + c;
#pragma unroll 8
                for (int i = 0; i < n; ++i) {  // ascending, one double chain per quantity (the loads run ahead of the adds)
                    a += (double)p0[(int64_t)i * 2 * C];
                    b += (double)p0[(int64_t)i * 2 * C + C];
                }
            }
            sh[0][cl][cx] = a;
            sh[1][cl][cx] = b;
        }
        __syncthreads();
        if (g != 0 || c >= C) return false;
        for (int k = 0; k < ncl; ++k) {
            s += sh[0][k][cx];
            q += sh[1][k][cx];
        }
        return true;
    }
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        int b = g, it = 0;
        for (; b + NG < nblk; b += 2 * NG, ++it) {
            s0 += part[((int64_t)b * 2 + 0) * C + c];
            q0 += part[((int64_t)b * 2 + 1) * C + c];
            s1 += part[((int64_t)(b + NG) * 2 + 0) * C + c];
            q1 += part[((int64_t)(b + NG) * 2 + 1) * C + c];
            if ((it & 31) == 31) {
                s += (double)s0 + (double)s1;
                q += (double)q0 + (double)q1;
                s0 = s1 = q0 = q1 = 0.f;
            }
        }
        for (; b < nblk; b += NG) {
            s0 += part[((int64_t)b * 2 + 0) * C + c];
            q0 += part[((int64_t)b * 2 + 1) * C + c];
        }
        s += (double)s0 + (double)s1;
        q += (double)q0 + (double)q1;
    }
    sh[0][g][cx] = s;
    sh[1][g][cx] = q;
    __syncthreads();
    if (g < NG / 16) {  // NG/16 threads per channel add 16 groups each
        s = 0.0;
        q = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s += sh[0][g * 16 + k][cx];
            q += sh[1][g * 16 + k][cx];
        }
    }
    __syncthreads();
    if (g < NG / 16) {
        sh[0][g][cx] = s;
        sh[1][g][cx] = q;
    }
    __syncthreads();
    if (g != 0 || c >= C) return false;
    s = 0.0;
    q = 0.0;
#pragma unroll
    for (int k = 0; k < NG / 16; ++k) {
        s += sh[0][k][cx];
        q += sh[1][k][cx];
    }
    return true;
}

// deferred running-statistics updates of many layers, one workgroup (256 threads) per item: the arithmetic of bn_finalize_kernel on fp32 totals
__global__ __launch_bounds__(256) void bn_running_update_kernel(const cdet_bn_running_item* __restrict__ items) {
    const cdet_bn_running_item it = items[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256)
        bn_stats_from_totals(it.totals[c], it.totals[it.C + c], it.inv_count, it.unbias, 0.f, it.momentum, nullptr, nullptr, it.running_mean + c,
                             it.running_var + c);
}

__global__ __launch_bounds__(BN_RED_THREADS) void bn_finalize_kernel(const float* __restrict__ stats, int nblk, int C, double inv_count,
                                                                     double unbias, float eps, float momentum, float* running_mean,
                                                                     float* running_var, float* mean, float* invstd) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double s, q;
    if (!bn_reduce_partials(stats, nblk, C, c, s, q)) return;
    // The totals are rounded to fp32 before they are used: the SyncBatchNorm launch list all-reduces exactly these fp32 [sum, sumsq] vectors
    // (bn_bwd_sums_kernel -> all-reduce -> this kernel with nblk = 1), so the per-GPU list and the synchronised one derive mean / invstd from the
    // same numbers -- bit-identical at world 1, and exactly "twice the sums over twice the count" when two ranks hold the same shard
    // (tests/test_gpu_distributed.py).
    bn_stats_from_totals((float)s, (float)q, inv_count, unbias, eps, momentum, mean + c, invstd + c, running_mean ? running_mean + c : nullptr,
                         running_mean ? running_var + c : nullptr);
}

// ------------------------------------------------------------------------------------------------
// y = silu(gamma*(z-mean)*invstd + beta) (+ residual)
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const uint16_t* __restrict__ res, int res_ld, int res_coff,
                                                          uint16_t* __restrict__ y, int y_ld, int y_coff, int64_t M, int CV, int rev) {
    const ColMap cm = col_map(CV);
    if (!cm.active) return;
    const int c = cm.cv * 8;
    const Vec8 mu = loadf8(mean + c), is = loadf8(invstd + c), ga = loadf8(gamma + c), be = loadf8(beta + c);
    Vec8 sc, sh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc.v[i] = ga.v[i] * is.v[i];
        sh.v[i] = be.v[i] - mu.v[i] * sc.v[i];
    }
    // rev: walk the rows from the end (the producer wrote them front to back: its tail is what the L2 / MALL still hold)
    const int64_t step = (int64_t)gridDim.x * cm.rows_per_pass;
    int64_t r = (int64_t)blockIdx.x * cm.rows_per_pass + cm.rl;
    auto R = [&](int64_t i) { return rev ? M - 1 - i : i; };
    for (; r + 3 * step < M; r += 4 * step) {  // 4 independent rows in flight per thread
        u32x4 zr[4], rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) zr[u] = *reinterpret_cast<const u32x4*>(z + R(r + u * step) * z_ld + z_coff + c);
        if (res) {
#pragma unroll
            for (int u = 0; u < 4; ++u) rr[u] = *reinterpret_cast<const u32x4*>(res + R(r + u * step) * res_ld + res_coff + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Vec8 x;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x.v[2 * i] = Elem<DT>::to_f32((uint16_t)(zr[u][i] & 0xffff));
                x.v[2 * i + 1] = Elem<DT>::to_f32((uint16_t)(zr[u][i] >> 16));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = x.v[i] * sc.v[i] + sh.v[i];
                x.v[i] = a * __builtin_amdgcn_rcpf(1.0f + __expf(-a));
            }
            if (res) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x.v[2 * i] += Elem<DT>::to_f32((uint16_t)(rr[u][i] & 0xffff));
                    x.v[2 * i + 1] += Elem<DT>::to_f32((uint16_t)(rr[u][i] >> 16));
                }
            }
            store8<DT>(y + R(r + u * step) * y_ld + y_coff + c, x);
        }
    }
    for (; r < M; r += step) {
        Vec8 x = load8<DT>(z + R(r) * z_ld + z_coff + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = x.v[i] * sc.v[i] + sh.v[i];
            x.v[i] = a * __builtin_amdgcn_rcpf(1.0f + __expf(-a));
        }
        if (res) {
            const Vec8 rr = load8<DT>(res + R(r) * res_ld + res_coff + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] += rr.v[i];
        }
        store8<DT>(y + R(r) * y_ld + y_coff + c, x);
    }
}

// d silu(a)/da = s*(1 + a*(1-s)), s = sigmoid(a)
__device__ __forceinline__ float dsilu_f(float a) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
    return s * (1.f + a * (1.f - s));
}

// pass 1: per-channel partial sums of dact and dact*xhat
template <int DT>
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce_kernel(const uint16_t* __restrict__ dy, int dy_ld, int dy_coff,
                                                                 const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ part, int64_t M, int C, int CV) {
    extern __shared__ float shm[];  // [rows_per_pass][2][C]
    const ColMap cm = col_map(CV);
    const int c = cm.cv * 8;
    Vec8 s1, s2;
#pragma unroll
    for (int i = 0; i < 8; ++i) s1.v[i] = s2.v[i] = 0.f;
    if (cm.active) {
        const Vec8 mu = loadf8(mean + c), is = loadf8(invstd + c), ga = loadf8(gamma + c), be = loadf8(beta + c);
        const int64_t step = (int64_t)gridDim.x * cm.rows_per_pass;
        int64_t r = (int64_t)blockIdx.x * cm.rows_per_pass + cm.rl;
        for (; r + 3 * step < M; r += 4 * step) {
            Vec8 g[4], x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = load8<DT>(dy + (r + u * step) * dy_ld + dy_coff + c);
                x[u] = load8<DT>(z + (r + u * step) * z_ld + z_coff + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (x[u].v[i] - mu.v[i]) * is.v[i];
                    const float da = g[u].v[i] * dsilu_f(ga.v[i] * xh + be.v[i]);
                    s1.v[i] += da;
                    s2.v[i] += da * xh;
                }
        }
        for (; r < M; r += step) {
            const Vec8 g = load8<DT>(dy + r * dy_ld + dy_coff + c);
            const Vec8 x = load8<DT>(z + r * z_ld + z_coff + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (x.v[i] - mu.v[i]) * is.v[i];
                const float da = g.v[i] * dsilu_f(ga.v[i] * xh + be.v[i]);
                s1.v[i] += da;
                s2.v[i] += da * xh;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            shm[(cm.rl * 2 + 0) * C + c + i] = s1.v[i];
            shm[(cm.rl * 2 + 1) * C + c + i] = s2.v[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < cm.rows_per_pass; ++r) acc += shm[r * 2 * C + j];
        part[(int64_t)blockIdx.x * 2 * C + j] = acc;
    }
}

// pass 2a: reduce partials -> sums (and dgamma / dbeta)
__global__ __launch_bounds__(BN_RED_THREADS) void bn_bwd_sums_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ sums,
                                                                     float* dgamma, float* dbeta, int accumulate) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double s, q;
    if (!bn_reduce_partials(part, nblk, C, c, s, q)) return;
    sums[c] = (float)s;
    sums[C + c] = (float)q;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
}

// pass 2b: dz = gamma*invstd*(dact - mean(dact) - xhat*mean(dact*xhat))
// ALSO: the same pass adds dy into a second gradient slice, also[r][c] += dy[r][c] -- the shortcut addend of a Bottleneck
// (x + cv2(cv1(x)), models/common.py:107-117): the gradient of the sum goes to both addends, and cv2's backward is the pass that
// has it in registers (a separate add kernel re-read it: 3 tensor passes instead of 2).
template <int DT, bool ALSO>
__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const uint16_t* __restrict__ dy, int dy_ld, int dy_coff,
                                                                const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ sums, float inv_count,
                                                                uint16_t* __restrict__ dz, int dz_ld, int dz_coff, int64_t M, int C, int CV, int rev,
                                                                uint16_t* __restrict__ also, int also_ld, int also_coff) {
    const ColMap cm = col_map(CV);
    if (!cm.active) return;
    const int c = cm.cv * 8;
    const Vec8 mu = loadf8(mean + c), is = loadf8(invstd + c), ga = loadf8(gamma + c), be = loadf8(beta + c);
    Vec8 m1 = loadf8(sums + c), m2 = loadf8(sums + C + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        m1.v[i] *= inv_count;
        m2.v[i] *= inv_count;
    }
    const int64_t step = (int64_t)gridDim.x * cm.rows_per_pass;
    int64_t r = (int64_t)blockIdx.x * cm.rows_per_pass + cm.rl;
    auto R = [&](int64_t i) { return rev ? M - 1 - i : i; };
    for (; r + 3 * step < M; r += 4 * step) {
        Vec8 g[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[u] = load8<DT>(dy + R(r + u * step) * dy_ld + dy_coff + c);
            x[u] = load8<DT>(z + R(r + u * step) * z_ld + z_coff + c);
        }
        if (ALSO) {
            Vec8 s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = load8<DT>(also + R(r + u * step) * also_ld + also_coff + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[u].v[i] += g[u].v[i];
                store8<DT>(also + R(r + u * step) * also_ld + also_coff + c, s[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Vec8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (x[u].v[i] - mu.v[i]) * is.v[i];
                const float da = g[u].v[i] * dsilu_f(ga.v[i] * xh + be.v[i]);
                o.v[i] = ga.v[i] * is.v[i] * (da - m1.v[i] - xh * m2.v[i]);
            }
            store8<DT>(dz + R(r + u * step) * dz_ld + dz_coff + c, o);
        }
    }
    for (; r < M; r += step) {
        const Vec8 g = load8<DT>(dy + R(r) * dy_ld + dy_coff + c);
        const Vec8 x = load8<DT>(z + R(r) * z_ld + z_coff + c);
        if (ALSO) {
            Vec8 s = load8<DT>(also + R(r) * also_ld + also_coff + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) s.v[i] += g.v[i];
            store8<DT>(also + R(r) * also_ld + also_coff + c, s);
        }
        Vec8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (x.v[i] - mu.v[i]) * is.v[i];
            const float da = g.v[i] * dsilu_f(ga.v[i] * xh + be.v[i]);
            o.v[i] = ga.v[i] * is.v[i] * (da - m1.v[i] - xh * m2.v[i]);
        }
        store8<DT>(dz + R(r) * dz_ld + dz_coff + c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: lean forms of the three BatchNorm + SiLU tensor passes. The kernels above run at the same time per ELEMENT whether they move
// 4 or 6 bytes per element (profiles/r03_pmc_traffic.json: reduce 4.0, forward 4.8, apply 5.2 TB/s) -- they are bound by their VALU
// stream (13 - 15 instruction slots per element: 64-bit address arithmetic per row and tensor, three-op normalisation, unpaired fp32
// ops around the two transcendentals), not by HBM. Here:
//   * tensors are addressed through buffer descriptors: one 32-bit byte offset per tensor and thread, advanced by ONE add per row, and
//     rows beyond M read zeros / drop their stores (no tail loop);
//   * all arithmetic on pairs (v_pk_fma / v_pk_mul / v_pk_add_f32), the per-channel affine maps folded on the host side of the loop:
//     a = x k1 + k0 (one fma instead of sub, mul, fma); the backward sums as S1 = sum da and S2' = sum da x (xhat never formed in the
//     loop: sum da xhat = invstd (S2' - mean S1)); dz = da K + x A + B;
//   => 3.5 / 6.5 / 7 paired-instruction slots per element + the two transcendentals.
// Used when every tensor of the call is below 2 GiB (32-bit offsets with room for the row-ahead unroll); CDET_BN_V2=0 (profiling
// builds) selects the kernels above for A/B timing.
// ------------------------------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f2v unpack2(uint32_t w) {
    if (DT == CDET_BF16) return f2v{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    const f16x2_e h = __builtin_bit_cast(f16x2_e, w);
    return f2v{(float)h[0], (float)h[1]};
}
__device__ __forceinline__ f2v sigmoid_from(f2v a) {  // 1 / (1 + exp(-a)), the same instruction sequence as the scalar form above
    const f2v t = a * -1.4426950408889634f;
    const f2v e = f2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f2v d = e + 1.0f;
    return f2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bn_rsrc(const void* p, int64_t M, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)(M * ld * 2), 0x00020000);
}
struct Pair4 {
    f2v v[4];
};
__device__ __forceinline__ Pair4 loadf8p(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    Pair4 o;
    o.v[0] = f2v{a[0], a[1]}; o.v[1] = f2v{a[2], a[3]}; o.v[2] = f2v{b[0], b[1]}; o.v[3] = f2v{b[2], b[3]};
    return o;
}

constexpr int BN2_U = 4;  // rows in flight per thread

template <int DT>
__global__ __launch_bounds__(256) void bn_silu_fwd2_kernel(const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const uint16_t* __restrict__ res, int res_ld, int res_coff,
                                                           uint16_t* __restrict__ y, int y_ld, int y_coff, int64_t M, int CV, int rev) {
    const ColMap cm = col_map(CV);
    if (!cm.active) return;
    const int c = cm.cv * 8;
    const Pair4 mu = loadf8p(mean + c), is = loadf8p(invstd + c), ga = loadf8p(gamma + c), be = loadf8p(beta + c);
    f2v sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = ga.v[i] * is.v[i];
        sh[i] = be.v[i] - mu.v[i] * sc[i];
    }
    const __amdgpu_buffer_rsrc_t rz = bn_rsrc(z, M, z_ld), rr = bn_rsrc(res ? res : z, res ? M : 0, res_ld), ry = bn_rsrc(y, M, y_ld);
    const int step = (int)gridDim.x * cm.rows_per_pass;
    const int r0 = (int)blockIdx.x * cm.rows_per_pass + cm.rl;
    const int row = rev ? (int)M - 1 - r0 : r0, dir = rev ? -1 : 1;
    unsigned oz = (unsigned)((row * z_ld + z_coff + c) * 2), orr = (unsigned)((row * res_ld + res_coff + c) * 2), oy = (unsigned)((row * y_ld + y_coff + c) * 2);
    const unsigned iz = (unsigned)(dir * step * z_ld * 2), ir = (unsigned)(dir * step * res_ld * 2), iy = (unsigned)(dir * step * y_ld * 2);
    auto rows = [&](auto UC) __attribute__((always_inline)) {  // UC rows in flight, all inside [0, M)
        constexpr int U = decltype(UC)::value;
        u4v zr[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            zr[u] = __builtin_amdgcn_raw_buffer_load_b128(rz, (int)(oz + u * iz), 0, 0);
            if (res) rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(orr + u * ir), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u4v o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2v a = unpack2<DT>(zr[u][i]) * sc[i] + sh[i];
                f2v v = a * sigmoid_from(a);
                if (res) v += unpack2<DT>(rv[u][i]);
                o[i] = pack2e<DT>(v.x, v.y);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, ry, (int)(oy + u * iy), 0, 0);
        }
        oz += U * iz; orr += U * ir; oy += U * iy;
    };
    int r = r0;
    for (; r + (BN2_U - 1) * step < M; r += BN2_U * step) rows(std::integral_constant<int, BN2_U>{});
    for (; r < M; r += step) rows(std::integral_constant<int, 1>{});
}

// da = dy * silu'(a), a = x k1 + k0; silu'(a) = s (1 + a (1 - s))
__device__ __forceinline__ f2v dact_of(f2v g, f2v x, f2v k1, f2v k0) {
    const f2v a = x * k1 + k0;
    const f2v s = sigmoid_from(a);
    const f2v w = a * (1.0f - s) + 1.0f;
    return g * (s * w);
}

template <int DT>
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce2_kernel(const uint16_t* __restrict__ dy, int dy_ld, int dy_coff,
                                                                  const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ part, int64_t M, int C, int CV,
                                                                  const BnFold* __restrict__ fold) {
    extern __shared__ float shm[];  // [rows_per_pass][2][C]
    const ColMap cm = col_map(CV);
    const int c = cm.cv * 8;
    if (cm.active) {
        const Pair4 mu = loadf8p(mean + c), is = loadf8p(invstd + c), ga = loadf8p(gamma + c), be = loadf8p(beta + c);
        f2v k1[4], k0[4], s1[4], s2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            k1[i] = ga.v[i] * is.v[i];
            k0[i] = be.v[i] - mu.v[i] * k1[i];
            s1[i] = s2[i] = f2v{0.f, 0.f};
        }
        const __amdgpu_buffer_rsrc_t rg = bn_rsrc(dy, M, dy_ld), rz = bn_rsrc(z, M, z_ld);
        const int step = (int)gridDim.x * cm.rows_per_pass;
        const int r0 = (int)blockIdx.x * cm.rows_per_pass + cm.rl;
        unsigned og = (unsigned)((r0 * dy_ld + dy_coff + c) * 2), oz = (unsigned)((r0 * z_ld + z_coff + c) * 2);
        const unsigned ig = (unsigned)(step * dy_ld * 2), iz = (unsigned)(step * z_ld * 2);
        auto rows = [&](auto UC) __attribute__((always_inline)) {
            constexpr int U = decltype(UC)::value;
            u4v gr[U], zr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gr[u] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(og + u * ig), 0, 0);
                zr[u] = __builtin_amdgcn_raw_buffer_load_b128(rz, (int)(oz + u * iz), 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f2v x = unpack2<DT>(zr[u][i]);
                    const f2v da = dact_of(unpack2<DT>(gr[u][i]), x, k1[i], k0[i]);
                    s1[i] += da;
                    s2[i] = da * x + s2[i];
                }
            og += U * ig; oz += U * iz;
        };
        int r = r0;
        for (; r + (BN2_U - 1) * step < M; r += BN2_U * step) rows(std::integral_constant<int, BN2_U>{});
        for (; r < M; r += step) rows(std::integral_constant<int, 1>{});
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // sum da xhat = invstd (sum da x - mean sum da)
            const f2v q = is.v[i] * (s2[i] - mu.v[i] * s1[i]);
            shm[(cm.rl * 2 + 0) * C + c + 2 * i] = s1[i].x;
            shm[(cm.rl * 2 + 0) * C + c + 2 * i + 1] = s1[i].y;
            shm[(cm.rl * 2 + 1) * C + c + 2 * i] = q.x;
            shm[(cm.rl * 2 + 1) * C + c + 2 * i + 1] = q.y;
        }
    }
    __syncthreads();
    if (!CDET_FOLD(fold)) {
        for (int j = threadIdx.x; j < 2 * C; j += blockDim.x) {
            float acc = 0.f;
            for (int r = 0; r < cm.rows_per_pass; ++r) acc += shm[r * 2 * C + j];
            part[(int64_t)blockIdx.x * 2 * C + j] = acc;
        }
        return;
    }
    // the partial rows are reduced in this launch (bn_fold.h): write-through row, ticket, the last arrivers add up in the fixed tree order
    const __amdgpu_buffer_rsrc_t rs_ = bnf_rsrc(part);
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < cm.rows_per_pass; ++r) acc += shm[r * 2 * C + j];
        bnf_stf(rs_, (unsigned)(((int64_t)blockIdx.x * 2 * C + j) * 4), acc);
    }
    bn_fold_finish<false>(fold, part, (int)blockIdx.x, 0, C, 0, reinterpret_cast<volatile int*>(shm));
}

template <int DT, bool ALSO>
__global__ __launch_bounds__(256) void bn_silu_bwd_apply2_kernel(const uint16_t* __restrict__ dy, int dy_ld, int dy_coff,
                                                                 const uint16_t* __restrict__ z, int z_ld, int z_coff,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ sums, float inv_count,
                                                                 uint16_t* __restrict__ dz, int dz_ld, int dz_coff, int64_t M, int C, int CV, int rev,
                                                                 uint16_t* __restrict__ also, int also_ld, int also_coff) {
    const ColMap cm = col_map(CV);
    if (!cm.active) return;
    const int c = cm.cv * 8;
    const Pair4 mu = loadf8p(mean + c), is = loadf8p(invstd + c), ga = loadf8p(gamma + c), be = loadf8p(beta + c);
    const Pair4 m1 = loadf8p(sums + c), m2 = loadf8p(sums + C + c);
    // dz = gamma invstd (da - m1 - xhat m2), xhat = (x - mean) invstd   ->   dz = da K + x A + B
    f2v k1[4], k0[4], K[4], A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k1[i] = ga.v[i] * is.v[i];
        k0[i] = be.v[i] - mu.v[i] * k1[i];
        K[i] = k1[i];
        const f2v km2 = k1[i] * (m2.v[i] * inv_count) * is.v[i];
        A[i] = -km2;
        B[i] = km2 * mu.v[i] - k1[i] * (m1.v[i] * inv_count);
    }
    const __amdgpu_buffer_rsrc_t rg = bn_rsrc(dy, M, dy_ld), rz = bn_rsrc(z, M, z_ld), ro = bn_rsrc(dz, M, dz_ld),
                                 ra = bn_rsrc(ALSO ? also : dz, ALSO ? M : 0, also_ld);
    const int step = (int)gridDim.x * cm.rows_per_pass;
    const int r0 = (int)blockIdx.x * cm.rows_per_pass + cm.rl;
    const int row = rev ? (int)M - 1 - r0 : r0, dir = rev ? -1 : 1;
    unsigned og = (unsigned)((row * dy_ld + dy_coff + c) * 2), oz = (unsigned)((row * z_ld + z_coff + c) * 2),
             oo = (unsigned)((row * dz_ld + dz_coff + c) * 2), oa = (unsigned)((row * also_ld + also_coff + c) * 2);
    const unsigned ig = (unsigned)(dir * step * dy_ld * 2), iz = (unsigned)(dir * step * z_ld * 2), io = (unsigned)(dir * step * dz_ld * 2),
                   ia = (unsigned)(dir * step * also_ld * 2);
    auto rows = [&](auto UC) __attribute__((always_inline)) {
        constexpr int U = decltype(UC)::value;
        u4v gr[U], zr[U], ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gr[u] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(og + u * ig), 0, 0);
            zr[u] = __builtin_amdgcn_raw_buffer_load_b128(rz, (int)(oz + u * iz), 0, 0);
            if (ALSO) ar[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(oa + u * ia), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u4v o, s_;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2v g = unpack2<DT>(gr[u][i]), x = unpack2<DT>(zr[u][i]);
                const f2v da = dact_of(g, x, k1[i], k0[i]);
                const f2v v = da * K[i] + (x * A[i] + B[i]);
                o[i] = pack2e<DT>(v.x, v.y);
                if (ALSO) {
                    const f2v w = unpack2<DT>(ar[u][i]) + g;
                    s_[i] = pack2e<DT>(w.x, w.y);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)(oo + u * io), 0, 0);
            if (ALSO) __builtin_amdgcn_raw_buffer_store_b128(s_, ra, (int)(oa + u * ia), 0, 0);
        }
        og += U * ig; oz += U * iz; oo += U * io; oa += U * ia;
    };
    int r = r0;
    for (; r + (BN2_U - 1) * step < M; r += BN2_U * step) rows(std::integral_constant<int, BN2_U>{});
    for (; r < M; r += step) rows(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------------
// channel-slice copy / add, upsample, SPPF pools
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void copy_channels_kernel(const uint16_t* __restrict__ src, int s_ld, int s_coff, uint16_t* __restrict__ dst,
                                                            int d_ld, int d_coff, int64_t M, int CV, int accumulate) {
    const int64_t total = M * CV;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = v / CV;
        const int c = (int)(v - r * CV) * 8;
        if (accumulate) {
            Vec8 a = load8<DT>(src + r * s_ld + s_coff + c);
            const Vec8 b = load8<DT>(dst + r * d_ld + d_coff + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) a.v[i] += b.v[i];
            store8<DT>(dst + r * d_ld + d_coff + c, a);
        } else {
            *reinterpret_cast<u32x4*>(dst + r * d_ld + d_coff + c) = *reinterpret_cast<const u32x4*>(src + r * s_ld + s_coff + c);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void add_channels_kernel(const uint16_t* __restrict__ a, int a_ld, int a_coff, const uint16_t* __restrict__ b,
                                                           int b_ld, int b_coff, uint16_t* __restrict__ y, int y_ld, int y_coff, int64_t M, int CV) {
    const int64_t total = M * CV;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = v / CV;
        const int c = (int)(v - r * CV) * 8;
        Vec8 x = load8<DT>(a + r * a_ld + a_coff + c);
        const Vec8 z = load8<DT>(b + r * b_ld + b_coff + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) x.v[i] += z.v[i];
        store8<DT>(y + r * y_ld + y_coff + c, x);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void upsample2_kernel(const uint16_t* __restrict__ src, int s_ld, int s_coff, uint16_t* __restrict__ dst,
                                                        int d_ld, int d_coff, int N, int H, int W, int CV) {
    const int64_t total = (int64_t)N * 2 * H * 2 * W * CV;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = v / CV;
        const int c = (int)(v - p * CV) * 8;
        const int x = (int)(p % (2 * W));
        const int64_t t = p / (2 * W);
        const int y = (int)(t % (2 * H));
        const int n = (int)(t / (2 * H));
        const int64_t sp = ((int64_t)n * H + (y >> 1)) * W + (x >> 1);
        *reinterpret_cast<u32x4*>(dst + p * d_ld + d_coff + c) = *reinterpret_cast<const u32x4*>(src + sp * s_ld + s_coff + c);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const uint16_t* __restrict__ dd, int dd_ld, int dd_coff, uint16_t* __restrict__ ds,
                                                            int ds_ld, int ds_coff, int N, int H, int W, int CV, int accumulate) {
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = v / CV;
        const int c = (int)(v - p * CV) * 8;
        const int x = (int)(p % W);
        const int64_t t = p / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        Vec8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = 0.f;
        if (accumulate) acc = load8<DT>(ds + p * ds_ld + ds_coff + c);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t q = ((int64_t)n * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx;
                const Vec8 g = load8<DT>(dd + q * dd_ld + dd_coff + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc.v[i] += g.v[i];
            }
        store8<DT>(ds + p * ds_ld + ds_coff + c, acc);
    }
}

// 5x5 stride-1 pad-2 max pool from channel slice `coff` into slice `coff + C`; first maximum in (ky, kx) scan order wins
// (what torch's max_pool2d backward routes to).
template <int DT>
__global__ __launch_bounds__(256) void pool5_kernel(uint16_t* __restrict__ buf, int ld, int coff_in, int coff_out, int N, int H, int W, int CV) {
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = v / CV;
        const int c = (int)(v - p * CV) * 8;
        const int x = (int)(p % W);
        const int64_t t = p / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        Vec8 best;
#pragma unroll
        for (int i = 0; i < 8; ++i) best.v[i] = -INFINITY;
        for (int ky = -2; ky <= 2; ++ky) {
            const int yy = y + ky;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kx = -2; kx <= 2; ++kx) {
                const int xx = x + kx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const Vec8 s = load8<DT>(buf + (((int64_t)n * H + yy) * W + xx) * ld + coff_in + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) best.v[i] = fmaxf(best.v[i], s.v[i]);
            }
        }
        store8<DT>(buf + p * ld + coff_out + c, best);
    }
}

// Round 4: the three chained 5x5 pools of an SPPF (models/common.py:230-245: y1 = m(x), y2 = m(y1), y3 = m(y2)) as ONE launch when the map fits
// LDS (20 x 20 at 640 px): a workgroup owns one image x 32 channels, keeps the plane in LDS and runs each stage separably (row maximum, then
// column maximum -- the same window maximum, exact for 16-bit values), writing every stage's slice as it goes. Same bits as three pool5
// launches; two launches and two round trips through L2 less on the serial trunk.
template <int DT>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(uint16_t* __restrict__ buf, int ld, int coff, int C, int H, int W, int CV) {
    // (round 5: the planes live in LDS as fp32 -- widened once at the load, narrowed once per stored slice; a window maximum is then 8 v_max_f32 per
    //  neighbour instead of 16 conversions + 8 maxima + 8 conversions on packed 16-bit pairs -- and a workgroup owns 16 channels instead of 32, so the
    //  20 x 20 level of a batch-32 forward is 640 workgroups in one round instead of 320 in two: 77 -> see profiles/r05_fwd_shapes_eval.txt)
    extern __shared__ __attribute__((aligned(16))) unsigned char sp[];
    constexpr int G = 2;                                   // 8-channel vectors per workgroup
    const int groups = (CV + G - 1) / G;
    const int n = blockIdx.x / groups, g0 = (blockIdx.x % groups) * G;
    const int ng = CV - g0 < G ? CV - g0 : G;
    const int HW = H * W;
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8* const A = reinterpret_cast<f32x8*>(sp);         // [HW][G]
    f32x8* const B = A + HW * G;
    uint16_t* const base = buf + (int64_t)n * HW * ld;
    for (int e = threadIdx.x; e < HW * ng; e += 256) {
        const int p = e / ng, g = e - p * ng;
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + (int64_t)p * ld + coff + (g0 + g) * 8);
        f32x8 f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = Elem<DT>::to_f32((uint16_t)(v[i] & 0xffff));
            f[2 * i + 1] = Elem<DT>::to_f32((uint16_t)(v[i] >> 16));
        }
        A[p * G + g] = f;
    }
    __syncthreads();
    auto vmax = [](const f32x8& a, const f32x8& b) {
        f32x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaxf(a[i], b[i]);
        return o;
    };
    for (int stage = 0; stage < 3; ++stage) {
        for (int e = threadIdx.x; e < HW * ng; e += 256) {  // rows: B = max over x - 2 .. x + 2 of A
            const int p = e / ng, g = e - p * ng;
            const int x = p % W;
            f32x8 m = A[p * G + g];
#pragma unroll
            for (int k = -2; k <= 2; ++k)
                if (k != 0 && (unsigned)(x + k) < (unsigned)W) m = vmax(m, A[(p + k) * G + g]);
            B[p * G + g] = m;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < HW * ng; e += 256) {  // columns: A = max over y - 2 .. y + 2 of B, and out
            const int p = e / ng, g = e - p * ng;
            const int y = p / W;
            f32x8 m = B[p * G + g];
#pragma unroll
            for (int k = -2; k <= 2; ++k)
                if (k != 0 && (unsigned)(y + k) < (unsigned)H) m = vmax(m, B[(p + k * W) * G + g]);
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (uint32_t)Elem<DT>::from_f32(m[2 * i]) | ((uint32_t)Elem<DT>::from_f32(m[2 * i + 1]) << 16);
            *reinterpret_cast<u32x4*>(base + (int64_t)p * ld + coff + (stage + 1) * C + (g0 + g) * 8) = o;
            if (stage < 2) A[p * G + g] = m;                // (nobody reads A in this pass)
        }
        __syncthreads();
    }
}

// backward of one pool stage: din[q] += sum over windows p containing q of dout[p] * [argmax_p == q], argmax_p = first
// maximum of window p in (ky, kx) scan order, recomputed from the saved forward input slice. One block owns a T x T pixel
// tile of one image for one 8-channel vector (T = 20 when the whole map fits one tile -- the 20x20 level of a 640 input: four
// 16-tiles would decode 2.9x the pixels -- else 16). The (T+8)^2 input halo is decoded once into LDS (-inf outside the image, so
// the scans need no bounds checks); the argmax is found separably (row pass: first kx per input row, column pass: first ky whose
// row maximum is the window maximum -- identical to the row-major scan); phase 3 gathers per pixel in fixed (wy, wx) order,
// so the result is deterministic.
template <int DT, int T>
__global__ __launch_bounds__(256) void pool5_bwd_kernel(const uint16_t* __restrict__ buf, uint16_t* __restrict__ dbuf, int ld, int coff_in,
                                                        int coff_out, int N, int H, int W, int CV, int tiles_x, int tiles_y) {
    constexpr int TI = T + 8, TW = T + 4;  // input halo / window-centre extent
    __shared__ float s_in[TI * TI][8];
    __shared__ float s_rmax[TI * TW][8];
    __shared__ unsigned s_rarg[TI * TW];  // first kx of the row maximum, 4 bits per channel
    __shared__ u32x4 s_g[TW * TW];
    __shared__ unsigned long long s_idx[TW * TW];
    // Consecutive block ids go to consecutive XCDs, each with an L2 of its own. The eight channel vectors that share a 128-byte line of a
    // pixel row (the 16-byte pieces of neighbouring blocks) are therefore given to blocks of ONE XCD: logical id = position inside the XCD's
    // contiguous share of the grid (the mapping of wgrad_gemm_kernel) -- with the plain id every XCD fetched every line (197 MB per launch
    // for 16 MB of tensors, profiles/r04_pmc_traffic.json).
    int b;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7, j = blockIdx.x >> 3;
        b = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + j;
    }
    const int c = (b % CV) * 8;
    b /= CV;
    const int x0 = (b % tiles_x) * T;
    b /= tiles_x;
    const int y0 = (b % tiles_y) * T;
    const int n = b / tiles_y;
    const int tid = threadIdx.x;
    const int64_t img = (int64_t)n * H * W;
    for (int i = tid; i < TI * TI; i += 256) {
        const int yy = y0 - 4 + i / TI, xx = x0 - 4 + i % TI;
        Vec8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] = -INFINITY;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = load8<DT>(buf + (img + (int64_t)yy * W + xx) * ld + coff_in + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) s_in[i][k] = v.v[k];
    }
    for (int i = tid; i < TW * TW; i += 256) {
        const int py = y0 - 2 + i / TW, px = x0 - 2 + i % TW;
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W)
            v = *reinterpret_cast<const u32x4*>(dbuf + (img + (int64_t)py * W + px) * ld + coff_out + c);
        s_g[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < TI * TW; i += 256) {  // row pass: input row r, window column wx -> max / first kx over the 5 columns
        const int r = i / TW, wx = i % TW;
        float best[8];
        unsigned arg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            best[k] = -INFINITY;
            arg[k] = 15u;
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float* e = s_in[r * TI + wx + kx];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e[k] > best[k]) {
                    best[k] = e[k];
                    arg[k] = (unsigned)kx;
                }
        }
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s_rmax[i][k] = best[k];
            packed |= arg[k] << (4 * k);
        }
        s_rarg[i] = packed;
    }
    __syncthreads();
    for (int i = tid; i < TW * TW; i += 256) {  // column pass
        const int wy = i / TW, wx = i % TW;
        const int py = y0 - 2 + wy, px = x0 - 2 + wx;
        unsigned long long packed = ~0ull;  // windows centred outside the image never match
        if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {
            float best[8];
            unsigned arg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                best[k] = -INFINITY;
                arg[k] = 255u;
            }
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int ri = (wy + ky) * TW + wx;
                const float* e = s_rmax[ri];
                const unsigned ra = s_rarg[ri];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (e[k] > best[k]) {
                        best[k] = e[k];
                        arg[k] = (unsigned)(ky * 5) + ((ra >> (4 * k)) & 0xfu);
                    }
            }
            packed = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) packed |= (unsigned long long)(arg[k] & 0xffu) << (8 * k);
        }
        s_idx[i] = packed;
    }
    __syncthreads();
    for (int i = tid; i < T * T; i += 256) {
        const int qy = i / T, qx = i % T;
        const int y = y0 + qy, x = x0 + qx;
        if (y >= H || x >= W) continue;
        uint16_t* dq = dbuf + (img + (int64_t)y * W + x) * ld + coff_in + c;
        Vec8 acc = load8<DT>(dq);
#pragma unroll
        for (int wy = 0; wy < 5; ++wy)
#pragma unroll
            for (int wx = 0; wx < 5; ++wx) {
                // window centre p = q + (wy-2, wx-2); q sits at (4-wy, 4-wx) inside it
                const int wi = (qy + wy) * TW + qx + wx;
                const unsigned long long idx = s_idx[wi];
                const unsigned long long want = 0x0101010101010101ull * (unsigned long long)((4 - wy) * 5 + (4 - wx));
                const unsigned long long diff = idx ^ want;
                if (!((diff - 0x0101010101010101ull) & ~diff & 0x8080808080808080ull)) continue;  // no zero byte -> no channel matches
                const Vec8 g = load8<DT>(reinterpret_cast<const uint16_t*>(&s_g[wi]));
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((((unsigned)(diff >> (8 * k))) & 0xffu) == 0u) acc.v[k] += g.v[k];
            }
        store8<DT>(dq, acc);
    }
}


// per-channel sum over rows of a [M, C] slice (bias gradient of the head's 1x1 projections): block partials, then
// bn_bwd_sums_kernel-style fixed-order finish.
template <int DT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const uint16_t* __restrict__ src, int ld, int coff, float* __restrict__ part, int64_t M,
                                                             int C, int CV) {
    extern __shared__ float shm[];  // [rows_per_pass][C]
    const ColMap cm = col_map(CV);
    const int c = cm.cv * 8;
    Vec8 s1;
#pragma unroll
    for (int i = 0; i < 8; ++i) s1.v[i] = 0.f;
    if (cm.active) {
        for (int64_t r = (int64_t)blockIdx.x * cm.rows_per_pass + cm.rl; r < M; r += (int64_t)gridDim.x * cm.rows_per_pass) {
            const Vec8 g = load8<DT>(src + r * ld + coff + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) s1.v[i] += g.v[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) shm[cm.rl * C + c + i] = s1.v[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < C; j += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < cm.rows_per_pass; ++r) acc += shm[r * C + j];
        part[(int64_t)blockIdx.x * C + j] = acc;
    }
}

// 16 channels x 16 partial-row groups per workgroup, fixed summation order (same shape as bn_bwd_sums_kernel)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nblk, int C, int C_out, float* out, int accumulate) {
    __shared__ double sh[16][16];
    const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    double s = 0.0;
    if (c < C_out)
        for (int b = g; b < nblk; b += 16) s += (double)part[(int64_t)b * C + c];
    sh[g][cx] = s;
    __syncthreads();
    if (g == 0 && c < C_out) {
        s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sh[k][cx];
        out[c] = (accumulate ? out[c] : 0.f) + (float)s;
    }
}

// NCHW image (uint8 scaled by 1/255, or float) -> NHWC with the 3 channels padded to 8 (16-byte pixels), 16-bit: lets the stem
// run on the generic implicit-GEMM kernels (K = 72 -> 128) instead of a VALU kernel.
template <int DT>
__global__ __launch_bounds__(256) void image_to_nhwc8_kernel(const void* __restrict__ img, int img_dtype, uint16_t* __restrict__ out, int N,
                                                             int H, int W) {
    const int64_t total = (int64_t)N * H * W;
    const int64_t plane = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = p / plane, r = p - n * plane;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t i = (n * 3 + c) * plane + r;
            v[c] = img_dtype == CDET_U8 ? (float)((const uint8_t*)img)[i] * (1.0f / 255.0f) : load_elem(img, i, img_dtype);
        }
        u32x4 o = {pack2e<DT>(v[0], v[1]), pack2e<DT>(v[2], 0.f), 0u, 0u};
        *reinterpret_cast<u32x4*>(out + p * 8) = o;
    }
}

// CDET_BN_REV bit 0: bn_silu_fwd walks rows back to front, bit 1: bn_silu_bwd_apply does
static int bn_rev() {
    static int v = -1;
    if (v < 0) {
        v = tune_env("CDET_BN_REV", 0);
    }
    return v;
}

static int bn_v2() {
    static int v = -1;
    if (v < 0) v = tune_env("CDET_BN_V2", 1);
    return v;
}
// every tensor of the call below 2 GiB: the lean kernels address with 32-bit byte offsets
static inline bool bn_small(int64_t M, int ld) { return M * (int64_t)ld * 2 < (1ll << 31) - 65536; }

static inline int grid_for(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > 8192 ? 8192 : b);
}

}  // namespace cdet

using namespace cdet;

#define DISPATCH16(dtype, CALL)                                        \
    do {                                                               \
        if ((dtype) == CDET_BF16) { constexpr int DT = CDET_BF16; CALL; } \
        else { constexpr int DT = CDET_F16; CALL; }                    \
    } while (0)

static int check16(const char* fn, int dtype, int C, int a, int b, int c, int d) {
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "%s: dtype must be bf16/f16", fn);
    CDET_CHECK_ARG(C > 0 && C % 8 == 0 && a % 8 == 0 && b % 8 == 0 && c % 8 == 0 && d % 8 == 0,
                   "%s: channels / ld / coff must be multiples of 8 (C=%d)", fn, C);
    return 0;
}

extern "C" int cdet_bn_finalize(const float* stats, int32_t nblk, int32_t C, int64_t count, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* invstd, void* stream) {
    CDET_CHECK_ARG(stats && mean && invstd && nblk > 0 && C > 0 && count > 0, "cdet_bn_finalize: bad arguments");
    const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(div_up(C, 16)), dim3(BN_RED_THREADS), 0, (hipStream_t)stream, stats, nblk, C, 1.0 / (double)count,
                       unbias, eps, momentum, running_mean, running_var, mean, invstd);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_bn_silu_fwd(const void* z, int32_t z_ld, int32_t z_coff, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, const void* residual, int32_t res_ld, int32_t res_coff, void* y, int32_t y_ld,
                                int32_t y_coff, int64_t M, int32_t C, int32_t dtype, void* stream) {
    if (int e = check16("cdet_bn_silu_fwd", dtype, C, z_ld, z_coff, y_ld, y_coff)) return e;
    CDET_CHECK_ARG(C / 8 <= 256, "cdet_bn_silu_fwd: C too large");
    const int CV = C / 8, rpp = 256 / CV;
    const int grid = grid_for(M, rpp * 8);
    if (bn_v2() && bn_small(M, z_ld) && bn_small(M, y_ld) && (!residual || bn_small(M, res_ld))) {
        DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_fwd2_kernel<DT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)z, z_ld,
                                             z_coff, mean, invstd, gamma, beta, (const uint16_t*)residual, res_ld, res_coff, (uint16_t*)y, y_ld,
                                             y_coff, M, CV, bn_rev() & 1));
        CDET_LAUNCH_CHECK();
        return 0;
    }
    DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_fwd_kernel<DT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)z, z_ld,
                                         z_coff, mean, invstd, gamma, beta, (const uint16_t*)residual, res_ld, res_coff, (uint16_t*)y, y_ld,
                                         y_coff, M, CV, bn_rev() & 1));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_bn_bwd_blocks(int64_t M) {
    static int div = -1, cap = -1;
    if (div < 0) {
        div = tune_env("CDET_BN_BWD_DIV", 64);  // measured per task pass (reduce + sums ms): 32/1024 3.16, 64/1024 3.15, 64/512 2.78, 128/512 2.90, 100/384 3.16
        cap = tune_env("CDET_BN_BWD_CAP", 512);
    }
    int64_t b = (M + div - 1) / div;  // >= `div` pixel rows per block, at most cap / 256 blocks per CU
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

static int bn_silu_bwd_reduce_impl(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, float* part, int64_t M, int32_t C, int32_t dtype,
                                   const cdet_bn_fold* fold, void* stream);

extern "C" int cdet_bn_silu_bwd_reduce(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                       const float* mean, const float* invstd, const float* gamma, const float* beta, float* part, int64_t M,
                                       int32_t C, int32_t dtype, void* stream) {
    return bn_silu_bwd_reduce_impl(dy, dy_ld, dy_coff, z, z_ld, z_coff, mean, invstd, gamma, beta, part, M, C, dtype, nullptr, stream);
}

extern "C" int cdet_bn_silu_bwd_reduce_fold(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                            const float* mean, const float* invstd, const float* gamma, const float* beta, float* part, int64_t M,
                                            int32_t C, int32_t dtype, const cdet_bn_fold* fold_dev, void* stream) {
    CDET_CHECK_ARG(fold_dev, "cdet_bn_silu_bwd_reduce_fold: null descriptor");
#ifndef CDET_EXPERIMENTS
    CDET_CHECK_ARG(false, "cdet_bn_silu_bwd_reduce_fold: the in-launch BatchNorm fold is compiled into -DCDET_EXPERIMENTS builds only (cdet_has_experiments())");
#endif
    CDET_CHECK_ARG(bn_v2() && bn_small(M, dy_ld) && bn_small(M, z_ld), "cdet_bn_silu_bwd_reduce_fold: tensors of 2 GiB and more take the two-launch form");
    return bn_silu_bwd_reduce_impl(dy, dy_ld, dy_coff, z, z_ld, z_coff, mean, invstd, gamma, beta, part, M, C, dtype, fold_dev, stream);
}

extern "C" int64_t cdet_bn_fold_cl_doubles(int32_t nrows, int32_t C) {
    if (nrows <= 0 || C <= 0 || nrows > BNF_CL * BNF_MAX_CL) return -1;
    return (int64_t)((nrows + BNF_CL - 1) / BNF_CL) * 2 * C;
}

extern "C" int cdet_bn_running_update(const cdet_bn_running_item* items_dev, int32_t n_items, int32_t max_C, void* stream) {
    CDET_CHECK_ARG(items_dev && n_items > 0 && max_C > 0, "cdet_bn_running_update: bad arguments");
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    CDET_LAUNCH_CHECK();
    return 0;
}

static int bn_silu_bwd_reduce_impl(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, float* part, int64_t M, int32_t C, int32_t dtype,
                                   const cdet_bn_fold* fold, void* stream) {
    if (int e = check16("cdet_bn_silu_bwd_reduce", dtype, C, dy_ld, dy_coff, z_ld, z_coff)) return e;
    CDET_CHECK_ARG(C / 8 <= 256, "cdet_bn_silu_bwd_reduce: C too large");
    const int CV = C / 8, rpp = 256 / CV;
    const size_t shm = (size_t)rpp * 2 * C * sizeof(float);
    CDET_CHECK_ARG(shm <= 64 * 1024, "cdet_bn_silu_bwd_reduce: LDS budget exceeded");
    if (bn_v2() && bn_small(M, dy_ld) && bn_small(M, z_ld)) {
        DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_reduce2_kernel<DT>), dim3(cdet_bn_bwd_blocks(M)), dim3(256), shm, (hipStream_t)stream,
                                             (const uint16_t*)dy, dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, part,
                                             M, C, CV, fold));
        CDET_LAUNCH_CHECK();
        return 0;
    }
    DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_reduce_kernel<DT>), dim3(cdet_bn_bwd_blocks(M)), dim3(256), shm, (hipStream_t)stream,
                                         (const uint16_t*)dy, dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, part,
                                         M, C, CV));
    CDET_LAUNCH_CHECK();
    return 0;
}

static int bn_silu_bwd_apply_impl(const char* fn, const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta, const float* part, int32_t nblk,
                                  float* dgamma, float* dbeta, int32_t accumulate, void* dz, int32_t dz_ld, int32_t dz_coff, int64_t M, int32_t C,
                                  int32_t dtype, int64_t count, void* also, int32_t also_ld, int32_t also_coff, void* stream) {
    if (int e = check16(fn, dtype, C, dy_ld, dy_coff, z_ld, z_coff)) return e;
    CDET_CHECK_ARG(dz_ld % 8 == 0 && dz_coff % 8 == 0 && part && nblk >= 0, "%s: bad arguments", fn);
    CDET_CHECK_ARG(also_ld % 8 == 0 && also_coff % 8 == 0, "%s: the second gradient slice needs ld / coff in multiples of 8", fn);
    // nblk > 0: reduce the partials first, the sums live behind them: part[nblk*2*C .. nblk*2*C + 2*C)
    // nblk == 0 (SyncBatchNorm): `part` already holds the (all-reduced) sums [2C]; dgamma/dbeta were produced by cdet_bn_bwd_sums
    float* sums = const_cast<float*>(part) + (int64_t)nblk * 2 * C;
    if (nblk > 0) {
        hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(div_up(C, 16)), dim3(BN_RED_THREADS), 0, (hipStream_t)stream, part, nblk, C, sums, dgamma, dbeta, accumulate);
        CDET_LAUNCH_CHECK();
    }
    const int64_t cnt = count > 0 ? count : M;
    const int CV = C / 8, rpp = 256 / CV;
    const int grid = grid_for(M, rpp * 8);
    if (bn_v2() && bn_small(M, dy_ld) && bn_small(M, z_ld) && bn_small(M, dz_ld) && (!also || bn_small(M, also_ld))) {
        if (also) {
            DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_apply2_kernel<DT, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy,
                                                 dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, sums, 1.0f / (float)cnt,
                                                 (uint16_t*)dz, dz_ld, dz_coff, M, C, CV, (bn_rev() >> 1) & 1, (uint16_t*)also, also_ld, also_coff));
        } else {
            DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_apply2_kernel<DT, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy,
                                                 dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, sums, 1.0f / (float)cnt,
                                                 (uint16_t*)dz, dz_ld, dz_coff, M, C, CV, (bn_rev() >> 1) & 1, (uint16_t*)nullptr, 0, 0));
        }
        CDET_LAUNCH_CHECK();
        return 0;
    }
    if (also) {
        DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<DT, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy,
                                             dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, sums, 1.0f / (float)cnt,
                                             (uint16_t*)dz, dz_ld, dz_coff, M, C, CV, (bn_rev() >> 1) & 1, (uint16_t*)also, also_ld, also_coff));
    } else {
        DISPATCH16(dtype, hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<DT, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy,
                                             dy_ld, dy_coff, (const uint16_t*)z, z_ld, z_coff, mean, invstd, gamma, beta, sums, 1.0f / (float)cnt,
                                             (uint16_t*)dz, dz_ld, dz_coff, M, C, CV, (bn_rev() >> 1) & 1, (uint16_t*)nullptr, 0, 0));
    }
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_bn_silu_bwd_apply(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta, const float* part,
                                      int32_t nblk, float* dgamma, float* dbeta, int32_t accumulate, void* dz, int32_t dz_ld, int32_t dz_coff,
                                      int64_t M, int32_t C, int32_t dtype, int64_t count, void* stream) {
    return bn_silu_bwd_apply_impl("cdet_bn_silu_bwd_apply", dy, dy_ld, dy_coff, z, z_ld, z_coff, mean, invstd, gamma, beta, part, nblk, dgamma, dbeta,
                                  accumulate, dz, dz_ld, dz_coff, M, C, dtype, count, nullptr, 0, 0, stream);
}

extern "C" int cdet_bn_silu_bwd_apply_add(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                          const float* mean, const float* invstd, const float* gamma, const float* beta, const float* part,
                                          int32_t nblk, float* dgamma, float* dbeta, int32_t accumulate, void* dz, int32_t dz_ld, int32_t dz_coff,
                                          int64_t M, int32_t C, int32_t dtype, int64_t count, void* also, int32_t also_ld, int32_t also_coff,
                                          void* stream) {
    CDET_CHECK_ARG(also, "cdet_bn_silu_bwd_apply_add: null pointer");
    return bn_silu_bwd_apply_impl("cdet_bn_silu_bwd_apply_add", dy, dy_ld, dy_coff, z, z_ld, z_coff, mean, invstd, gamma, beta, part, nblk, dgamma,
                                  dbeta, accumulate, dz, dz_ld, dz_coff, M, C, dtype, count, also, also_ld, also_coff, stream);
}

extern "C" int cdet_copy_channels(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff, int64_t M,
                                  int32_t C, int32_t dtype, int32_t accumulate, void* stream) {
    if (int e = check16("cdet_copy_channels", dtype, C, src_ld, src_coff, dst_ld, dst_coff)) return e;
    const int CV = C / 8;
    DISPATCH16(dtype, hipLaunchKernelGGL((copy_channels_kernel<DT>), dim3(grid_for(M * CV, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                                         (const uint16_t*)src, src_ld, src_coff, (uint16_t*)dst, dst_ld, dst_coff, M, CV, accumulate));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_add_channels(const void* a, int32_t a_ld, int32_t a_coff, const void* b, int32_t b_ld, int32_t b_coff, void* y, int32_t y_ld,
                                 int32_t y_coff, int64_t M, int32_t C, int32_t dtype, void* stream) {
    if (int e = check16("cdet_add_channels", dtype, C, a_ld, a_coff, b_ld, b_coff)) return e;
    CDET_CHECK_ARG(y_ld % 8 == 0 && y_coff % 8 == 0, "cdet_add_channels: y ld/coff must be multiples of 8");
    const int CV = C / 8;
    DISPATCH16(dtype, hipLaunchKernelGGL((add_channels_kernel<DT>), dim3(grid_for(M * CV, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                                         (const uint16_t*)a, a_ld, a_coff, (const uint16_t*)b, b_ld, b_coff, (uint16_t*)y, y_ld, y_coff, M, CV));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_upsample2(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff, int32_t N,
                              int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (int e = check16("cdet_upsample2", dtype, C, src_ld, src_coff, dst_ld, dst_coff)) return e;
    const int CV = C / 8;
    DISPATCH16(dtype, hipLaunchKernelGGL((upsample2_kernel<DT>), dim3(grid_for((int64_t)N * 4 * H * W * CV, 256 * 4)), dim3(256), 0,
                                         (hipStream_t)stream, (const uint16_t*)src, src_ld, src_coff, (uint16_t*)dst, dst_ld, dst_coff, N, H, W, CV));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_upsample2_bwd(const void* ddst, int32_t ddst_ld, int32_t ddst_coff, void* dsrc, int32_t dsrc_ld, int32_t dsrc_coff,
                                  int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate, void* stream) {
    if (int e = check16("cdet_upsample2_bwd", dtype, C, ddst_ld, ddst_coff, dsrc_ld, dsrc_coff)) return e;
    const int CV = C / 8;
    DISPATCH16(dtype, hipLaunchKernelGGL((upsample2_bwd_kernel<DT>), dim3(grid_for((int64_t)N * H * W * CV, 256 * 2)), dim3(256), 0,
                                         (hipStream_t)stream, (const uint16_t*)ddst, ddst_ld, ddst_coff, (uint16_t*)dsrc, dsrc_ld, dsrc_coff, N, H,
                                         W, CV, accumulate));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_sppf_pool(void* buf, int32_t ld, int32_t coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (int e = check16("cdet_sppf_pool", dtype, C, ld, coff, 0, 0)) return e;
    const int CV = C / 8;
    if ((int64_t)H * W * 2 * 32 * 2 <= 60 * 1024 && tune_env("CDET_SPPF_FUSED", 1)) {  // the plane of 16 channels twice in LDS, as fp32: one launch for the chain
        const size_t lds = (size_t)H * W * 2 * 32 * 2;
        DISPATCH16(dtype, hipLaunchKernelGGL((sppf_pool3_kernel<DT>), dim3(N * div_up(CV, 2)), dim3(256), lds, (hipStream_t)stream, (uint16_t*)buf, ld, coff,
                                             C, H, W, CV));
        CDET_LAUNCH_CHECK();
        return 0;
    }
    for (int i = 0; i < 3; ++i) {
        DISPATCH16(dtype, hipLaunchKernelGGL((pool5_kernel<DT>), dim3(grid_for((int64_t)N * H * W * CV, 256)), dim3(256), 0, (hipStream_t)stream,
                                             (uint16_t*)buf, ld, coff + i * C, coff + (i + 1) * C, N, H, W, CV));
        CDET_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int cdet_sppf_pool_bwd(const void* buf, void* dbuf, int32_t ld, int32_t coff, int32_t N, int32_t H, int32_t W, int32_t C,
                                  int32_t dtype, void* stream) {
    if (int e = check16("cdet_sppf_pool_bwd", dtype, C, ld, coff, 0, 0)) return e;
    const bool t20 = H > 16 && H <= 20 && W > 16 && W <= 20;  // the whole map as one tile
    const int T = t20 ? 20 : 16;
    const int CV = C / 8, tx = div_up(W, T), ty = div_up(H, T);
    CDET_CHECK_ARG((int64_t)N * ty * tx * CV < (1ll << 31), "cdet_sppf_pool_bwd: grid too large");
    for (int i = 2; i >= 0; --i) {
        const dim3 grid((unsigned)((int64_t)N * ty * tx * CV));
        if (t20) {
            DISPATCH16(dtype, hipLaunchKernelGGL((pool5_bwd_kernel<DT, 20>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)buf,
                                                 (uint16_t*)dbuf, ld, coff + i * C, coff + (i + 1) * C, N, H, W, CV, tx, ty));
        } else {
            DISPATCH16(dtype, hipLaunchKernelGGL((pool5_bwd_kernel<DT, 16>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)buf,
                                                 (uint16_t*)dbuf, ld, coff + i * C, coff + (i + 1) * C, N, H, W, CV, tx, ty));
        }
        CDET_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int cdet_colsum(const void* src, int32_t ld, int32_t coff, int64_t M, int32_t C, int32_t C_out, int32_t dtype, float* out,
                           int32_t accumulate, float* part, void* stream) {
    if (int e = check16("cdet_colsum", dtype, C, ld, coff, 0, 0)) return e;
    CDET_CHECK_ARG(out && part && C_out <= C && C / 8 <= 256, "cdet_colsum: bad arguments");
    const int CV = C / 8, rpp = 256 / CV;
    const int nblk = cdet_bn_bwd_blocks(M);
    const size_t shm = (size_t)rpp * C * sizeof(float);
    DISPATCH16(dtype, hipLaunchKernelGGL((colsum_partial_kernel<DT>), dim3(nblk), dim3(256), shm, (hipStream_t)stream, (const uint16_t*)src, ld, coff,
                                         part, M, C, CV));
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(div_up(C_out, 16)), dim3(256), 0, (hipStream_t)stream, part, nblk, C, C_out, out, accumulate);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_image_to_nhwc8(const void* img_nchw, int32_t img_dtype, void* out, int32_t N, int32_t H, int32_t W, int32_t dtype,
                                   void* stream) {
    CDET_CHECK_ARG(img_nchw && out, "cdet_image_to_nhwc8: null pointer");
    CDET_CHECK_ARG(dtype == CDET_BF16 || dtype == CDET_F16, "cdet_image_to_nhwc8: dtype must be bf16/f16");
    const int grid = grid_for((int64_t)N * H * W, 256 * 4);
    DISPATCH16(dtype, hipLaunchKernelGGL((image_to_nhwc8_kernel<DT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img_nchw, img_dtype,
                                         (uint16_t*)out, N, H, W));
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_bn_bwd_sums(const float* part, int32_t nblk, int32_t C, float* sums, float* dgamma, float* dbeta, int32_t accumulate,
                                void* stream) {
    CDET_CHECK_ARG(part && sums && nblk > 0 && C > 0, "cdet_bn_bwd_sums: bad arguments");
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(div_up(C, 16)), dim3(BN_RED_THREADS), 0, (hipStream_t)stream, part, nblk, C, sums, dgamma, dbeta, accumulate);
    CDET_LAUNCH_CHECK();
    return 0;
}
