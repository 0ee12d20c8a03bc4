// BatchNorm partial sums reduced INSIDE the launch that produces them (round 5): the per-layer bn_finalize / bn_bwd_sums launches of the
// training plans (reference models/common.py:57-62: nn.BatchNorm2d in train mode; 245 + 194 launches per iteration of the 2-task YOLOv8x, each a
// handful of workgroups on the dependent chain between a convolution and its normalisation pass) fold into the kernels that write the partials.
//
// Every producing workgroup owns one partial row (a column range of it: [c0, c0 + ncols) of row `row`). The rows form a fixed tree:
//   level 1  clusters of BNF_CL = 32 consecutive rows: the workgroup that draws the cluster's last ticket adds the cluster's rows, ascending, in
//            double, and publishes the cluster sum;
//   level 2  the workgroup that draws the last cluster ticket adds the cluster sums, ascending, in double, rounds the totals to fp32 (the vector
//            SyncBatchNorm would all-reduce) and finishes like bn_finalize_kernel / bn_bwd_sums_kernel.
// The ORDER of every addition is fixed by row numbers alone -- whichever workgroup happens to be last adds the same numbers in the same order --
// and the stand-alone kernels (elementwise.hip: bn_reduce_partials, tree form) use the same tree: CDET_BN_FOLD=0 and the SyncBatchNorm launch lists
// give bit-identical statistics. No atomics on the sums; the only atomics are the tickets.
//
// Visibility across CUs / XCDs follows MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility" and
// cdna_hip_programming.md (in-launch split-K reduction), the write-through form: partial rows and cluster sums are stored with sc1 (agent-scope
// write-through) stores, every wave drains vmcnt(0), the workgroup meets at a barrier, ONE lane draws the ticket with a relaxed agent-scope
// fetch_add; the reducer reads with sc1 loads (they bypass the CU's L1; the producers' sc1 stores left no line behind in any L2). No __threadfence()
// (= buffer_wbl2 of an L2 full of freshly written activations).
#pragma once
#include "common.h"

// The in-launch reduction lost its A/B (profiles/r05_bn_fold_ab.txt: -2.6 % on the training step) and is kept for the record only: its code is
// compiled into the kernels of -DCDET_EXPERIMENTS builds; in the product build CDET_FOLD() is `false`, the branches fold away and the entry
// points that take a descriptor refuse it.
#ifdef CDET_EXPERIMENTS
#define CDET_FOLD(p) ((p) != nullptr)
#else
#define CDET_FOLD(p) false
#endif

namespace cdet {

constexpr int BNF_CL = 32;           // rows per cluster
constexpr int BNF_MAX_CL = 128;      // clusters per launch (4096 rows): beyond that the stand-alone kernels run (legacy order)
constexpr int BNF_MAX_CB = 8;        // column blocks (workgroups sharing a row) with tickets of their own
constexpr int BNF_TICKET_WORDS = BNF_MAX_CB * (1 + BNF_MAX_CL);

typedef cdet_bn_fold BnFold;

typedef __attribute__((ext_vector_type(2))) unsigned bnf_u2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bnf_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0x7fffffff, 0x00020000);
}
// aux = 16: sc1 (agent scope) -- loads bypass the L1, stores write through
__device__ __forceinline__ float bnf_ldf(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ double bnf_ldd(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void bnf_stf(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void bnf_std(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bnf_u2, v), r, (int)byte_off, 0, 16);
}

// Sum of n (<= BNF_CL) values `stride` bytes apart, ascending, in double: loads in two batches of 16 (all of a batch issued before its first add).
__device__ __forceinline__ double bnf_sum_rows_f(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned stride, int n) {
    double acc = 0.0;
#pragma unroll
    for (int h = 0; h < BNF_CL; h += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = h + i < n ? bnf_ldf(r, off + (unsigned)(h + i) * stride) : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (h + i < n) acc += (double)v[i];
    }
    return acc;
}

// mean / invstd / running statistics of one channel from its fp32 totals [sum, sumsq] -- ONE definition for bn_finalize_kernel, the in-launch
// fold and the batched running-statistics update. Every multiply-add is an EXPLICIT fma whose other operands are plain products: nothing is left
// for the compiler's contraction to decide (measured: with `a * b + c * d` the same source rounded differently inside conv_halo_kernel and inside
// bn_finalize_kernel, `#pragma clang fp contract(off)` notwithstanding) -- the call sites must agree to the bit.
__device__ __forceinline__ void bn_stats_from_totals(float sf, float qf, double inv_count, double unbias, float eps, float momentum, float* mean,
                                                     float* invstd, float* running_mean, float* running_var) {
    const double m = (double)sf * inv_count;
    const double mm = m * m;
    double var = fma((double)qf, inv_count, -mm);
    if (var < 0.0) var = 0.0;
    if (mean != nullptr) *mean = (float)m;
    if (invstd != nullptr) *invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
        const float keep = 1.f - momentum;
        const float km = keep * *running_mean;
        *running_mean = fmaf(momentum, (float)m, km);
        const float kv = keep * *running_var;
        const double vu = var * unbias;
        *running_var = fmaf(momentum, (float)vu, kv);
    }
}

// Called by ALL threads of the workgroup (uniformly) after the workgroup's share of its partial row has been stored with bnf_stf.
//   part   [nrows][2][C] fp32 partial rows of the launch
//   row    this workgroup's row; [c0, c0 + ncols) its column range; cb its column-block index (< BNF_MAX_CB)
//   flag   one LDS word nothing else uses any more
// FWD: totals -> mean / invstd (+ running statistics); else (backward) totals -> sums [2C] and dgamma / dbeta.
//   live   threads that take part (default: the whole workgroup; conv_halo_kernel's K-split form arrives here with its first 256 threads only)
template <bool FWD>
__device__ __forceinline__ void bn_fold_finish(const BnFold* __restrict__ fp, const float* __restrict__ part, int row, int c0, int ncols, int cb,
                                               volatile int* flag, int live = 0) {
    const int tid = threadIdx.x, nt = live ? live : (int)blockDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the row has left the CU (sc1 stores)
    __syncthreads();
    const int nrows = fp->nrows, C = fp->C, ncl = fp->ncl;
    unsigned* tk = fp->tickets + (size_t)cb * (1 + BNF_MAX_CL);
    const int cl = row / BNF_CL, r0 = cl * BNF_CL;
    const int n_in = min(BNF_CL, nrows - r0);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(tk + 1 + cl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == (unsigned)(n_in - 1) ? 1 : 0;
    }
    __syncthreads();
    const int last_in_cluster = *flag;
    __syncthreads();
    if (!last_in_cluster) return;
    // ---- level 1: this cluster's rows, ascending, in double
    const __amdgpu_buffer_rsrc_t rp = bnf_rsrc(part), rc = bnf_rsrc(fp->cl_sums);
    for (int j = tid; j < 2 * ncols; j += nt) {
        const int which = j >= ncols ? 1 : 0, col = c0 + (which ? j - ncols : j);
        if (col < C) {
            const double acc = bnf_sum_rows_f(rp, (unsigned)((((int64_t)r0 * 2 + which) * C + col) * 4), (unsigned)(2 * C * 4), n_in);
            bnf_std(rc, (unsigned)((((int64_t)cl * 2 + which) * C + col) * 8), acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(tk + 1 + cl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the launches of a plan run one after the other: ready for the next)
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == (unsigned)(ncl - 1) ? 1 : 0;
    }
    __syncthreads();
    const int last_cluster = *flag;
    __syncthreads();
    if (!last_cluster) return;
    // ---- level 2: the cluster sums, ascending, in double; then what bn_finalize_kernel / bn_bwd_sums_kernel do
    for (int j = tid; j < ncols; j += nt) {
        const int col = c0 + j;
        if (col >= C) continue;
        double s = 0.0, q = 0.0;
        for (int k0 = 0; k0 < ncl; k0 += 8) {
            double vs[8], vq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool in = k0 + i < ncl;
                vs[i] = in ? bnf_ldd(rc, (unsigned)((((int64_t)(k0 + i) * 2 + 0) * C + col) * 8)) : 0.0;
                vq[i] = in ? bnf_ldd(rc, (unsigned)((((int64_t)(k0 + i) * 2 + 1) * C + col) * 8)) : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k0 + i < ncl) {
                    s += vs[i];
                    q += vq[i];
                }
        }
        const float sf = (float)s, qf = (float)q;  // the fp32 totals: exactly what the SyncBatchNorm list all-reduces (bn_finalize_kernel)
        if (fp->totals != nullptr) {
            fp->totals[col] = sf;
            fp->totals[C + col] = qf;
        }
        if (FWD) {
            bn_stats_from_totals(sf, qf, fp->inv_count, fp->unbias, fp->eps, fp->momentum, fp->mean + col, fp->invstd + col,
                                 fp->running_mean != nullptr ? fp->running_mean + col : nullptr,
                                 fp->running_mean != nullptr ? fp->running_var + col : nullptr);
        } else {
            if (fp->dbeta != nullptr) fp->dbeta[col] = (fp->accumulate ? fp->dbeta[col] : 0.f) + sf;
            if (fp->dgamma != nullptr) fp->dgamma[col] = (fp->accumulate ? fp->dgamma[col] : 0.f) + qf;
        }
    }
    if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace cdet
