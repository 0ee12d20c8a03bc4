// Fused multi-tensor optimizer step for gfx950 (trainers/averaging.py:205-223 + utils/torch_utils.py:302-312):
// global-norm clip (over ALL gradients, before the per-block division), per-block division by the number of tasks the
// block serves, SGD with Nesterov momentum and weight decay, gradient zeroing and the EMA lerp -- two launches instead of
// ~1300 tiny kernels and python loops over ~400 parameter / 884 state tensors.
#include "common.h"

namespace cdet {

constexpr int OPT_BLOCKS_PER_SLOT = 32;   // partial sums per slot of the norm kernel (fixed: it sizes the caller's buffer)
constexpr int OPT_SGD_BLOCKS = 256;       // update kernel: measured 1.18 / 1.08 / 1.01 / 0.98 / 0.99 ms at 32 / 64 / 128 / 256 / 512 blocks per slot

// Block b of a slot sums the contiguous chunk [b n / B, (b + 1) n / B) of the gradient: 16-byte loads over the aligned middle of the chunk (the bucket
// slices behind an odd-sized bias vector start unaligned), four independent partial sums per lane. (The strided scalar form ran at 1.5 TB/s: 0.28 ms of
// the iteration's serial tail for 420 MB.)
// `scaler` (GradScaler state, NULL = no loss scaling): the norm is that of the UNSCALED gradients g / scale -- what the reference's
// scaler.unscale_() leaves in p.grad before clip_grad_norm_ (trainers/averaging.py:207-208) -- so a large loss scale cannot overflow the sum.
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const cdet_param_slot* __restrict__ slots, float* __restrict__ out, const float* __restrict__ scaler) {
    __shared__ float sh[4];
    const cdet_param_slot sl = slots[blockIdx.y];
    const float inv = scaler ? 1.f / scaler[0] : 1.f;
    float acc = 0.f;
    if (sl.g) {
        const int64_t per = (sl.n + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per;
        const int64_t hi = lo + per < sl.n ? lo + per : sl.n;
        if (lo < hi) {
            const float* g = sl.g;
            int64_t a0 = lo + ((4 - (int)((reinterpret_cast<uintptr_t>(g + lo) >> 2) & 3)) & 3);
            if (a0 > hi) a0 = hi;
            for (int64_t i = lo + threadIdx.x; i < a0; i += 256) acc += g[i] * g[i];
            // (the unscale factor is applied to the block's partial sum below: (inv g)^2 summed = inv^2 x sum g^2 up to rounding; with inv a power of
            //  two -- every GradScaler scale is -- it is exact, and with scaler == NULL the arithmetic is the unscaled kernel's, bit for bit)
            const int64_t n4 = (hi - a0) >> 2;
            const f32x4* g4 = reinterpret_cast<const f32x4*>(g + a0);
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
            for (int64_t j = threadIdx.x; j < n4; j += 256) {
                const f32x4 v = g4[j];
                a4 += v * v;
            }
            acc += (a4[0] + a4[1]) + (a4[2] + a4[3]);
            for (int64_t i = a0 + (n4 << 2) + threadIdx.x; i < hi; i += 256) acc += g[i] * g[i];
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = sh[0] + sh[1] + sh[2] + sh[3];
        out[1 + blockIdx.y * gridDim.x + blockIdx.x] = scaler ? tot * inv * inv : tot;
    }
}

__global__ __launch_bounds__(1024) void sqnorm_finish_kernel(float* out, int n) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) acc += (double)out[1 + i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

struct GroupLr {
    float v[4];
};

// GradScaler step semantics (reference trainers/averaging.py:61, 205-223: scaler.unscale_ -> clip_grad_norm_ -> scaler.step -> scaler.update ->
// zero_grad -> ema.update): when the gradient norm is not finite (an inf / NaN anywhere in the gradients) the optimizer step is SKIPPED -- weights
// and momentum buffers keep their bits, the gradients are zeroed, and the EMA lerp still runs (the reference calls ema.update regardless). The norm
// is the found-inf flag: inf^2 = inf and NaN propagate through the fixed-order sum of cdet_grad_sqnorm.
__global__ __launch_bounds__(256) void sgd_ema_kernel(const cdet_param_slot* __restrict__ slots, const float* __restrict__ sqnorm, float max_norm,
                                                      GroupLr lrs, float momentum, float ema_decay, const float* __restrict__ scaler,
                                                      float* __restrict__ skip_count) {
    const cdet_param_slot sl = slots[blockIdx.y];
    const float lr = lrs.v[sl.group & 3];
    float coef = 1.f;
    bool skip = false;
    if (sqnorm) {
        const float sq = sqnorm[0];
        skip = !(sq <= 3.402823466e38f);                // inf or NaN: found_inf
        const float total = sqrtf(sq);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    }
    if (scaler) coef *= 1.f / scaler[0];                // scaler.unscale_ (a power of two: exact)
    if (skip_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {  // plans without loss scaling: the skip bookkeeping rides on this launch
        skip_count[1] = skip ? 1.f : 0.f;                                        // (nothing in the launch reads these two words)
        if (skip) skip_count[0] += 1.f;
    }
    if (skip) {
        // found_inf: p and the momentum buffer are not touched; optimizer.zero_grad() and ema.update(model) still happen
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sl.n; i += (int64_t)gridDim.x * 256) {
            if (sl.g) sl.g[i] = 0.f;
            if (sl.ema) sl.ema[i] = sl.ema[i] * ema_decay + (1.f - ema_decay) * sl.p[i];
        }
        return;
    }
    // 16-byte path when the slot's four arrays allow it (conv weights: 80 % of the bytes); scalar otherwise (odd-sized bias vectors
    // shift the alignment of the bucket slices behind them)
    const uintptr_t al = (uintptr_t)sl.p | (uintptr_t)sl.g | (uintptr_t)sl.mom | (uintptr_t)sl.ema;
    if ((al & 15) == 0 && (sl.n & 3) == 0 && sl.n >= 1024) {
        const int64_t n4 = sl.n >> 2;
        f32x4* P = reinterpret_cast<f32x4*>(sl.p);
        f32x4* G = reinterpret_cast<f32x4*>(sl.g);
        f32x4* Mo = reinterpret_cast<f32x4*>(sl.mom);
        f32x4* E = reinterpret_cast<f32x4*>(sl.ema);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            f32x4 p = P[i];
            if (G) {
                f32x4 g = G[i] * coef * sl.inv_div;
                if (sl.weight_decay != 0.f) g += sl.weight_decay * p;
                const f32x4 buf = sl.first_step ? g : momentum * Mo[i] + g;
                Mo[i] = buf;
                g += momentum * buf;
                p -= lr * g;
                P[i] = p;
                G[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (E) E[i] = E[i] * ema_decay + (1.f - ema_decay) * p;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sl.n; i += (int64_t)gridDim.x * 256) {
        float p = sl.p[i];
        if (sl.g) {
            float g = sl.g[i] * coef * sl.inv_div;
            if (sl.weight_decay != 0.f) g += sl.weight_decay * p;
            float buf = sl.first_step ? g : momentum * sl.mom[i] + g;
            sl.mom[i] = buf;
            g += momentum * buf;  // nesterov
            p -= lr * g;
            sl.p[i] = p;
            sl.g[i] = 0.f;        // optimizer.zero_grad()
        }
        if (sl.ema) sl.ema[i] = sl.ema[i] * ema_decay + (1.f - ema_decay) * p;
    }
}

// scaler.update() (torch.cuda.amp.GradScaler, reference trainers/averaging.py:61, 220): state = {scale, growth_tracker, skipped steps, found_inf of
// this step}. found_inf: scale *= backoff, tracker = 0; else tracker += 1 and, at growth_interval, scale *= growth, tracker = 0. growth_interval <= 0
// keeps the scale fixed (bf16 plans: scale 1, only the skip semantics apply). One thread; runs behind the update kernels that read the old scale.
__global__ void scaler_update_kernel(float* __restrict__ st, const float* __restrict__ sqnorm, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool found = !(sqnorm[0] <= 3.402823466e38f);
    st[3] = found ? 1.f : 0.f;
    if (found) {
        st[2] += 1.f;
        if (interval > 0) {
            st[0] *= backoff;
            st[1] = 0.f;
        }
    } else if (interval > 0) {
        const float tr = st[1] + 1.f;
        if (tr >= (float)interval) {
            st[0] *= growth;
            st[1] = 0.f;
        } else {
            st[1] = tr;
        }
    }
}

// dst += src; src = 0 -- the per-task gradient buckets of the blocks several tasks share are folded into the block's bucket in task order
__global__ __launch_bounds__(256) void accumulate_clear_kernel(float* __restrict__ dst, float* __restrict__ src, int64_t n) {
    const int64_t n4 = n >> 2;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
        f32x4 a = reinterpret_cast<const f32x4*>(dst)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[i];
        a += b;
        reinterpret_cast<f32x4*>(dst)[i] = a;
        reinterpret_cast<f32x4*>(src)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        dst[i] += src[i];
        src[i] = 0.f;
    }
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_accumulate_clear(float* dst, float* src, int64_t n, void* stream) {
    CDET_CHECK_ARG(dst && src && n > 0, "cdet_accumulate_clear: bad arguments");
    CDET_CHECK_ARG((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0, "cdet_accumulate_clear: 16-byte aligned buffers");
    const int64_t n4 = n >> 2;
    int grid = (int)((n4 + 256 * 8 - 1) / (256 * 8));
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(accumulate_clear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_grad_sqnorm(const cdet_param_slot* slots_dev, int32_t n_slots, float* out, const float* scaler, void* stream) {
    CDET_CHECK_ARG(slots_dev && out && n_slots > 0, "cdet_grad_sqnorm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(OPT_BLOCKS_PER_SLOT, n_slots), dim3(256), 0, s, slots_dev, out, scaler);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(1024), 0, s, out, n_slots * OPT_BLOCKS_PER_SLOT);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_sgd_ema_step(const cdet_param_slot* slots_dev, int32_t n_slots, const float* sqnorm, float max_norm, const float* lrs,
                                 int32_t n_groups, float momentum, float ema_decay, const float* scaler, float* skip_count, void* stream) {
    CDET_CHECK_ARG(slots_dev && n_slots > 0 && lrs && n_groups >= 1 && n_groups <= 4, "cdet_sgd_ema_step: bad arguments");
    GroupLr gl{{0.f, 0.f, 0.f, 0.f}};
    for (int i = 0; i < n_groups; ++i) gl.v[i] = lrs[i];
    hipLaunchKernelGGL(sgd_ema_kernel, dim3(tune_env("CDET_OPT_BLOCKS", OPT_SGD_BLOCKS), n_slots), dim3(256), 0, (hipStream_t)stream, slots_dev, sqnorm,
                       max_norm, gl, momentum, ema_decay, scaler, skip_count);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_scaler_update(float* scaler, const float* sqnorm, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream) {
    CDET_CHECK_ARG(scaler && sqnorm, "cdet_scaler_update: null pointer");
    CDET_CHECK_ARG(growth_interval <= 0 || (growth_factor > 1.f && backoff_factor > 0.f && backoff_factor < 1.f), "cdet_scaler_update: growth > 1, 0 < backoff < 1");
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, sqnorm, growth_factor, backoff_factor, growth_interval);
    CDET_LAUNCH_CHECK();
    return 0;
}
