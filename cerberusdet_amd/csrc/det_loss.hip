// Detection loss for gfx950: task-aligned assignment + BCE + CIoU + DFL, forward and d/d(head maps) in one call.
// Replaces utils/loss.py:133-181 (Loss.__call__), utils/tal.py:56-178 (TaskAlignedAssigner), utils/metrics.py:373-412
// (bbox_iou CIoU) and their autograd. Nothing of size [b, n_max, topk, A] (the reference's one_hot, tal.py:150) or even
// [b, n_max, A] is ever materialised: the per-(image, GT) pass keeps its 8400 metrics in LDS and emits only its <= 10
// selected anchors.
//
//   K1 decode    : per (image, anchor): DFL softmax-expectation -> pred box (grid units + pixels)            [N*A threads]
//   K2 gt_topk   : per (image, GT) workgroup: align = sigmoid(cls)^alpha * clamp(CIoU,0)^beta over all anchors (LDS),
//                  top-k by (value desc, anchor index asc), keep those whose centre is inside the GT          [N*n_max WGs]
//   K3 resolve   : per (image, anchor): anchors claimed by >1 GT go to argmax_g CIoU over ALL padded GTs (tal.py:44-50,
//                  first maximum), final (align, overlap) of the winner, per-GT maxima via atomicMax          [N*A threads]
//   K4 norm      : per (image, anchor): norm = align*max_overlap[g]/(max_align[g]+eps); block partials of sum(norm)
//   K5 loss_grad : per (image, anchor): BCE over classes, CIoU + DFL on foreground; gradients written in place of the
//                  head maps' layout; block partials of the three loss sums
//   K6 finish    : fixed-order reduction of the partials (double) -> out_loss[5]
// Deterministic: no floating-point atomics (atomicMax on non-negative floats as ints is order-independent).
#include "common.h"

namespace cdet {

constexpr float IOU_EPS = 1e-7f;   // utils/metrics.py:373
constexpr float TAL_EPS = 1e-9f;   // utils/tal.py:58
constexpr int REG_MAX = 16;
constexpr int MAX_TOPK = 16;

struct LossArgs {
    const void* f[3];
    void* df[3];
    int h[3], w[3], a_off[4];
    float stride[3];
    int N, nc, n_max, A, f_ld, dtype, grad_dtype, topk;
    int vec;  // rows are multiples of eight elements and every map pointer is 16-byte aligned: loss_grad_kernel moves them as 16-byte vectors
    float alpha, beta, gain_box, gain_cls, gain_dfl, grad_scale;
    const float* grad_scale_dev;  // GradScaler scale on the device (NULL: none)
    const float* gt;        // [N, n_max, 5]
    // workspace
    float* pbox;            // [N, A, 4] grid units (xyxy)
    float* dist;            // [N, A, 4] decoded ltrb
    int* sel;               // [N, n_max, MAX_TOPK] selected anchors (-1 = none)
    int* cnt;               // [N, A] number of GTs that selected the anchor
    int* gsel;              // [N, A] (one of) the selecting GT(s)
    int* tgt;               // [N, A] final GT index (target_gt_idx)
    float* a_align;         // [N, A]
    float* a_over;          // [N, A]
    float* norm;            // [N, A]
    unsigned int* g_align;  // [N, n_max] float bits
    unsigned int* g_over;   // [N, n_max]
    float* part;            // [nblk, 4] partial sums (norm | box, cls, dfl)
    double* tss;            // [1]
    // optional outputs
    float* out_loss;
    uint8_t* o_fg;
    int* o_gt_idx;
    int* o_labels;
    float* o_bboxes;
    float* o_scores;
};

__device__ __forceinline__ void anchor_of(const LossArgs& a, int an, int& lvl, int& la, float& ax, float& ay) {
    lvl = an >= a.a_off[2] ? 2 : (an >= a.a_off[1] ? 1 : 0);
    la = an - a.a_off[lvl];
    ax = (float)(la % a.w[lvl]) + 0.5f;
    ay = (float)(la / a.w[lvl]) + 0.5f;
}
__device__ __forceinline__ int64_t feat_base(const LossArgs& a, int n, int lvl, int la) {
    return ((int64_t)n * a.h[lvl] * a.w[lvl] + la) * a.f_ld;
}

// CIoU of box1 (b) vs box2 (g), utils/metrics.py:373-412 with xywh=False
__device__ __forceinline__ float ciou_f(float b1x1, float b1y1, float b1x2, float b1y2, float b2x1, float b2y1, float b2x2, float b2y2) {
    const float w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + IOU_EPS;
    const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + IOU_EPS;
    const float iw = fmaxf(fminf(b1x2, b2x2) - fmaxf(b1x1, b2x1), 0.f);
    const float ih = fmaxf(fminf(b1y2, b2y2) - fmaxf(b1y1, b2y1), 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + IOU_EPS;
    const float iou = inter / uni;
    const float cw = fmaxf(b1x2, b2x2) - fminf(b1x1, b2x1);
    const float ch = fmaxf(b1y2, b2y2) - fminf(b1y1, b2y1);
    const float c2 = cw * cw + ch * ch + IOU_EPS;
    const float sx = b2x1 + b2x2 - b1x1 - b1x2, sy = b2y1 + b2y2 - b1y1 - b1y2;
    const float rho2 = (sx * sx + sy * sy) / 4.f;
    const float dat = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = 0.40528473456935109f * dat * dat;  // 4/pi^2
    const float alpha = v / (v - iou + (1.f + IOU_EPS));
    return iou - (rho2 / c2 + v * alpha);
}

// x^e for the exponents TAL uses (alpha 0.5 -> sqrt, like ATen's pow specialisation; beta 6 -> 3 multiplies), generic otherwise
__device__ __forceinline__ float pow_sel(float x, float e) {
    if (e == 0.5f) return sqrtf(x);
    if (e == 6.f) {
        const float x2 = x * x;
        return x2 * x2 * x2;
    }
    if (e == 1.f) return x;
    return powf(x, e);
}

// ---------------------------------------------------------------------------------------------- K1
// A head-map row is f_ld floats with f_ld % 8 == 0 and a 16-byte aligned base: the row of an anchor is read as 16-byte vectors (fp32 maps) and its
// gradient leaves as 16-byte vectors of eight 16-bit values. Round 4 walked both element by element -- 88 scalar loads and 88 two-byte stores per lane,
// each touching 64 different cache lines per wave instruction: 232 us per launch for 142 MB (2.2 TB/s of useful traffic at best).
__device__ __forceinline__ void load_row16(const void* f, int64_t off, int dtype, float (&v)[16], bool vec) {
    if (vec && dtype == CDET_F32) {
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(f) + off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = p[q];
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = load_elem(f, off + b, dtype);
    }
}
// eight consecutive gradient values -> one 16-byte store (16-bit gradient maps); element-wise otherwise
__device__ __forceinline__ void store_grad8(void* df, int64_t off, int dtype, const float (&g)[8], bool vec, int n = 8) {
    if (vec && n == 8 && (dtype == CDET_BF16 || dtype == CDET_F16)) {
        u32x4 pk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lo = dtype == CDET_BF16 ? f32_to_bf16_bits(g[2 * q]) : f32_to_f16_bits(g[2 * q]);
            const uint32_t hi = dtype == CDET_BF16 ? f32_to_bf16_bits(g[2 * q + 1]) : f32_to_f16_bits(g[2 * q + 1]);
            pk[q] = lo | (hi << 16);
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(df) + off) = pk;
    } else {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < n) store_elem(df, off + b, g[b], dtype);
    }
}

__global__ __launch_bounds__(256) void loss_decode_kernel(const LossArgs a) {
    const int64_t total = (int64_t)a.N * a.A;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / a.A), an = (int)(idx - (int64_t)n * a.A);
        int lvl, la;
        float ax, ay;
        anchor_of(a, an, lvl, la, ax, ay);
        const int64_t fb = feat_base(a, n, lvl, la);
        float d[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[REG_MAX], mx = -INFINITY;
            load_row16(a.f[lvl], fb + s * REG_MAX, a.dtype, v, a.vec != 0);
#pragma unroll
            for (int b = 0; b < REG_MAX; ++b) mx = fmaxf(mx, v[b]);
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int b = 0; b < REG_MAX; ++b) {
                const float e = expf(v[b] - mx);
                den += e;
                num += e * (float)b;
            }
            d[s] = num / den;
        }
        *reinterpret_cast<f32x4*>(a.dist + idx * 4) = f32x4{d[0], d[1], d[2], d[3]};
        *reinterpret_cast<f32x4*>(a.pbox + idx * 4) = f32x4{ax - d[0], ay - d[1], ax + d[2], ay + d[3]};
        a.cnt[idx] = 0;
        a.gsel[idx] = 0;
    }
    // reset per-GT maxima
    const int64_t ng = (int64_t)a.N * a.n_max;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (int64_t)gridDim.x * blockDim.x) {
        a.g_align[i] = 0u;
        a.g_over[i] = 0u;
    }
}

// ---------------------------------------------------------------------------------------------- K2
__global__ __launch_bounds__(256) void loss_gt_topk_kernel(const LossArgs a) {
    extern __shared__ float metric[];  // [A] masked align metric; bit 31 never set (values >= 0)
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int n = blockIdx.x / a.n_max, g = blockIdx.x % a.n_max;
    const float* gt = a.gt + ((int64_t)n * a.n_max + g) * 5;
    const float gx1 = gt[1], gy1 = gt[2], gx2 = gt[3], gy2 = gt[4];
    const int gcls = (int)gt[0];
    int* sel = a.sel + ((int64_t)n * a.n_max + g) * MAX_TOPK;
    const bool valid = (gx1 + gy1 + gx2 + gy2) > 0.f;  // mask_gt, loss.py:155
    if (!valid) {  // topk indices are forced to 0 and de-duplicated away (tal.py:148-152): no positives
        if (threadIdx.x < MAX_TOPK) sel[threadIdx.x] = -1;
        return;
    }
    for (int an = threadIdx.x; an < a.A; an += blockDim.x) {
        int lvl, la;
        float ax, ay;
        anchor_of(a, an, lvl, la, ax, ay);
        const float st = a.stride[lvl];
        const float px = ax * st, py = ay * st;
        const float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
        float m = 0.f;
        if (dmin > TAL_EPS) {  // select_candidates_in_gts, tal.py:13-27 (metric is multiplied by the mask BEFORE topk)
            const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pbox + ((int64_t)n * a.A + an) * 4);
            const float ov = fmaxf(ciou_f(gx1, gy1, gx2, gy2, pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st), 0.f);
            const float logit = load_elem(a.f[lvl], feat_base(a, n, lvl, la) + 4 * REG_MAX + gcls, a.dtype);
            const float sc = 1.f / (1.f + expf(-logit));
            m = pow_sel(sc, a.alpha) * pow_sel(ov, a.beta);
        } else {
            m = -1.f;  // outside: masked metric is 0 but it can never become positive; keep it distinguishable
        }
        metric[an] = m;
    }
    __syncthreads();
    // k rounds of arg-max by (value desc, anchor index asc) over ALL anchors. Anchors outside the GT carry masked metric 0
    // in the reference (tal.py:117); its topk may pick them, and `mask_topk * mask_in_gts` (tal.py:120) removes them again.
    // They are encoded -1 here: ranked as value 0, dropped after selection.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < a.topk; ++r) {
        float bv = -2.f;
        int bi = 0x7fffffff;
        for (int an = threadIdx.x; an < a.A; an += blockDim.x) {
            float v = metric[an];
            if (v == -3.f) continue;          // already taken
            const float key = v < 0.f ? 0.f : v;  // reference value (masked metric)
            if (key > bv || (key == bv && an < bi)) {
                bv = key;
                bi = an;
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float ov = __shfl_xor(bv, m);
            const int oi = __shfl_xor(bi, m);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float fv = red_v[0];
            int fi = red_i[0];
            for (int w = 1; w < 4; ++w)
                if (red_v[w] > fv || (red_v[w] == fv && red_i[w] < fi)) {
                    fv = red_v[w];
                    fi = red_i[w];
                }
            const float mv = metric[fi];
            sel[r] = mv >= 0.f ? fi : -1;  // keep only anchors whose centre lies inside the GT
            metric[fi] = -3.f;
        }
        __syncthreads();
    }
    if (threadIdx.x >= a.topk && threadIdx.x < MAX_TOPK) sel[threadIdx.x] = -1;
    __syncthreads();
    if (threadIdx.x < a.topk) {
        const int an = sel[threadIdx.x];
        if (an >= 0) {
            atomicAdd(a.cnt + (int64_t)n * a.A + an, 1);
            atomicMax(a.gsel + (int64_t)n * a.A + an, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------- K3
__global__ __launch_bounds__(256) void loss_resolve_kernel(const LossArgs a) {
    const int64_t total = (int64_t)a.N * a.A;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / a.A), an = (int)(idx - (int64_t)n * a.A);
        const int c = a.cnt[idx];
        int g = 0;
        float al = 0.f, ov = 0.f;
        if (c > 0) {
            int lvl, la;
            float ax, ay;
            anchor_of(a, an, lvl, la, ax, ay);
            const float st = a.stride[lvl];
            const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pbox + idx * 4);
            const float bx1 = pb[0] * st, by1 = pb[1] * st, bx2 = pb[2] * st, by2 = pb[3] * st;
            g = a.gsel[idx];
            if (c > 1) {  // tal.py:44-50: argmax over ALL (padded) GTs of clamp(CIoU, 0), first maximum
                float best = -1.f;
                for (int k = 0; k < a.n_max; ++k) {
                    const float* gt = a.gt + ((int64_t)n * a.n_max + k) * 5;
                    const float o = fmaxf(ciou_f(gt[1], gt[2], gt[3], gt[4], bx1, by1, bx2, by2), 0.f);
                    if (o > best) {
                        best = o;
                        g = k;
                    }
                }
            }
            const float* gt = a.gt + ((int64_t)n * a.n_max + g) * 5;
            ov = fmaxf(ciou_f(gt[1], gt[2], gt[3], gt[4], bx1, by1, bx2, by2), 0.f);
            const float logit = load_elem(a.f[lvl], feat_base(a, n, lvl, la) + 4 * REG_MAX + (int)gt[0], a.dtype);
            al = pow_sel(1.f / (1.f + expf(-logit)), a.alpha) * pow_sel(ov, a.beta);
            atomicMax(a.g_align + (int64_t)n * a.n_max + g, __float_as_uint(al));
            atomicMax(a.g_over + (int64_t)n * a.n_max + g, __float_as_uint(ov));
        }
        a.tgt[idx] = g;
        a.a_align[idx] = al;
        a.a_over[idx] = ov;
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------- K4
__global__ __launch_bounds__(256) void loss_norm_kernel(const LossArgs a) {
    __shared__ float sh[4];
    const int64_t total = (int64_t)a.N * a.A;
    float local = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / a.A);
        float nm = 0.f;
        if (a.cnt[idx] > 0) {
            const int g = a.tgt[idx];
            const float pa = __uint_as_float(a.g_align[(int64_t)n * a.n_max + g]);
            const float po = __uint_as_float(a.g_over[(int64_t)n * a.n_max + g]);
            nm = a.a_align[idx] * po / (pa + TAL_EPS);  // tal.py:103-106
        }
        a.norm[idx] = nm;
        local += nm;
    }
    const float s = block_sum(local, sh);
    if (threadIdx.x == 0) a.part[blockIdx.x * 4 + 0] = s;
}

// Sum of column `col` of the nblk x 4 block partials in double, by ONE wave: lane l adds rows l, l + 64, ... (ascending), then a fixed butterfly over
// the lanes -- the same order on every run. (One thread walking all 512 rows took 35 - 45 us per launch on the chain between forward and backward.)
__device__ __forceinline__ double wave_col_sum(const float* __restrict__ part, int nblk, int col) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += (double)part[i * 4 + col];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
    return s;
}

__global__ __launch_bounds__(64) void loss_tss_kernel(const LossArgs a, int nblk) {
    const double s = wave_col_sum(a.part, nblk, 0);
    if (threadIdx.x == 0) a.tss[0] = s > 1.0 ? s : 1.0;  // loss.py:164
}

// ---------------------------------------------------------------------------------------------- K5
__device__ __forceinline__ float sel_gt(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }  // d max(a,b)/da
__device__ __forceinline__ float sel_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }  // d min(a,b)/da

__global__ __launch_bounds__(256) void loss_grad_kernel(const LossArgs a) {
    __shared__ float sh[4];
    const int64_t total = (int64_t)a.N * a.A;
    const float tss = (float)a.tss[0];
    // d(2*bs*total)/d(total), times the caller's scale, times the GradScaler's (scaler.scale(loss).backward(), averaging.py:158; a power of two)
    const float gmul = 2.f * (float)a.N * a.grad_scale * (a.grad_scale_dev ? a.grad_scale_dev[0] : 1.f);
    float l_box = 0.f, l_cls = 0.f, l_dfl = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / a.A), an = (int)(idx - (int64_t)n * a.A);
        int lvl, la;
        float ax, ay;
        anchor_of(a, an, lvl, la, ax, ay);
        const int64_t fb = feat_base(a, n, lvl, la);
        const bool fg = a.cnt[idx] > 0;
        const int g = a.tgt[idx];
        const float* gt = a.gt + ((int64_t)n * a.n_max + g) * 5;
        const int label = (int)gt[0];
        const float nm = a.norm[idx];
        // ---- classification: BCEWithLogits(sum) / tss  (loss.py:168)
        const bool vec = a.vec != 0;
        for (int c0 = 0; c0 < a.f_ld - 4 * REG_MAX; c0 += 8) {  // the class channels in runs of eight (the row's tail is layout padding: zero gradient)
            float xs[8], gc[8];
            const int nrun = min(8, a.f_ld - 4 * REG_MAX - c0);  // (< 8 only for a row length that is not a multiple of eight: element-wise then)
            if (vec && a.dtype == CDET_F32) {
                const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.f[lvl]) + fb + 4 * REG_MAX + c0);
                const f32x4 t0 = p[0], t1 = p[1];
                xs[0] = t0[0]; xs[1] = t0[1]; xs[2] = t0[2]; xs[3] = t0[3]; xs[4] = t1[0]; xs[5] = t1[1]; xs[6] = t1[2]; xs[7] = t1[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[j] = j < nrun ? load_elem(a.f[lvl], fb + 4 * REG_MAX + c0 + j, a.dtype) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                gc[j] = 0.f;
                if (c < a.nc) {
                    const float x = xs[j];
                    const float t = (fg && c == label) ? nm : 0.f;
                    l_cls += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                    const float s = 1.f / (1.f + expf(-x));
                    gc[j] = gmul * a.gain_cls * (s - t) / tss;
                    if (a.o_scores) a.o_scores[idx * a.nc + c] = t;
                }
            }
            if (a.df[lvl]) store_grad8(a.df[lvl], fb + 4 * REG_MAX + c0, a.grad_dtype, gc, vec, nrun);
        }
        if (a.o_fg) a.o_fg[idx] = fg ? 1 : 0;
        if (a.o_gt_idx) a.o_gt_idx[idx] = g;
        if (a.o_labels) a.o_labels[idx] = label;
        if (a.o_bboxes) *reinterpret_cast<f32x4*>(a.o_bboxes + idx * 4) = f32x4{gt[1], gt[2], gt[3], gt[4]};
        float dd[4] = {0.f, 0.f, 0.f, 0.f};  // dL/d(dist ltrb) from the box loss
        float tl[4] = {0.f, 0.f, 0.f, 0.f};
        float wgt = 0.f;
        if (fg) {
            const float st = a.stride[lvl];
            wgt = nm;  // target_scores.sum(-1): one-hot * norm
            const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pbox + idx * 4);
            const float x1 = pb[0], y1 = pb[1], x2 = pb[2], y2 = pb[3];
            const float X1 = gt[1] / st, Y1 = gt[2] / st, X2 = gt[3] / st, Y2 = gt[4] / st;  // target_bboxes /= stride, loss.py:172
            // ---- CIoU value and gradient w.r.t. (x1,y1,x2,y2); alpha is a constant (no_grad, metrics.py:405-406)
            const float w1 = x2 - x1, h1 = y2 - y1 + IOU_EPS, w2 = X2 - X1, h2 = Y2 - Y1 + IOU_EPS;
            const float iwr = fminf(x2, X2) - fmaxf(x1, X1), ihr = fminf(y2, Y2) - fmaxf(y1, Y1);
            const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
            const float inter = iw * ih;
            const float uni = w1 * h1 + w2 * h2 - inter + IOU_EPS;
            const float iou = inter / uni;
            const float cw = fmaxf(x2, X2) - fminf(x1, X1), ch = fmaxf(y2, Y2) - fminf(y1, Y1);
            const float c2 = cw * cw + ch * ch + IOU_EPS;
            const float sx = X1 + X2 - x1 - x2, sy = Y1 + Y2 - y1 - y2;
            const float rho2 = (sx * sx + sy * sy) / 4.f;
            const float dat = atanf(w2 / h2) - atanf(w1 / h1);
            const float kk = 0.40528473456935109f;
            const float v = kk * dat * dat;
            const float alpha = v / (v - iou + (1.f + IOU_EPS));
            const float ciou = iou - (rho2 / c2 + v * alpha);
            l_box += (1.f - ciou) * wgt;
            const float piw = iwr >= 0.f ? 1.f : 0.f, pih = ihr >= 0.f ? 1.f : 0.f;
            // d iw / d{x1,x2}, d ih / d{y1,y2}
            const float diw_x1 = -piw * sel_gt(x1, X1), diw_x2 = piw * sel_lt(x2, X2);
            const float dih_y1 = -pih * sel_gt(y1, Y1), dih_y2 = pih * sel_lt(y2, Y2);
            float dci[4];
            const float datan = 1.f / (w1 * w1 + h1 * h1);  // d atan(w1/h1) = (h1 dw1 - w1 dh1) / (w1^2+h1^2)
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // k: 0=x1 1=y1 2=x2 3=y2
                const float dinter = (k == 0 ? diw_x1 * ih : k == 2 ? diw_x2 * ih : k == 1 ? iw * dih_y1 : iw * dih_y2);
                const float dw1h1 = (k == 0 ? -h1 : k == 2 ? h1 : k == 1 ? -w1 : w1);
                const float duni = dw1h1 - dinter;
                const float diou = (dinter * uni - inter * duni) / (uni * uni);
                const float dcw = (k == 2 ? sel_gt(x2, X2) : k == 0 ? -sel_lt(x1, X1) : 0.f);
                const float dch = (k == 3 ? sel_gt(y2, Y2) : k == 1 ? -sel_lt(y1, Y1) : 0.f);
                const float dc2 = 2.f * cw * dcw + 2.f * ch * dch;
                const float drho2 = (k == 0 || k == 2) ? -sx * 0.5f : -sy * 0.5f;
                const float dw1 = (k == 2 ? 1.f : k == 0 ? -1.f : 0.f), dh1 = (k == 3 ? 1.f : k == 1 ? -1.f : 0.f);
                const float da1 = (h1 * dw1 - w1 * dh1) * datan;
                const float dv = 2.f * kk * dat * (-da1);
                dci[k] = diou - (drho2 * c2 - rho2 * dc2) / (c2 * c2) - alpha * dv;
            }
            const float cb = -gmul * a.gain_box * wgt / tss;  // d(gain*sum((1-ciou)*w)/tss)
            // x1 = ax - l, y1 = ay - t, x2 = ax + r, y2 = ay + b
            dd[0] = -cb * dci[0];
            dd[1] = -cb * dci[1];
            dd[2] = cb * dci[2];
            dd[3] = cb * dci[3];
            // ---- DFL targets: bbox2dist clamp(0, reg_max-1-0.01) (tal.py:208-211, BboxLoss(reg_max-1))
            tl[0] = fminf(fmaxf(ax - X1, 0.f), (float)(REG_MAX - 1) - 0.01f);
            tl[1] = fminf(fmaxf(ay - Y1, 0.f), (float)(REG_MAX - 1) - 0.01f);
            tl[2] = fminf(fmaxf(X2 - ax, 0.f), (float)(REG_MAX - 1) - 0.01f);
            tl[3] = fminf(fmaxf(Y2 - ay, 0.f), (float)(REG_MAX - 1) - 0.01f);
        }
        const f32x4 dist = *reinterpret_cast<const f32x4*>(a.dist + idx * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[REG_MAX], mx = -INFINITY;
            load_row16(a.f[lvl], fb + s * REG_MAX, a.dtype, v, vec);
#pragma unroll
            for (int b = 0; b < REG_MAX; ++b) mx = fmaxf(mx, v[b]);
            float den = 0.f;
#pragma unroll
            for (int b = 0; b < REG_MAX; ++b) {
                v[b] = expf(v[b] - mx);
                den += v[b];
            }
            const float lse = logf(den);  // log-sum-exp minus mx
            const int li = (int)tl[s];
            const float wl = (float)(li + 1) - tl[s], wr = 1.f - wl;
            if (fg) {
                // CE(logits, k) = lse - (logit_k - mx) = lse - log(v[k])
                const float ce_l = lse - logf(v[li]), ce_r = lse - logf(v[li + 1]);
                l_dfl += (ce_l * wl + ce_r * wr) * 0.25f * wgt;
            }
            if (a.df[lvl]) {
                const float cdfl = fg ? gmul * a.gain_dfl * wgt * 0.25f / tss : 0.f;
#pragma unroll
                for (int h8 = 0; h8 < REG_MAX; h8 += 8) {
                    float g8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int b = h8 + j;
                        const float p = v[b] / den;
                        float gr = dd[s] * p * ((float)b - dist[s]);                       // through the softmax expectation
                        gr += cdfl * (p - (b == li ? wl : 0.f) - (b == li + 1 ? wr : 0.f));  // DFL cross-entropies
                        g8[j] = gr;
                    }
                    store_grad8(a.df[lvl], fb + s * REG_MAX + h8, a.grad_dtype, g8, vec);
                }
            }
        }
    }
    const float sb = block_sum(l_box, sh);
    const float sc = block_sum(l_cls, sh);
    const float sd = block_sum(l_dfl, sh);
    if (threadIdx.x == 0) {
        a.part[blockIdx.x * 4 + 1] = sb;
        a.part[blockIdx.x * 4 + 2] = sc;
        a.part[blockIdx.x * 4 + 3] = sd;
    }
}

__global__ __launch_bounds__(64) void loss_finish_kernel(const LossArgs a, int nblk) {
    const double sb = wave_col_sum(a.part, nblk, 1), sc = wave_col_sum(a.part, nblk, 2), sd = wave_col_sum(a.part, nblk, 3);
    if (threadIdx.x == 0) {
        const double tss = a.tss[0];
        const float lb = (float)(sb / tss) * a.gain_box, lc = (float)(sc / tss) * a.gain_cls, ld = (float)(sd / tss) * a.gain_dfl;
        a.out_loss[0] = lb;
        a.out_loss[1] = lc;
        a.out_loss[2] = ld;
        a.out_loss[3] = lb + lc + ld;
        a.out_loss[4] = 2.f * (float)a.N * (lb + lc + ld);  // loss.py:179-181
    }
}

constexpr int LOSS_BLOCKS = 512;
static int64_t al(int64_t v) { return (v + 255) / 256 * 256; }

struct WsLayout {
    int64_t pbox, dist, sel, cnt, gsel, tgt, a_align, a_over, norm, g_align, g_over, part, tss, total;
};
static WsLayout ws_layout(int N, int A, int n_max) {
    WsLayout L;
    int64_t o = 0;
    const int64_t NA = (int64_t)N * A, NG = (int64_t)N * (n_max > 0 ? n_max : 1);
    L.pbox = o; o += al(NA * 16);
    L.dist = o; o += al(NA * 16);
    L.sel = o; o += al(NG * MAX_TOPK * 4);
    L.cnt = o; o += al(NA * 4);
    L.gsel = o; o += al(NA * 4);
    L.tgt = o; o += al(NA * 4);
    L.a_align = o; o += al(NA * 4);
    L.a_over = o; o += al(NA * 4);
    L.norm = o; o += al(NA * 4);
    L.g_align = o; o += al(NG * 4);
    L.g_over = o; o += al(NG * 4);
    L.part = o; o += al(LOSS_BLOCKS * 16);
    L.tss = o; o += 256;
    L.total = o;
    return L;
}

// Loss.preprocess (reference utils/loss.py:111-124): label rows (image index, class, normalised cx cy w h) -> gt[N][n_max][5]
// (class, x1, y1, x2, y2 in pixels). A label's slot is the number of labels of the same image in front of it (the reference's
// boolean-mask gather keeps file order). One thread per label; the label count of a batch is a few hundred.
__global__ void pad_targets_kernel(const float* __restrict__ bidx, const float* __restrict__ cls, const float* __restrict__ box, int n, int N,
                                   int n_max, float w, float h, float* __restrict__ gt, int* __restrict__ dropped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b = bidx[i];
    const int img = (int)b;
    if (img < 0 || img >= N) return;
    int pos = 0;
    for (int j = 0; j < i; ++j) pos += bidx[j] == b ? 1 : 0;
    if (pos >= n_max) {  // the caller's n_max was too small: never overwrite another label silently
        if (dropped) atomicAdd(dropped, 1);
        return;
    }
    const float cx = box[4 * i] * w, cy = box[4 * i + 1] * h, bw = box[4 * i + 2] * w, bh = box[4 * i + 3] * h;
    float* o = gt + ((int64_t)img * n_max + pos) * 5;
    o[0] = cls[i];
    o[1] = cx - bw / 2;
    o[2] = cy - bh / 2;
    o[3] = cx + bw / 2;
    o[4] = cy + bh / 2;
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_pad_targets(const float* batch_idx, const float* cls, const float* bboxes, int32_t n, int32_t N, int32_t n_max, float img_w,
                                float img_h, float* gt, int32_t* dropped, void* stream) {
    CDET_CHECK_ARG(gt && N > 0 && n_max > 0 && n >= 0, "cdet_pad_targets: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        const hipError_t e = hipMemsetAsync(gt, 0, (size_t)N * n_max * 5 * sizeof(float), s);
        CDET_CHECK_ARG(e == hipSuccess, "cdet_pad_targets: memset failed: %s", hipGetErrorString(e));
    }
    if (n == 0) return 0;
    CDET_CHECK_ARG(batch_idx && cls && bboxes, "cdet_pad_targets: null label arrays");
    hipLaunchKernelGGL(pad_targets_kernel, dim3((n + 127) / 128), dim3(128), 0, s, batch_idx, cls, bboxes, n, N, n_max, img_w, img_h, gt, dropped);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t cdet_det_loss_ws_bytes(const cdet_loss_desc* d) {
    if (!d) return -1;
    const int A = d->hw[0] * d->hw[1] + d->hw[2] * d->hw[3] + d->hw[4] * d->hw[5];
    return ws_layout(d->N, A, d->n_max).total;
}

extern "C" int cdet_det_loss(const cdet_loss_desc* d, const void* f0, const void* f1, const void* f2, const float* gt, void* df0, void* df1,
                             void* df2, float* out_loss5, uint8_t* fg_mask, int32_t* target_gt_idx, int32_t* target_labels,
                             float* target_bboxes, float* target_scores, void* ws, void* stream) {
    CDET_CHECK_ARG(d && f0 && f1 && f2 && out_loss5 && ws, "cdet_det_loss: null pointer");
    CDET_CHECK_ARG(d->n_max == 0 || gt, "cdet_det_loss: gt is null");
    CDET_CHECK_ARG(d->topk > 0 && d->topk <= MAX_TOPK, "cdet_det_loss: topk must be in [1, %d]", MAX_TOPK);
    CDET_CHECK_ARG(d->f_ld >= 64 + d->nc, "cdet_det_loss: f_ld (%d) < 64 + nc (%d)", d->f_ld, 64 + d->nc);
    LossArgs a;
    a.f[0] = f0; a.f[1] = f1; a.f[2] = f2;
    a.df[0] = df0; a.df[1] = df1; a.df[2] = df2;
    int off = 0;
    for (int i = 0; i < 3; ++i) {
        a.h[i] = d->hw[2 * i]; a.w[i] = d->hw[2 * i + 1]; a.stride[i] = d->stride[i];
        a.a_off[i] = off;
        off += a.h[i] * a.w[i];
    }
    a.a_off[3] = off;
    a.A = off;
    a.N = d->N; a.nc = d->nc; a.n_max = d->n_max; a.f_ld = d->f_ld; a.dtype = d->dtype; a.grad_dtype = d->grad_dtype; a.topk = d->topk;
    {
        uintptr_t al16 = 0;
        const void* ptrs[6] = {f0, f1, f2, df0, df1, df2};
        for (const void* q : ptrs) al16 |= reinterpret_cast<uintptr_t>(q);
        a.vec = (d->f_ld % 8 == 0 && (al16 & 15) == 0) ? 1 : 0;
    }
    a.alpha = d->alpha; a.beta = d->beta; a.gain_box = d->gain_box; a.gain_cls = d->gain_cls; a.gain_dfl = d->gain_dfl;
    a.grad_scale = d->grad_scale;
    a.grad_scale_dev = d->grad_scale_dev;
    a.gt = gt;
    const WsLayout L = ws_layout(d->N, a.A, d->n_max);
    char* p = (char*)ws;
    a.pbox = (float*)(p + L.pbox); a.dist = (float*)(p + L.dist); a.sel = (int*)(p + L.sel); a.cnt = (int*)(p + L.cnt);
    a.gsel = (int*)(p + L.gsel); a.tgt = (int*)(p + L.tgt); a.a_align = (float*)(p + L.a_align); a.a_over = (float*)(p + L.a_over);
    a.norm = (float*)(p + L.norm); a.g_align = (unsigned int*)(p + L.g_align); a.g_over = (unsigned int*)(p + L.g_over);
    a.part = (float*)(p + L.part); a.tss = (double*)(p + L.tss);
    a.out_loss = out_loss5; a.o_fg = fg_mask; a.o_gt_idx = target_gt_idx; a.o_labels = target_labels; a.o_bboxes = target_bboxes;
    a.o_scores = target_scores;
    hipStream_t s = (hipStream_t)stream;
    const int64_t NA = (int64_t)d->N * a.A;
    int nb = (int)((NA + 255) / 256);
    if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_decode_kernel, dim3(nb), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    if (d->n_max > 0) {
        const size_t shm = (size_t)a.A * sizeof(float);
        CDET_CHECK_ARG(shm <= 60 * 1024, "cdet_det_loss: too many anchors for the LDS metric buffer (A=%d)", a.A);
        hipLaunchKernelGGL(loss_gt_topk_kernel, dim3(d->N * d->n_max), dim3(256), shm, s, a);
        CDET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(loss_resolve_kernel, dim3(nb), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_norm_kernel, dim3(nb), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_tss_kernel, dim3(1), dim3(64), 0, s, a, nb);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_grad_kernel, dim3(nb), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, a, nb);
    CDET_LAUNCH_CHECK();
    return 0;
}
