// Batched NMS for gfx950: all images of the batch in ONE call, no host round trips.
// Replaces the per-image python loop of utils/general.py:411-474 (amax filter, boolean compaction, best-class /
// multi-label expansion, argsort, class offsets) and torchvision.ops.nms (general.py:464).
//
//   1. filter+compact (one workgroup per image): order-preserving compaction (block prefix sums) of the candidates in
//      ORIGINAL index order -- anchor order, or (anchor, class) row-major order for multi_label -- so that the sort
//      key (score desc, original position asc) reproduces a stable descending argsort. Scores are compared in the INPUT
//      dtype (fp16 stays fp16) and xywh->xyxy is evaluated with the input dtype's rounding, as the reference does;
//      everything after that is fp32 (general.py:446-449 promotes through `j.float()`).
//   2. sort (one workgroup per image): bitonic sort of 64-bit keys; in LDS when the image has <= 4096 candidates,
//      through the L2-resident workspace otherwise.
//   3. greedy suppression (one wavefront per image): candidates are visited in tiles of 64 (one per lane); a tile is first
//      tested against the list of already-kept boxes (LDS), then resolved internally with a 64x64 bit matrix held one
//      row per lane. Kept boxes are in descending score order, so the scan STOPS after max_det keeps -- the result equals
//      torchvision.ops.nms(...)[:max_det] without ever building the O(n^2) matrix.
// IoU arithmetic is torchvision's: inter/(area_a+area_b-inter) > thr, fp32, IEEE division, no eps, no +1.
#include "common.h"

namespace cdet {

constexpr float MAX_WH = 7680.0f;  // general.py:413

struct Cand {        // 32 bytes
    float x1, y1, x2, y2, conf, cls;
    uint32_t pos, anchor;  // position in the compacted list; the anchor the row came from (mask coefficients are gathered by it)
};

__device__ __forceinline__ float round_to(float v, int dtype) {
    if (dtype == CDET_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
    if (dtype == CDET_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    return v;
}

// monotone map float -> uint32 (ascending)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* sh_wave, int& total) {
    // 256 threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh_wave[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh_wave[w];
    total = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
    return base + inc - v;
}

struct NmsArgs {
    const void* pred;
    int N, nc, A, dtype;
    float conf_thres, iou_thres;
    int agnostic, multi_label, max_det, max_nms, max_cand;
    const int* classes;
    int n_classes;
    Cand* cand;             // [N][max_cand]
    unsigned long long* keys;  // [N][cap2]
    int cap2, sel_cap;      // per image: cap2 keys + sel_cap keys of the max_nms selection (validation settings)
    int* count;             // [N]
    int* seg_count;         // [N][n_seg]: candidates per anchor segment (filter pass 1 -> the slot base of pass 2)
    int n_seg, seg_len;     // anchors are split into n_seg runs of seg_len (a multiple of 256)
    float* out_rows;
    int* out_count;
    int* out_anchor;        // optional [N][max_det]
};

__device__ __forceinline__ bool class_allowed(const NmsArgs& a, int c) {
    if (a.n_classes <= 0) return true;
    for (int i = 0; i < a.n_classes; ++i)
        if (a.classes[i] == c) return true;
    return false;
}

// One workgroup per (image, anchor segment). The compacted list keeps the ANCHOR order (its position is the tie rule of the sort), so a
// segment needs the number of candidates in front of it: pass 1 (WRITE = false) counts per segment, pass 2 (WRITE = true) starts from the
// sum of the earlier segments' counts and writes. One workgroup walking all 8400 anchors of an image in 33 dependent rounds took 290 us.
constexpr int NMS_SEG_MAX = 32;
template <bool WRITE>
__global__ __launch_bounds__(256) void nms_filter_kernel(const NmsArgs a) {
    __shared__ int sh_wave[4];
    __shared__ int sh_base;
    const int n = blockIdx.x / a.n_seg, seg = blockIdx.x - n * a.n_seg;
    const int A = a.A, nc = a.nc;
    const float thr = round_to(a.conf_thres, a.dtype);
    const int64_t img = (int64_t)n * (4 + nc) * A;
    Cand* out = a.cand + (int64_t)n * a.max_cand;
    if (threadIdx.x == 0) {
        int b = 0;
        if (WRITE)
            for (int s = 0; s < seg; ++s) b += a.seg_count[n * a.n_seg + s];
        sh_base = b;
    }
    __syncthreads();
    const int first = sh_base;
    const int a_end = min(A, (seg + 1) * a.seg_len);
    __syncthreads();
    for (int a0 = seg * a.seg_len; a0 < a_end; a0 += 256) {
        const int an = a0 + threadIdx.x;
        int cnt = 0;
        float best = -INFINITY;
        int bestc = 0;
        if (an < a_end) {
            for (int c = 0; c < nc; ++c) {
                const float s = load_elem(a.pred, img + (int64_t)(4 + c) * A + an, a.dtype);
                if (s > best) {  // first maximum wins, like torch.max(1)
                    best = s;
                    bestc = c;
                }
                if (a.multi_label && s > thr && class_allowed(a, c)) ++cnt;
            }
            if (!a.multi_label) cnt = (best > thr && class_allowed(a, bestc)) ? 1 : 0;
        }
        int total;
        const int excl = block_exclusive_scan(cnt, sh_wave, total);
        const int base = sh_base;
        if (WRITE && cnt > 0) {
            const float cx = load_elem(a.pred, img + 0 * (int64_t)A + an, a.dtype), cy = load_elem(a.pred, img + 1 * (int64_t)A + an, a.dtype);
            const float w = load_elem(a.pred, img + 2 * (int64_t)A + an, a.dtype), h = load_elem(a.pred, img + 3 * (int64_t)A + an, a.dtype);
            const float hw = round_to(w / 2.f, a.dtype), hh = round_to(h / 2.f, a.dtype);
            Cand cd;
            cd.x1 = round_to(cx - hw, a.dtype);
            cd.y1 = round_to(cy - hh, a.dtype);
            cd.x2 = round_to(cx + hw, a.dtype);
            cd.y2 = round_to(cy + hh, a.dtype);
            cd.anchor = (uint32_t)an;
            int slot = base + excl;
            if (a.multi_label) {
                for (int c = 0; c < nc; ++c) {
                    const float s = load_elem(a.pred, img + (int64_t)(4 + c) * A + an, a.dtype);
                    if (s > thr && class_allowed(a, c)) {
                        if (slot < a.max_cand) {
                            cd.conf = s;
                            cd.cls = (float)c;
                            cd.pos = (uint32_t)slot;
                            out[slot] = cd;
                        }
                        ++slot;
                    }
                }
            } else if (slot < a.max_cand) {
                cd.conf = best;
                cd.cls = (float)bestc;
                cd.pos = (uint32_t)slot;
                out[slot] = cd;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_base = base + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (!WRITE) a.seg_count[n * a.n_seg + seg] = sh_base - first;
        else if (seg == a.n_seg - 1) a.count[n] = sh_base < a.max_cand ? sh_base : a.max_cand;
    }
}

// key = (~ord(conf) << 32) | pos : ascending sort == conf descending, original position ascending
constexpr int SORT_LDS_MAX = 4096;

__device__ __forceinline__ unsigned long long cand_key(const Cand* cand, int i) {
    return (((unsigned long long)(~f2ord(cand[i].conf))) << 32) | (unsigned long long)i;
}

// Bitonic sort (ascending) of p2 = 2^k keys, LDS-resident while the partners of a compare-exchange sit in the same 4096-key chunk: all stages of the
// sizes <= 4096 run chunk by chunk in ONE pass through LDS; for a larger size only its strides >= 4096 go through global memory (L2), the rest of the
// size again chunk by chunk. 32768 keys: 6 global passes + 4 LDS phases instead of 120 global passes. 1024 threads.
__device__ void bitonic_sort_hybrid(unsigned long long* __restrict__ k, int p2, unsigned long long* __restrict__ sk) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = p2 < SORT_LDS_MAX ? p2 : SORT_LDS_MAX;
    for (int size_lo = 2; size_lo <= p2;) {
        // global strides of this size (none while size <= chunk)
        int size_hi = size_lo;  // the LDS phase below covers sizes size_lo .. size_hi
        if (size_lo <= chunk) size_hi = chunk;
        else {
            for (int stride = size_lo >> 1; stride >= chunk; stride >>= 1) {
                for (int i = tid; i < (p2 >> 1); i += nt) {
                    const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                    const int hi = lo + stride;
                    const bool up = ((lo & size_lo) == 0);
                    const unsigned long long x = k[lo], y = k[hi];
                    if ((x > y) == up) {
                        k[lo] = y;
                        k[hi] = x;
                    }
                }
                __syncthreads();
            }
        }
        for (int c0 = 0; c0 < p2; c0 += chunk) {
            for (int i = tid; i < chunk; i += nt) sk[i] = k[c0 + i];
            __syncthreads();
            for (int size = size_lo; size <= size_hi; size <<= 1) {
                for (int stride = min(size >> 1, chunk >> 1); stride > 0; stride >>= 1) {
                    for (int i = tid; i < (chunk >> 1); i += nt) {
                        const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                        const int hi = lo + stride;
                        const bool up = (((c0 + lo) & size) == 0);
                        const unsigned long long x = sk[lo], y = sk[hi];
                        if ((x > y) == up) {
                            sk[lo] = y;
                            sk[hi] = x;
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = tid; i < chunk; i += nt) k[c0 + i] = sk[i];
            __syncthreads();
        }
        size_lo = size_hi << 1;
    }
}

// The K-th smallest (0-based rank K - 1) of cnt distinct 64-bit keys: MSB-first radix select, 8-bit digits, one 256-bin histogram per wave (a level's
// keys mostly share their high digits -- confidences of one exponent range -- so a single LDS histogram would serialise all 16 waves on a few bins).
__device__ unsigned long long radix_select_kth(const unsigned long long* __restrict__ k, int cnt, int K, unsigned* __restrict__ hist /* [16][256] */,
                                               unsigned* __restrict__ bins /* [256] */, unsigned long long* __restrict__ sh_pref, int* __restrict__ sh_rank) {
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
    if (tid == 0) {
        *sh_pref = 0ull;
        *sh_rank = K - 1;
    }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 16 * 256; i += nt) hist[i] = 0u;
        __syncthreads();
        const unsigned long long pref = *sh_pref;
        const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < cnt; i += nt) {
            const unsigned long long key = k[i];
            if ((key & himask) == pref) atomicAdd(&hist[wave * 256 + (int)((key >> shift) & 0xffull)], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            unsigned t = 0;
            for (int w = 0; w < 16; ++w) t += hist[w * 256 + tid];
            bins[tid] = t;
        }
        __syncthreads();
        if (tid == 0) {
            int r = *sh_rank, b = 0;
            while (b < 255 && r >= (int)bins[b]) {
                r -= (int)bins[b];
                ++b;
            }
            *sh_rank = r;
            *sh_pref = pref | ((unsigned long long)b << shift);
        }
        __syncthreads();
    }
    return *sh_pref;
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const NmsArgs a) {
    __shared__ unsigned long long sk[SORT_LDS_MAX];
    __shared__ unsigned hist[16 * 256], bins[256];
    __shared__ unsigned long long sh_pref;
    __shared__ int sh_rank, sh_slot;
    const int n = blockIdx.x;
    const int cnt = a.count[n];
    const Cand* cand = a.cand + (int64_t)n * a.max_cand;
    unsigned long long* keys = a.keys + (int64_t)n * (a.cap2 + a.sel_cap);
    if (cnt <= 1) {
        if (threadIdx.x == 0 && cnt == 1) keys[0] = 0ull;
        return;
    }
    int p2 = 1;
    while (p2 < cnt) p2 <<= 1;
    if (p2 <= SORT_LDS_MAX) {  // the inference regime: a few hundred candidates per image, one pass through LDS
        for (int i = threadIdx.x; i < p2; i += blockDim.x) sk[i] = i < cnt ? cand_key(cand, i) : ~0ull;
        __syncthreads();
        for (int size = 2; size <= p2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < (p2 >> 1); i += blockDim.x) {
                    const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned long long x = sk[lo], y = sk[hi];
                    if ((x > y) == up) {
                        sk[lo] = y;
                        sk[hi] = x;
                    }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) keys[i] = sk[i];
        return;
    }
    // validation settings (conf 0.001, multi_label: up to A x nc candidates per image). The reference keeps the max_nms = 30000 best before the
    // suppression (x[x[:, 4].argsort(descending=True)[:max_nms]], general.py:416,459): only those are sorted here -- selected first (the keys are
    // distinct, so the set of the K smallest is unique and the result equals the first K of the full sort), then sorted.
    const int K = cnt < a.max_nms ? cnt : a.max_nms;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) keys[i] = cand_key(cand, i);
    __syncthreads();
    if (K == cnt) {
        for (int i = cnt + threadIdx.x; i < p2; i += blockDim.x) keys[i] = ~0ull;
        __syncthreads();
        bitonic_sort_hybrid(keys, p2, sk);
        return;
    }
    const unsigned long long kth = radix_select_kth(keys, cnt, K, hist, bins, &sh_pref, &sh_rank);
    unsigned long long* sel = keys + a.cap2;
    int pk = 1;
    while (pk < K) pk <<= 1;
    if (threadIdx.x == 0) sh_slot = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < cnt; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        const unsigned long long key = i < cnt ? keys[i] : ~0ull;
        const bool take = i < cnt && key <= kth;
        const unsigned long long m = __ballot(take);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&sh_slot, __popcll(m));
        base = __shfl(base, 0);
        if (take) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
    }
    for (int i = K + threadIdx.x; i < pk; i += blockDim.x) sel[i] = ~0ull;
    __syncthreads();
    bitonic_sort_hybrid(sel, pk, sk);
    for (int i = threadIdx.x; i < K; i += blockDim.x) keys[i] = sel[i];
}

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2,
                                       float barea, float thr) {
#pragma clang fp contract(off)  // torchvision evaluates w*h and (a + b - inter) separately: no FMA fusion
    const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1), xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float iou = inter / (aarea + barea - inter);
    return iou > thr;
}

constexpr int KEPT_MAX = 2048;  // LDS list of kept boxes (max_det is clamped to this)
// Greedy suppression, one workgroup of FOUR waves per image over tiles of 64 score-sorted candidates. All waves hold the tile (lane = candidate).
//   1. against the boxes kept so far: wave w tests kept boxes w, w + 4, ... -- "suppressed by ANY earlier kept box" is an OR, so the split changes
//      nothing; the four dead masks meet in LDS;
//   2. inside the tile: row `lane` of the suppression matrix (later candidates this one would suppress), wave w computing columns 16 w .. 16 w + 15;
//   3. wave 0 walks the tile in order (keep the first alive candidate, clear its row from the alive mask, ...) exactly as the one-wave form did,
//      appends the kept boxes to the LDS list and writes the output rows.
// One wave doing all of 1 and 2 spent ~11 k clocks per tile (a division per pair: torchvision's iou > thr is kept as it is); four waves ~3 k.
__global__ __launch_bounds__(256) void nms_greedy_kernel(const NmsArgs a) {
    __shared__ float kx1[KEPT_MAX], ky1[KEPT_MAX], kx2[KEPT_MAX], ky2[KEPT_MAX], kar[KEPT_MAX];
    __shared__ unsigned long long s_dead[4], s_row[4][64];
    __shared__ int s_nkept;
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cnt = a.count[n];
    if (cnt > a.max_nms) cnt = a.max_nms;  // general.py:459
    const Cand* cand = a.cand + (int64_t)n * a.max_cand;
    const unsigned long long* keys = a.keys + (int64_t)n * (a.cap2 + a.sel_cap);
    float* rows = a.out_rows + (int64_t)n * a.max_det * 6;
    const float thr = a.iou_thres;
    int nkept = 0;
    for (int t0 = 0; t0 < cnt && nkept < a.max_det; t0 += 64) {
        const int i = t0 + lane;
        const bool have = i < cnt;
        Cand c;
        c.x1 = c.y1 = c.x2 = c.y2 = c.conf = c.cls = 0.f;
        c.anchor = 0u;
        if (have) c = cand[(uint32_t)(keys[i] & 0xffffffffull)];
        const float off = a.agnostic ? 0.f : c.cls * MAX_WH;
        const float bx1 = c.x1 + off, by1 = c.y1 + off, bx2 = c.x2 + off, by2 = c.y2 + off;
        const float area = (bx2 - bx1) * (by2 - by1);
        bool dead = false;
        for (int k = wave; k < nkept && !dead; k += 4)
            if (iou_gt(kx1[k], ky1[k], kx2[k], ky2[k], kar[k], bx1, by1, bx2, by2, area, thr)) dead = true;
        const unsigned long long dm = __ballot(dead);
        if (lane == 0) s_dead[wave] = dm;
        // intra-tile: row `lane` of the suppression matrix = later lanes this box would suppress (this wave's 16 columns)
        unsigned long long row = 0ull;
        for (int jj = 0; jj < 16; ++jj) {
            const int j = wave * 16 + jj;
            const float ox1 = __shfl(bx1, j), oy1 = __shfl(by1, j), ox2 = __shfl(bx2, j), oy2 = __shfl(by2, j), oar = __shfl(area, j);
            if (j > lane && iou_gt(bx1, by1, bx2, by2, area, ox1, oy1, ox2, oy2, oar, thr)) row |= (1ull << j);
        }
        s_row[wave][lane] = row;
        __syncthreads();
        if (wave == 0) {
            row = s_row[0][lane] | s_row[1][lane] | s_row[2][lane] | s_row[3][lane];
            unsigned long long alive_mask = __ballot(have) & ~(s_dead[0] | s_dead[1] | s_dead[2] | s_dead[3]);
            unsigned long long keep_mask = 0ull;
            int kept_here = 0;
            for (int j = 0; j < 64; ++j) {
                if (!((alive_mask >> j) & 1ull)) continue;
                if (nkept + kept_here >= a.max_det) break;
                keep_mask |= (1ull << j);
                ++kept_here;
                const unsigned long long rj = __shfl(row, j);  // uniform j: broadcast lane j's row
                alive_mask &= ~rj;
            }
            if ((keep_mask >> lane) & 1ull) {
                const int slot = nkept + __popcll(keep_mask & ((1ull << lane) - 1ull));
                if (slot < KEPT_MAX) {
                    kx1[slot] = bx1; ky1[slot] = by1; kx2[slot] = bx2; ky2[slot] = by2; kar[slot] = area;
                }
                float* r = rows + (int64_t)slot * 6;
                r[0] = c.x1; r[1] = c.y1; r[2] = c.x2; r[3] = c.y2; r[4] = c.conf; r[5] = c.cls;
                if (a.out_anchor != nullptr) a.out_anchor[(int64_t)n * a.max_det + slot] = (int)c.anchor;
            }
            if (lane == 0) s_nkept = nkept + kept_here;
        }
        __syncthreads();
        nkept = s_nkept;
    }
    if (threadIdx.x == 0) a.out_count[n] = nkept;
}

static inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}


// ------------------------------------------------------------------------------------------------
// cross-task merge (one workgroup per image): see cdet_merge_tasks in include/cerberus_hip.h
// ------------------------------------------------------------------------------------------------
constexpr int MERGE_MAX = 2048;

__global__ __launch_bounds__(256) void merge_tasks_kernel(const cdet_merge_desc d, const float* __restrict__ scale, float* __restrict__ out_rows,
                                                          int* __restrict__ out_count) {
#pragma clang fp contract(off)  // torch evaluates w*h, the sums and the division as separate fp32 operations: no FMA fusion here
    __shared__ float bx[MERGE_MAX][6];
    __shared__ unsigned char tk[MERGE_MAX], dead[MERGE_MAX], hit[MERGE_MAX];
    __shared__ int start[9];
    __shared__ unsigned long long wbest[4];
    __shared__ int s_any, s_n_dead;
    const int img = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < d.T; ++t) {
            start[t] = acc;
            acc += min(d.counts[t][img], d.max_det);
        }
        start[d.T] = acc;
        s_n_dead = 0;
    }
    __syncthreads();
    const int n = start[d.T];
    for (int t = 0; t < d.T; ++t) {
        const int cnt = start[t + 1] - start[t];
        const float* src = d.rows[t] + (int64_t)img * d.max_det * 6;
        for (int e = tid; e < cnt * 6; e += 256) {
            const int r = e / 6, c = e - r * 6;
            float v = src[e];
            if (c == 5) v += (float)d.cls_offset[t];
            bx[start[t] + r][c] = v;
        }
        for (int r = tid; r < cnt; r += 256) {
            tk[start[t] + r] = (unsigned char)t;
            dead[start[t] + r] = 0;
        }
    }
    __syncthreads();
    // greedy row scan (general.py:527-545). Rows of the last task have no later task -> no hits.
    const int r_end = start[d.T - 1];
    for (int r = 0; r < r_end; ++r) {
        if (dead[r]) continue;  // uniform: LDS flag, barrier-separated from its writers
        const float ax1 = bx[r][0], ay1 = bx[r][1], ax2 = bx[r][2], ay2 = bx[r][3];
        const float aarea = (ax2 - ax1) * (ay2 - ay1);
        const int j0 = start[tk[r] + 1];
        // best hit: highest score, lowest index on ties -> key = (score bits, ~index); scores are positive floats
        unsigned long long best = 0ull;
        for (int j = j0 + tid; j < n; j += 256) {
            const float iw = fmaxf(fminf(ax2, bx[j][2]) - fmaxf(ax1, bx[j][0]), 0.f);
            const float ih = fmaxf(fminf(ay2, bx[j][3]) - fmaxf(ay1, bx[j][1]), 0.f);
            const float inter = iw * ih;
            const float iou = inter / (aarea + (bx[j][2] - bx[j][0]) * (bx[j][3] - bx[j][1]) - inter + 1e-7f);
            const bool h = iou > d.iou_thres;
            hit[j] = h ? 1 : 0;
            if (h) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(bx[j][4]) << 32) | (unsigned)(0xffffffffu - (unsigned)j);
                best = key > best ? key : best;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = __shfl_xor(best, m);
            best = o > best ? o : best;
        }
        if ((tid & 63) == 0) wbest[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = wbest[0];
            for (int w = 1; w < 4; ++w) b = wbest[w] > b ? wbest[w] : b;
            wbest[0] = b;
            s_any = b != 0ull;
        }
        __syncthreads();
        if (s_any) {  // uniform
            const unsigned long long b = wbest[0];
            const int bj = (int)(0xffffffffu - (unsigned)(b & 0xffffffffull));
            const float bs = __uint_as_float((unsigned)(b >> 32));
            const bool row_wins = bx[r][4] > bs;  // the row itself comes LAST in the candidate list: it needs a strictly higher score
            for (int j = j0 + tid; j < n; j += 256)
                if (hit[j] && (row_wins || j != bj)) dead[j] = 1;
            if (tid == 0 && !row_wins) dead[r] = 1;
        }
        __syncthreads();
    }
    // survivors (all rows if everything was deleted, general.py:547-548), order preserved
    if (tid == 0) {
        int nd = 0;
        for (int i = 0; i < n; ++i) nd += dead[i];
        s_n_dead = nd;
    }
    __syncthreads();
    const bool keep_all = s_n_dead == n;
    float* dst = out_rows + (int64_t)img * d.T * d.max_det * 6;
    if (tid == 0) {
        int k = 0;
        for (int i = 0; i < n; ++i) {
            if (!keep_all && dead[i]) continue;
            float x1 = bx[i][0], y1 = bx[i][1], x2 = bx[i][2], y2 = bx[i][3];
            if (scale != nullptr) {  // scale_boxes + clip_boxes + round (cerberusdet_inference.py:161, general.py:313-357)
                const float g = scale[img * 5 + 0], px = scale[img * 5 + 1], py = scale[img * 5 + 2], h0 = scale[img * 5 + 3], w0 = scale[img * 5 + 4];
                x1 = rintf(fminf(fmaxf((x1 - px) / g, 0.f), w0));
                y1 = rintf(fminf(fmaxf((y1 - py) / g, 0.f), h0));
                x2 = rintf(fminf(fmaxf((x2 - px) / g, 0.f), w0));
                y2 = rintf(fminf(fmaxf((y2 - py) / g, 0.f), h0));
            }
            float* o = dst + (int64_t)k * 6;
            o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = bx[i][4]; o[5] = bx[i][5];
            ++k;
        }
        out_count[img] = k;
    }
}


// ------------------------------------------------------------------------------------------------
// validation matcher (one workgroup per image): see cdet_match_predictions in include/cerberus_hip.h
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void match_predictions_kernel(const cdet_match_desc d, const float* __restrict__ det_rows,
                                                                const int* __restrict__ det_count, const float* __restrict__ labels,
                                                                const int* __restrict__ label_start, const float* __restrict__ iouv,
                                                                uint8_t* __restrict__ correct) {
#pragma clang fp contract(off)  // box_iou evaluates products, sums and the division as separate fp32 operations
    extern __shared__ int owner[];  // [m][T]: lowest proposing prediction index per (label, IoU level)
    const int img = blockIdx.x, tid = threadIdx.x, T = d.T;
    const int n = min(det_count[img], d.max_det);
    const int l0 = label_start[img], m = min(label_start[img + 1] - l0, d.max_labels);
    for (int e = tid; e < m * T; e += 256) owner[e] = 0x7fffffff;
    uint8_t* out = correct + (int64_t)img * d.max_det * T;
    for (int e = tid; e < d.max_det * T; e += 256) out[e] = 0;
    __syncthreads();
    const float* lab = labels + (int64_t)l0 * 5;
    // pass 1: every prediction finds its label and announces itself at the levels it reaches
    for (int i = tid; i < n; i += 256) {
        const float* r = det_rows + ((int64_t)img * d.max_det + i) * 6;
        const float bx1 = r[0], by1 = r[1], bx2 = r[2], by2 = r[3], cls = r[5];
        const float barea = (bx2 - bx1) * (by2 - by1);
        float best = -1.f;
        int bl = -1;
        for (int l = 0; l < m; ++l) {
            if (lab[l * 5] != cls) continue;
            const float ax1 = lab[l * 5 + 1], ay1 = lab[l * 5 + 2], ax2 = lab[l * 5 + 3], ay2 = lab[l * 5 + 4];
            const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
            const float inter = iw * ih;
            const float iou = inter / ((ax2 - ax1) * (ay2 - ay1) + barea - inter + 1e-7f);
            if (iou >= best) {
                best = iou;
                bl = l;
            }
        }
        if (bl >= 0)
            for (int t = 0; t < T; ++t)
                if (best >= iouv[t]) atomicMin(&owner[bl * T + t], i);
    }
    __syncthreads();
    // pass 2: recompute (cheaper than keeping per-thread state across the barrier for max_det > 256) and test ownership
    for (int i = tid; i < n; i += 256) {
        const float* r = det_rows + ((int64_t)img * d.max_det + i) * 6;
        const float bx1 = r[0], by1 = r[1], bx2 = r[2], by2 = r[3], cls = r[5];
        const float barea = (bx2 - bx1) * (by2 - by1);
        float best = -1.f;
        int bl = -1;
        for (int l = 0; l < m; ++l) {
            if (lab[l * 5] != cls) continue;
            const float ax1 = lab[l * 5 + 1], ay1 = lab[l * 5 + 2], ax2 = lab[l * 5 + 3], ay2 = lab[l * 5 + 4];
            const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
            const float inter = iw * ih;
            const float iou = inter / ((ax2 - ax1) * (ay2 - ay1) + barea - inter + 1e-7f);
            if (iou >= best) {
                best = iou;
                bl = l;
            }
        }
        if (bl >= 0)
            for (int t = 0; t < T; ++t) out[i * T + t] = (best >= iouv[t] && owner[bl * T + t] == i) ? 1 : 0;
    }
}

}  // namespace cdet

using namespace cdet;

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// keys of the max_nms selection (only when an image can hold more candidates than max_nms and more than the LDS sort takes)
static int nms_sel_cap(const cdet_nms_desc* d) {
    const int max_nms = d->max_nms > 0 ? d->max_nms : 30000;
    return (d->max_cand > max_nms && d->max_cand > SORT_LDS_MAX) ? next_pow2(max_nms) : 0;
}

extern "C" int64_t cdet_nms_ws_bytes(const cdet_nms_desc* d) {
    if (!d) return -1;
    const int cap2 = next_pow2(d->max_cand);
    return align256((int64_t)d->N * d->max_cand * sizeof(Cand)) + align256((int64_t)d->N * (cap2 + nms_sel_cap(d)) * 8) + align256((int64_t)d->N * 4) +
           align256((int64_t)d->N * NMS_SEG_MAX * 4);
}

extern "C" int cdet_nms_batched_idx(const cdet_nms_desc* d, const void* pred, float* out_rows, int32_t* out_count, int32_t* out_anchor, void* ws,
                                    void* stream) {
    CDET_CHECK_ARG(d && pred && out_rows && out_count && ws, "cdet_nms_batched: null pointer");
    CDET_CHECK_ARG(d->conf_thres >= 0.f && d->conf_thres <= 1.f, "Invalid Confidence threshold %f, valid values are between 0.0 and 1.0", d->conf_thres);
    CDET_CHECK_ARG(d->iou_thres >= 0.f && d->iou_thres <= 1.f, "Invalid IoU %f, valid values are between 0.0 and 1.0", d->iou_thres);
    CDET_CHECK_ARG(d->max_det > 0 && d->max_det <= KEPT_MAX, "cdet_nms_batched: max_det must be in [1, %d]", KEPT_MAX);
    CDET_CHECK_ARG(d->max_cand > 0 && d->N > 0 && d->nc > 0 && d->A > 0, "cdet_nms_batched: bad sizes");
    NmsArgs a;
    a.pred = pred; a.N = d->N; a.nc = d->nc; a.A = d->A; a.dtype = d->dtype;
    a.conf_thres = d->conf_thres; a.iou_thres = d->iou_thres; a.agnostic = d->agnostic;
    a.multi_label = (d->multi_label && d->nc > 1) ? 1 : 0;  // general.py:419
    a.max_det = d->max_det; a.max_nms = d->max_nms > 0 ? d->max_nms : 30000; a.max_cand = d->max_cand;
    a.classes = d->classes; a.n_classes = d->classes ? d->n_classes : 0;
    a.cap2 = next_pow2(d->max_cand);
    a.sel_cap = nms_sel_cap(d);
    char* p = (char*)ws;
    a.cand = (Cand*)p; p += align256((int64_t)d->N * d->max_cand * sizeof(Cand));
    a.keys = (unsigned long long*)p; p += align256((int64_t)d->N * (a.cap2 + a.sel_cap) * 8);
    a.count = (int*)p; p += align256((int64_t)d->N * 4);
    a.seg_count = (int*)p;
    // enough segments to put a few workgroups on every CU, each a whole number of 256-anchor rounds
    int n_seg = (1024 + d->N - 1) / d->N;
    if (n_seg > NMS_SEG_MAX) n_seg = NMS_SEG_MAX;
    const int rounds = (d->A + 255) / 256;
    if (n_seg > rounds) n_seg = rounds;
    a.seg_len = (rounds + n_seg - 1) / n_seg * 256;
    a.n_seg = (d->A + a.seg_len - 1) / a.seg_len;
    a.out_rows = out_rows; a.out_count = out_count; a.out_anchor = out_anchor;
    hipStream_t s = (hipStream_t)stream;
    if (a.n_seg > 1) {
        hipLaunchKernelGGL(nms_filter_kernel<false>, dim3(d->N * a.n_seg), dim3(256), 0, s, a);
        CDET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_filter_kernel<true>, dim3(d->N * a.n_seg), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_sort_kernel, dim3(d->N), dim3(1024), 0, s, a);
    CDET_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(d->N), dim3(256), 0, s, a);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_nms_batched(const cdet_nms_desc* d, const void* pred, float* out_rows, int32_t* out_count, void* ws, void* stream) {
    return cdet_nms_batched_idx(d, pred, out_rows, out_count, nullptr, ws, stream);
}

extern "C" int cdet_merge_tasks(const cdet_merge_desc* d, const float* scale, float* out_rows, int32_t* out_count, void* stream) {
    CDET_CHECK_ARG(d && out_rows && out_count, "cdet_merge_tasks: null pointer");
    CDET_CHECK_ARG(d->N > 0 && d->T >= 1 && d->T <= 8 && d->max_det > 0 && d->T * d->max_det <= MERGE_MAX,
                   "cdet_merge_tasks: needs 1 <= T <= 8 and T*max_det <= %d (T=%d max_det=%d)", MERGE_MAX, d->T, d->max_det);
    for (int t = 0; t < d->T; ++t) CDET_CHECK_ARG(d->rows[t] && d->counts[t], "cdet_merge_tasks: null per-task pointer");
    hipLaunchKernelGGL(merge_tasks_kernel, dim3(d->N), dim3(256), 0, (hipStream_t)stream, *d, scale, out_rows, out_count);
    CDET_LAUNCH_CHECK();
    return 0;
}

extern "C" int cdet_match_predictions(const cdet_match_desc* d, const float* det_rows, const int32_t* det_count, const float* labels,
                                      const int32_t* label_start, const float* iouv, uint8_t* correct, void* stream) {
    CDET_CHECK_ARG(d && det_rows && det_count && label_start && iouv && correct, "cdet_match_predictions: null pointer");
    CDET_CHECK_ARG(d->N > 0 && d->max_det > 0 && d->T >= 1 && d->T <= 16 && d->max_labels >= 0, "cdet_match_predictions: bad descriptor");
    const size_t shm = (size_t)(d->max_labels > 0 ? d->max_labels : 1) * d->T * sizeof(int);
    CDET_CHECK_ARG(shm <= 160 * 1024, "cdet_match_predictions: max_labels*T too large for LDS (%d labels)", d->max_labels);
    CDET_CHECK_ARG(labels || d->max_labels == 0, "cdet_match_predictions: labels is null");
    if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)match_predictions_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(match_predictions_kernel, dim3(d->N), dim3(256), shm, (hipStream_t)stream, *d, det_rows, det_count, labels, label_start,
                       iouv, correct);
    CDET_LAUNCH_CHECK();
    return 0;
}
