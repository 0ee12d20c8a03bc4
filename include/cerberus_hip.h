/*
 * cerberus_hip.h -- C-ABI of libcerberus_hip.so: hand-written CDNA4 (gfx950) kernels for the
 * CerberusDet hot path (shared backbone -> per-task neck -> Detect head -> {loss | NMS}).
 *
 * The reference (ai-forever/CerberusDet) has NO native/FFI interface: every op on its hot path is a
 * stock torch op reached from Python. Each entry point below therefore cites
 * the reference Python call site(s) it replaces (paths relative to cerberusdet/ in the reference).
 *
 * Conventions
 *   - plain C: pointers + sizes + POD descriptors, no torch / C++ types;
 *   - the CALLER owns every buffer (device pointers of tensors allocated by the host framework);
 *     the library never allocates or frees device memory and keeps no pointer past return;
 *   - every launch is asynchronous on the hipStream_t passed last (void* here so that the header
 *     needs no HIP include); no entry point synchronises;
 *   - return 0 on success, <0 on failure (-1000-x = argument validation, otherwise -hipError_t);
 *     cdet_last_error() returns a thread-local message;
 *   - thread-safe: no global mutable state besides the thread-local error string;
 *   - activations are NHWC ("channels last"); a tensor is described by (ptr, ld, coff): pixel p,
 *     channel c lives at ptr[p*ld + coff + c] -- this lets producers write straight into channel
 *     slices of a concat buffer, so torch.cat (common.py:191,245,295) never materialises.
 */
#ifndef CERBERUS_HIP_H
#define CERBERUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDET_ABI_VERSION 1

/* element types */
enum { CDET_BF16 = 0, CDET_F16 = 1, CDET_F32 = 2, CDET_U8 = 3 };
/* activations */
enum { CDET_ACT_NONE = 0, CDET_ACT_SILU = 1 };
/* conv gather modes */
enum { CDET_CONV_FWD = 0, CDET_CONV_DGRAD = 1 };

int cdet_version(void);
const char* cdet_last_error(void);
/* device properties the host needs to size workspaces: out[0]=CU count, out[1]=LDS bytes/CU, out[2]=gfx arch number */
int cdet_device_info(int32_t* out3);

/* Kernel-form switches (csrc/switches.h): pins for A/B timing and for the parity tests that compare two forms of one kernel. The library reads
 * its CDET_* environment variables ONCE, when it is loaded; no launch path calls getenv afterwards. The reference has no counterpart (its
 * "switches" are torch.backends flags set at import, train.py:28-30); bench.py prints cdet_active_switches() so that a run's configuration is
 * part of its record. name: the switch ("conv_pp") or its environment variable ("CDET_CONV_PP"); value CDET_SWITCH_DEFAULT restores the default. */
#define CDET_SWITCH_DEFAULT (-2147483647 - 1)
int cdet_set_switch(const char* name, int32_t value);
int cdet_get_switch(const char* name, int32_t* value);   /* CDET_SWITCH_DEFAULT-valued result: not pinned, the geometry rules decide */
int cdet_active_switches(char* buf, int32_t n);          /* "name=value ..." of every switch off its default + the build flavour; returns the length */
int cdet_has_experiments(void);                          /* 1: built with -DCDET_EXPERIMENTS (the opt-in forms kept for the record are present) */

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (bf16/f16 in, fp32 accumulate).
 * Replaces nn.Conv2d inside Conv.forward / fuseforward (models/common.py:57-68), the biased 1x1 head
 * projections (models/yolo.py:82-84) and -- in DGRAD mode -- autograd's convolution_backward(input).
 *
 *   FWD  : y[n,oy,ox,co] = sum_{kh,kw,ci} x[n, oy*s-pad+kh, ox*s-pad+kw, ci] * w[co][kh][kw][ci]
 *   DGRAD: y[n,iy,ix,ci] = sum_{kh,kw,co} x[n,(iy+pad-kh)/s,(ix+pad-kw)/s, co] * w[ci][kh][kw][co]
 *          (terms with non-integer or out-of-range source coordinates are zero)
 * then    y = act(y * scale[c] + bias[c]) (+ residual), each optional.
 * `w` is the PACKED weight produced by cdet_pack_weight: [Crows][Kpad] of the activation dtype,
 * K = kh*kw*Csrc contiguous, zero-padded to a multiple of 64.
 * If stats != NULL the kernel also writes per-(pixel-block, channel) partial sums of the RAW fp32
 * convolution result: stats[(blk*2+0)*Cout + c] = sum, stats[(blk*2+1)*Cout + c] = sum of squares
 * (blk < cdet_conv2d_stat_blocks(desc)); this feeds train-mode BatchNorm (common.py:61).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, Hs, Ws, Cs;     /* source tensor: batch, height, width, channels (reduction channels)      */
    int32_t Hd, Wd, Cd;        /* destination tensor: height, width, channels (GEMM rows of `w`)          */
    int32_t kh, kw, stride, pad;
    int32_t mode;              /* CDET_CONV_FWD / CDET_CONV_DGRAD                                         */
    int32_t dtype;             /* element type of x, w, residual (CDET_BF16 / CDET_F16)                  */
    int32_t out_dtype;         /* CDET_BF16 / CDET_F16 / CDET_F32                                        */
    int32_t act;               /* CDET_ACT_*                                                             */
    int32_t src_ld, src_coff;  /* source pixel stride / channel offset (elements)                        */
    int32_t dst_ld, dst_coff;
    int32_t res_ld, res_coff;  /* residual (same dtype as x), used when residual != NULL                 */
    int32_t accumulate;        /* !=0: y += result (out_dtype must be F32); used for gradient fan-in     */
} cdet_conv_desc;

int cdet_conv2d_stat_blocks(const cdet_conv_desc* d);
int cdet_conv2d(const cdet_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* bias,
                const void* residual, void* y, float* stats, void* stream);

/* OIHW fp32 master weight -> packed GEMM operand.
 *   transpose == 0: rows = O, K order (kh, kw, i)   [forward operand]
 *   transpose == 1: rows = I, K order (kh, kw, o)   [DGRAD operand]
 * row_scale (optional, length O): multiplies output channel o (BN folding, utils/torch_utils.py:191-217).
 * Replaces the implicit fp32->half cast of autocast (trainers/averaging.py:158) and fuse_conv_and_bn. */
int cdet_pack_weight(const float* w_oihw, void* w_packed, int32_t O, int32_t O_pad, int32_t I, int32_t kh, int32_t kw,
                     int32_t transpose, const float* row_scale, int32_t dtype, void* stream);
/* O_pad >= O: output channels O..O_pad-1 are zero (head maps are padded to a multiple of 8 channels). */
int64_t cdet_packed_weight_elems(int32_t O_pad, int32_t I, int32_t kh, int32_t kw, int32_t transpose);

/* The forward AND the DGRAD operand of many convolutions in ONE launch: after an optimizer step every Conv of the model needs
 * both re-cast (the per-module autocast casts of trainers/averaging.py:158; ~320 launches per step otherwise, each reading
 * OIHW with a 36-byte stride). One workgroup transposes a [32 o][32 i][kh*kw] tile through LDS: the fp32 master is read once,
 * contiguously, and both layouts are written in runs of 32 elements. Only VALID elements are written: the K tail up to Kpad
 * and the rows O..O_pad-1 of the destination buffers must have been zeroed once by the caller. kh*kw <= 9.
 * `items` is a DEVICE array; first_block / n_blocks (= ceil(O/32)*ceil(I/32)) partition the grid of n_blocks_total workgroups
 * (first_block ascending, contiguous). w_dgrad may be NULL. */
typedef struct {
    const float* w_oihw;
    void* w_fwd;   /* cdet_pack_weight(..., transpose = 0) layout */
    void* w_dgrad; /* cdet_pack_weight(..., transpose = 1) layout, or NULL */
    int32_t O, O_pad, I, kh, kw;
    int32_t first_block, n_blocks;
} cdet_pack_item;
int cdet_pack_weights_batched(const cdet_pack_item* items, int32_t n_items, int32_t n_blocks_total, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tap-resident convolution (csrc/conv_halo.hip): the fast path for the stride-1 3x3 and 1x1 Conv modules
 * (models/common.py:57-68 -- 80 % of the FLOPs of the YOLOv8x path) and for their data gradient.
 * Same arithmetic and epilogue as cdet_conv2d (y = act(conv * scale + bias) (+ residual), optional BN partial sums in
 * `stats` with cdet_conv2d_tiled_stat_blocks(d) pixel blocks of 256 or 128), but the weight operand is the TILED pack below and
 * the pixel tile is staged in LDS once per 32-channel chunk with its halo and reused by all nine taps.
 * Supported (cdet_conv2d_tiled_ok(d) == 1): kh == kw in {1,3}, stride 1, pad kh/2, channels / ld / coff multiples of 8,
 * 16-bit input; output of the same type, or CDET_F32 (round 4: Detect's biased 1x1 projections, models/yolo.py:82-100, and with
 * d->accumulate the fp32 fan-in form y += result; no `stats` then); 3x3: Ws <= 95 (256 consecutive pixels + linear halo) or Hs, Ws
 * multiples of 16 (16 x 16 pixel patches); source / weight buffers below 3 GiB (32-bit buffer addressing). The data gradient of such a convolution is a FWD call (d->mode = CDET_CONV_FWD,
 * source = dY with Cs = Cout, destination = dX with Cd = Cin) on the DGRAD operand written by the packer: autograd's
 * convolution_backward(input) for a stride-1 "same" convolution is the forward convolution with the taps flipped and
 * the channel roles swapped.
 * ---------------------------------------------------------------------------------------------- */
int cdet_conv2d_tiled_ok(const cdet_conv_desc* d);
int cdet_conv2d_tiled_stat_blocks(const cdet_conv_desc* d);
int cdet_conv2d_tiled(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                      const void* residual, void* y, float* stats, void* stream);
/* Round 4 -- the same 1x1 convolution over a VIRTUAL Concat (+ nearest 2x Upsample): models/common.py:288-295 (Concat) and nn.Upsample in front of
 * every neck C2f's cv1 (models/yolo.py:172-203 walks them as separate modules and materialises both). Here the K loop visits up to three channel
 * segments, each a slice of its own NHWC buffer (`upsample`: read through (y / 2, x / 2) of an H/2 x W/2 map): neither the upsampled map nor the
 * concatenated tensor is written. d describes the convolution (Cs = total channels; its src_ld / src_coff are ignored); all but the last segment hold
 * a multiple of 32 channels; w_tiled = the ordinary forward operand over the concatenated channels. Same arithmetic and summation order as
 * cdet_conv2d_tiled on the materialised tensor. */
typedef struct {
    const void* x;             /* NHWC buffer of the segment (N x H x W, or N x H/2 x W/2 when upsample != 0) */
    int32_t ld, coff, C;       /* row pitch, first channel, channels                                           */
    int32_t upsample;
} cdet_cat_src;
int cdet_conv2d_tiled_cat_ok(const cdet_conv_desc* d, const cdet_cat_src* srcs, int32_t n_src);
int cdet_conv2d_tiled_cat(const cdet_conv_desc* d, const cdet_cat_src* srcs, int32_t n_src, const void* w_tiled, const float* scale, const float* bias,
                          const void* residual, void* y, void* stream);
/* Tiled operand: [ceil(rows/RB)][ceil(red/32)][kh*kw][RB][32] elements of the activation dtype, RB = 160 (96 when
 * rows <= 96), 16-byte slots XOR-swizzled (the LDS image of one (row block, chunk, tap) tile is a linear copy).
 * rows = O, red = I for the forward operand; rows = I, red = O, taps flipped, for the DGRAD operand. Rows beyond `rows`
 * are never written: the caller zeroes the buffers once (cdet_tiled_weight_elems elements). */
int64_t cdet_tiled_weight_elems(int32_t rows, int32_t red, int32_t kh, int32_t kw);
typedef struct {
    const float* w_oihw;
    void* w_fwd;   /* forward operand or NULL */
    void* w_dgrad; /* DGRAD operand or NULL */
    int32_t O, I, kh, kw;
    int32_t first_block, n_blocks; /* n_blocks = ceil(O/32) * ceil(I/32); first_block ascending, contiguous */
} cdet_pack_tiled_item;
/* every tiled operand of a model in ONE launch (`items` is a DEVICE array), the per-step re-cast of the fp32 masters
 * (trainers/averaging.py:158 autocast) */
int cdet_pack_weights_tiled(const cdet_pack_tiled_item* items, int32_t n_items, int32_t n_blocks_total, int32_t dtype, void* stream);
/* one convolution (w_fwd and/or w_dgrad may be NULL) */
int cdet_pack_weight_tiled(const float* w_oihw, void* w_fwd, void* w_dgrad, int32_t O, int32_t I, int32_t kh, int32_t kw,
                           int32_t dtype, void* stream);

/* Weight gradient: dw[o][i][kh][kw] (+)= sum_{n,oy,ox} dy[n,oy,ox,o] * x[n,oy*s-pad+kh,ox*s-pad+kw,i]
 * (autograd's convolution_backward(weight) for models/common.py:57). dw is fp32 OIHW; `ws` is a caller-provided
 * fp32 workspace of cdet_conv2d_wgrad_ws_elems(d) elements (split-K partials), may be NULL when that is 0. */
int64_t cdet_conv2d_wgrad_ws_elems(const cdet_conv_desc* d);
int cdet_conv2d_wgrad(const cdet_conv_desc* d, const void* x, const void* dy, float* dw_oihw, float* ws,
                      int32_t accumulate, void* stream);

/* Weight gradients of n layers of IDENTICAL geometry in one launch (the 2n Bottleneck convolutions of a C2f block: each is its own
 * autograd convolution_backward(weight) in the reference, models/common.py:107-117). One workgroup per CU over ALL layers, so the
 * pixel split per layer -- and with it the fp32 partial-slab round trip -- shrinks by n. `d` carries the shared geometry (N, H, W, Cs,
 * Cd, dtype, dst_ld/dst_coff of dy); each item its own tensors and the channel stride / offset of its x view.
 * cdet_conv2d_wgrad_groupable(d): 1 when the tap-resident kernel takes the geometry (stride-1 3x3; Cs % 32 == 0 and Cd >= 128, or
 * Cd <= 96 with Cs % 16 == 0). The item table lives in device memory, so the library cannot validate it: src_ld and src_coff of every
 * item must be multiples of 8 (16-byte rows), N*H*W*src_ld*2 < 3 GiB, and every dw distinct. */
typedef struct {
    const void* x;      /* source activation view (NHWC), channel stride src_ld, first channel src_coff */
    const void* dy;     /* gradient of the convolution output, layout d->dst_ld / d->dst_coff             */
    float* dw;          /* fp32 OIHW [Cd, Cs, 3, 3]                                                       */
    int32_t src_ld, src_coff;
} cdet_wgrad_item;
int32_t cdet_conv2d_wgrad_groupable(const cdet_conv_desc* d);
int64_t cdet_conv2d_wgrad_grouped_ws_elems(const cdet_conv_desc* d, int32_t n_items);
int cdet_conv2d_wgrad_grouped(const cdet_conv_desc* d, const cdet_wgrad_item* items_dev, int32_t n_items, float* ws,
                              int32_t accumulate, void* stream);

/* Stride-2 3x3 convolutions (models/common.py:57-62 with s = 2: the down-sampling rows of backbone and neck) on the tap-resident
 * machinery (csrc/conv_vt.hip): the input's four parity planes are staged as strided gathers and serve 4 / 2 / 2 / 1 taps each.
 * cdet_conv2d_s2_tiled: forward, arguments as cdet_conv2d_tiled (w_tiled = the forward operand of cdet_pack_weights_tiled).
 * cdet_conv2d_s2_tiled_dgrad: data gradient; d is the CDET_CONV_DGRAD descriptor cdet_conv2d takes (source = dY, destination = dX),
 * w_tiled = the DGRAD operand of cdet_pack_weights_tiled, residual = the gradient already in dX's place (fan-in) or NULL; the four
 * parity classes of dX run as ONE launch. scale / bias / stats must be NULL. Both take an fp32 destination (and d->accumulate) like
 * cdet_conv2d_tiled. _ok: 1 when the geometry is taken (d->mode selects forward or data gradient). */
int cdet_conv2d_s2_tiled_ok(const cdet_conv_desc* d);
int cdet_conv2d_s2_tiled_stat_blocks(const cdet_conv_desc* d);
int cdet_conv2d_s2_tiled(const cdet_conv_desc* d, const void* x, const void* w_tiled, const float* scale, const float* bias,
                         const void* residual, void* y, float* stats, void* stream);
int cdet_conv2d_s2_tiled_dgrad(const cdet_conv_desc* d, const void* dy, const void* w_tiled, const float* scale, const float* bias,
                               const void* residual, void* dx, float* stats, void* stream);

/* Eval-form fusion of the first TWO backbone rows (csrc/stem_conv1.hip, round 4): Conv(3, c1, 3, 2) + Conv(c1, c2, 3, 2), BatchNorm folded
 * (models/common.py:57-68 as run by models/yolo.py:172-203 in eval mode). The stem's map -- the largest tensor of the network -- is
 * produced per 16 x 16 output tile inside LDS and never written: same arithmetic and summation order as cdet_stem_conv followed by
 * cdet_conv2d_s2_tiled. img: NCHW as for cdet_stem_conv; w_stem_packed: cdet_stem_conv1_pack(w_stem fp32 [c1,3,3,3]) (cdet_stem_conv1_pack_elems
 * 16-bit elements); w1_tiled: the forward operand of cdet_pack_weights_tiled for the second row; y: NHWC [N, H/4, W/4, dst_ld] 16-bit.
 * cdet_stem_conv1_ok: H, W multiples of 4, c1 <= 96, c2 <= 160, both multiples of 8. */
int cdet_stem_conv1_ok(int32_t N, int32_t H, int32_t W, int32_t c1, int32_t c2, int32_t img_dtype, int32_t dtype, int32_t dst_ld, int32_t dst_coff);
int64_t cdet_stem_conv1_pack_elems(int32_t c1);
int cdet_stem_conv1_pack(const float* w_stem, void* out, int32_t c1, int32_t dtype, void* stream);
int cdet_stem_conv1(const void* img, int32_t img_dtype, const void* w_stem_packed, const float* stem_scale, const float* stem_bias,
                    const void* w1_tiled, const float* scale, const float* bias, void* y, int32_t N, int32_t H, int32_t W, int32_t c1,
                    int32_t c2, int32_t dtype, int32_t dst_ld, int32_t dst_coff, int32_t act, void* stream);

/* Stem convolution (models/common.py:57 for the first backbone row, Cin = 3): reads the image in the
 * reference's NCHW layout (uint8 scaled by 1/255 -- trainers/base_trainer.py:61-63 -- or float), 3x3 stride 2 pad 1,
 * writes NHWC. Direct (non-MFMA) kernel: K = 27, HBM-bound. Same epilogue/stat semantics as cdet_conv2d. */
int cdet_stem_conv(const void* img_nchw, int32_t img_dtype, const float* w_oihw, const float* scale, const float* bias,
                   void* y, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t out_dtype, int32_t act,
                   float* stats, void* stream);
int cdet_stem_conv_stat_blocks(int32_t N, int32_t H, int32_t W);
/* Image repack for running the stem on the generic MFMA kernels: NCHW (uint8 * 1/255, or float) -> NHWC with 3 channels
 * zero-padded to 8, 16-bit. (trainers/base_trainer.py:61-63 preprocess + the implicit NCHW->channels-last copy.) */
int cdet_image_to_nhwc8(const void* img_nchw, int32_t img_dtype, void* out_nhwc8, int32_t N, int32_t H, int32_t W, int32_t dtype,
                        void* stream);
/* d(stem weight) (autograd's convolution_backward(weight) of the first Conv, models/common.py:57) from dy ([N,H/2,W/2,Cout] rows of
 * dy_ld elements, bf16/f16) and the NCHW image itself (uint8 * 1/255, or float): MFMA kernel, per-workgroup fp32 partials in `ws`
 * (cdet_stem_conv_wgrad_ws_elems floats), fixed-order finish; (+)= into fp32 OIHW [Cout,3,3,3]. Cout % 8 == 0, <= 80; H, W even. */
int64_t cdet_stem_conv_wgrad_ws_elems(int32_t N, int32_t H, int32_t W);
int cdet_stem_conv_wgrad(const void* img_nchw, int32_t img_dtype, const void* dy, int32_t dy_ld, int32_t dtype, float* dw_oihw,
                         int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t accumulate, float* ws, void* stream);
/* Weight gradient of the stem computed on the channel-padded image copy (cdet_image_to_nhwc8 + cdet_conv2d_wgrad): fold the
 * first I_real input channels of dw_pad [O, I_pad, kh, kw] into dw [O, I_real, kh, kw] (+= when accumulate). Completes
 * `loss.backward()` for models/common.py:57's first Conv without leaving the launch list. */
int cdet_fold_padded_wgrad(const float* dw_pad, float* dw, int32_t O, int32_t I_pad, int32_t I_real, int32_t taps, int32_t accumulate,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inference pre-processing (cerberusdet_preprocessor.py:42-74, data/augmentations.py:59-89): letterbox of a list of
 * uint8 BGR HWC images into ONE [B,3,H,W] RGB tensor scaled by 1/255 (out_dtype F32 / F16 / BF16) or kept as U8.
 * Per image the host passes the source (device pointer, h, w, row pitch in bytes) and the letterbox geometry computed
 * exactly like augmentations.py:65-86: the resized size (new_w, new_h) and the top / left border; everything outside
 * the resized image is pad_value (114). Resize arithmetic: OpenCV's 8-bit INTER_LINEAR. `items` is a DEVICE array.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* img;
    int32_t h, w, pitch;
    int32_t new_w, new_h, top, left;
    int32_t area;                  /* 0: cv2.INTER_LINEAR (letterbox(), the inference pre-processing); 1: cv2.INTER_AREA when the image shrinks
                                      (load_image of the non-augmented loaders, data/datasets.py:473-476) -- occupies former padding */
} cdet_letterbox_item;
int cdet_letterbox_batch(const cdet_letterbox_item* items, int32_t B, void* out_nchw, int32_t H, int32_t W, int32_t out_dtype,
                         int32_t pad_value, void* stream);

/* Training-time augmentation, rendered in one launch per batch from the ORIGINAL decoded images: mosaic of four images
 * (data/datasets.py:483-527 load_mosaic, incl. the cv2.resize of load_image 470-477), cv2.warpAffine of random_perspective
 * (data/augmentations.py:151), mixup with a second mosaic (205-211), augment_hsv (43-57), flips, BGR->RGB, HWC->CHW
 * (datasets.py:420-438). The host samples every random parameter in the reference's order and transforms the labels
 * (cerberusdet_amd/augment.py); this call only renders. out: [B, 3, s, s] uint8 RGB. */
typedef struct {
    const void* img;               /* uint8 HWC BGR original image on the device                       */
    int32_t h0, w0, pitch;         /* its size, bytes per row                                          */
    int32_t h, w;                  /* size after load_image (long side -> s)                           */
    int32_t x1a, y1a, x2a, y2a;    /* destination rectangle on the 2s x 2s canvas                      */
    int32_t x1b, y1b;              /* origin of the pasted part in the resized image                   */
    int32_t reserved;
} cdet_aug_tile;
typedef struct {
    cdet_aug_tile tiles[8];        /* mosaic 0: tiles 0..3; the mixup partner: tiles 4..7              */
    double minv[18];               /* per mosaic 9 doubles, output -> canvas. perspective == 0: a11 a12 b1 a21 a22 b2 (+3 unused), the
                                    * inverse cv2.warpAffine derives; perspective != 0: the row-major 3x3 inverse cv2.warpPerspective derives */
    double mix_ratio;              /* weight of mosaic 0 (np.random.beta(32, 32))                      */
    int32_t n_mosaic;              /* 1, or 2 with mixup                                               */
    int32_t flipud, fliplr, use_hsv;
    int32_t canvas;                /* side of the square the tiles sit on: 2s (mosaic) or s (one letterboxed image, tile 0 only) */
    int32_t perspective;           /* hyp['perspective'] != 0: cv2.warpPerspective (augmentations.py:152-153) instead of cv2.warpAffine */
    uint8_t lut[768];              /* augment_hsv's hue / sat / val lookup tables                      */
} cdet_aug_sample;
int cdet_mosaic_augment_batch(const cdet_aug_sample* samples, int32_t B, void* out_u8_nchw, int32_t s, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d(eps 1e-3, momentum 0.03) + SiLU around the conv (common.py:61, torch_utils.py:184-186)
 * ---------------------------------------------------------------------------------------------- */
/* Reduce the conv kernel's partial sums -> mean, invstd (biased var), update running stats (unbiased var). */
int cdet_bn_finalize(const float* stats, int32_t nblk, int32_t C, int64_t count, float eps, float momentum,
                     float* running_mean, float* running_var, float* mean, float* invstd, void* stream);
/* Round 5 -- the reduction of the partial sums INSIDE the launch that produces them (csrc/bn_fold.h): the per-layer cdet_bn_finalize launch (and the
 * bn_bwd_sums launch inside cdet_bn_silu_bwd_apply) of the reference's nn.BatchNorm2d in train mode (models/common.py:57-62) disappear from the
 * per-GPU training plans. The producers -- cdet_conv2d_tiled_bn, cdet_conv2d_s2_tiled_bn (raw 16-bit convolution output + partial rows), and
 * cdet_bn_silu_bwd_reduce_fold -- take a DEVICE-resident descriptor: every workgroup stores its share of a partial row (write-through), draws a
 * ticket; the last arriver of each cluster of 32 rows adds the cluster's rows (ascending, double), the last arriver over the clusters adds the
 * cluster sums (ascending, double), rounds the totals to fp32 and finishes exactly as cdet_bn_finalize / cdet_bn_bwd_sums do. The order of every
 * addition is fixed by row numbers, and cdet_bn_finalize / cdet_bn_bwd_sums use the same tree (for <= 4096 rows): results are bit-identical with
 * and without the fold. fold_dev == NULL: the partial rows only (the caller runs cdet_bn_finalize). SyncBatchNorm plans keep the separate
 * kernels (the exchange sits between them).
 *   tickets   CDET_BN_FOLD_TICKET_WORDS 32-bit words, zeroed ONCE by the caller (the last arrivers reset them); one set per stream of launches
 *   cl_sums   cdet_bn_fold_cl_doubles(rows, C) doubles of scratch
 *   totals    optional [2][C] fp32: the rounded totals ([sum, sumsq] forward -- what a deferred running-statistics update or SyncBatchNorm
 *             reads; [sum dact, sum dact xhat] backward = the `sums` vector cdet_bn_silu_bwd_apply(nblk = 0) takes)
 *   forward   mean, invstd (out), running_mean / running_var (in/out, may be NULL), inv_count = 1 / count, unbias = count / (count - 1), eps, momentum
 *   backward  dgamma, dbeta (+= when accumulate), the rest unused
 *   nrows     partial rows of the launch; C channels; ncl = ceil(nrows / 32) */
typedef struct {
    uint32_t* tickets;
    double* cl_sums;
    float* totals;
    float *mean, *invstd, *running_mean, *running_var;
    float *dgamma, *dbeta;
    double inv_count, unbias;
    float eps, momentum;
    int32_t accumulate, nrows, C, ncl;
} cdet_bn_fold;
#define CDET_BN_FOLD_TICKET_WORDS (8 * (1 + 128))
int64_t cdet_bn_fold_cl_doubles(int32_t nrows, int32_t C);
int cdet_conv2d_tiled_bn_ok(const cdet_conv_desc* d);      /* 1: the launch's rows (<= 4096) and 160-cout blocks (<= 8) fit the fold */
int cdet_conv2d_tiled_bn(const cdet_conv_desc* d, const void* x, const void* w_tiled, void* y, float* stats, const cdet_bn_fold* fold_dev,
                         void* stream);
int cdet_conv2d_s2_tiled_bn_ok(const cdet_conv_desc* d);
int cdet_conv2d_s2_tiled_bn(const cdet_conv_desc* d, const void* x, const void* w_tiled, void* y, float* stats, const cdet_bn_fold* fold_dev,
                            void* stream);
/* Running-statistics updates of many layers from their fp32 totals in ONE launch (the later task's deferred updates of a shared block, see
 * engine.Plan.deferred_stats): per item exactly cdet_bn_finalize(totals, nblk = 1, ...)'s running_mean / running_var arithmetic. */
typedef struct {
    const float* totals;           /* [2][C] */
    float *running_mean, *running_var;
    double inv_count, unbias;
    float momentum;
    int32_t C;
} cdet_bn_running_item;
int cdet_bn_running_update(const cdet_bn_running_item* items_dev, int32_t n_items, int32_t max_C, void* stream);
/* y = silu(gamma*(z-mean)*invstd + beta) (+ residual); z,y [M, C] with strides. */
int cdet_bn_silu_fwd(const void* z, int32_t z_ld, int32_t z_coff, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, const void* residual, int32_t res_ld, int32_t res_coff,
                     void* y, int32_t y_ld, int32_t y_coff, int64_t M, int32_t C, int32_t dtype, void* stream);
/* Backward, pass 1: dact = dy * silu'(a), a = gamma*xhat+beta; partial sums of dact and dact*xhat per channel
 * -> part[(blk*2+{0,1})*C + c], blk < cdet_bn_bwd_blocks(M). */
int cdet_bn_bwd_blocks(int64_t M);
int cdet_bn_silu_bwd_reduce(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                            const float* mean, const float* invstd, const float* gamma, const float* beta,
                            float* part, int64_t M, int32_t C, int32_t dtype, void* stream);
/* The same pass with the reduction of its partial rows folded in (cdet_bn_fold above, backward form): fold_dev->totals receives the sums
 * [2C] that cdet_bn_silu_bwd_apply takes with nblk = 0, dgamma / dbeta are written by the last arriver. */
int cdet_bn_silu_bwd_reduce_fold(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 float* part, int64_t M, int32_t C, int32_t dtype, const cdet_bn_fold* fold_dev, void* stream);
/* pass 2: reduce partials -> dgamma(+)=, dbeta(+)=; dz = gamma*invstd*(dact - mean(dact) - xhat*mean(dact*xhat)).
 * `part` must hold (nblk*2*C + 2*C) floats: the reduced sums are stored behind the partials. */
int cdet_bn_silu_bwd_apply(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* part, int32_t nblk, float* dgamma, float* dbeta, int32_t accumulate,
                           void* dz, int32_t dz_ld, int32_t dz_coff, int64_t M, int32_t C, int32_t dtype, int64_t count, void* stream);
/* The same pass with the Bottleneck shortcut folded in (models/common.py:107-117, x + cv2(cv1(x))): dy is the gradient of the sum, so
 * besides dz the pass accumulates it into the shortcut addend's gradient slice: also[r][c] += dy[r][c] (also: [M, C] 16-bit with strides;
 * replaces a separate cdet_add_channels over the same tensors). */
int cdet_bn_silu_bwd_apply_add(const void* dy, int32_t dy_ld, int32_t dy_coff, const void* z, int32_t z_ld, int32_t z_coff,
                               const float* mean, const float* invstd, const float* gamma, const float* beta,
                               const float* part, int32_t nblk, float* dgamma, float* dbeta, int32_t accumulate,
                               void* dz, int32_t dz_ld, int32_t dz_coff, int64_t M, int32_t C, int32_t dtype, int64_t count,
                               void* also, int32_t also_ld, int32_t also_coff, void* stream);
/* SyncBatchNorm support (train.py:140-143 of the reference): reduce [nblk][2][C] partials (conv statistics or the backward
 * partials) to sums[2C] so that the host can all-reduce them between kernels; then cdet_bn_finalize(sums, nblk = 1, count =
 * global count) / cdet_bn_silu_bwd_apply(part = sums, nblk = 0, count = global count). `count` <= 0 means M. */
int cdet_bn_bwd_sums(const float* part, int32_t nblk, int32_t C, float* sums, float* dgamma, float* dbeta, int32_t accumulate,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-movement ops of the graph
 * ---------------------------------------------------------------------------------------------- */
/* dst[:, coff:coff+C] (=|+=) src[:, :C]  -- Concat (common.py:288-295) / chunk views / gradient fan-in. */
int cdet_copy_channels(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                       int64_t M, int32_t C, int32_t dtype, int32_t accumulate, void* stream);
/* nn.Upsample(None, 2, 'nearest') (model YAML neck rows) fused with the Concat that always follows it. */
int cdet_upsample2(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* backward of the above: dsrc[n,y,x,c] (=|+=) sum of the 4 children */
int cdet_upsample2_bwd(const void* ddst, int32_t ddst_ld, int32_t ddst_coff, void* dsrc, int32_t dsrc_ld,
                       int32_t dsrc_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate,
                       void* stream);
/* SPPF pooling (common.py:237,243-245): buf[:, coff + (i+1)*C : ...] = maxpool5x5s1p2^(i+1)(buf[:, coff : coff+C]), i=0..2 */
int cdet_sppf_pool(void* buf, int32_t ld, int32_t coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                   void* stream);
/* backward through the three chained pools: dbuf[:, coff:coff+C] += routed gradients of slices 1..3 */
int cdet_sppf_pool_bwd(const void* buf, void* dbuf, int32_t ld, int32_t coff, int32_t N, int32_t H, int32_t W, int32_t C,
                       int32_t dtype, void* stream);
/* ---- fp32-accurate eval path (csrc/precise.hip, cerberusdet_amd/precise.py; round 5) --------------------------------------------------
 * The reference evaluates an fp32 model in fp32 (cerberusdet/models/cerberus.py:804-882 on a `.float()` model). Here an fp32 map is carried as
 * three bf16 terms t = hi + mid + lo (exact to 2^-24 |t|) beside its fp32 value; a convolution is the six term pairs above 2^-24 accumulated in
 * fp32 by cdet_conv2d_tiled / cdet_conv2d_s2_tiled (fp32 destination, accumulate), and the three kernels below do everything between two
 * convolutions in fp32. All maps NHWC [N, H, W, ld] with a channel slice [coff, coff + C); a map's three term buffers share its geometry.
 *
 * cdet_split3: dst pixel (n, y, x) <- src pixel (n, y >> upsample, x >> upsample): copy / Concat slice / nn.Upsample(None, 2, 'nearest')
 * (models/common.py:288-295 and the neck rows of the model YAML) / the NCHW input image (src_nchw = 1: src is [N, C, H, W]). src_dtype F32 / BF16 / F16, or
 * U8 (value / 255 in fp32: the reference's preprocess_batch);
 * dst_f32 may be null (only the terms are wanted). */
int cdet_split3(const void* src, int32_t src_dtype, int32_t src_ld, int32_t src_coff, int32_t src_nchw, int32_t upsample, float* dst_f32, void* dst_hi,
                void* dst_mid, void* dst_lo, int32_t dst_ld, int32_t dst_coff, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* y = act(z * scale[c] + bias[c]) + res, fp32 throughout (scale / bias / res may be null): folded BatchNorm + SiLU + Bottleneck shortcut of
 * models/common.py:51-68, 107-117, or the bias of the head's projections (models/yolo.py:82-84). Writes y and / or its three terms. */
int cdet_epilogue_f32(const float* z, int32_t z_ld, int32_t z_coff, const float* scale, const float* bias, int32_t act, const float* res, int32_t res_ld,
                      int32_t res_coff, float* y, void* y_hi, void* y_mid, void* y_lo, int32_t y_ld, int32_t y_coff, int64_t M, int32_t C, void* stream);
/* Train-form nn.BatchNorm2d of an fp32 map (models/common.py:57-62 in model.train()): batch statistics in double (fixed summation order) ->
 * scale[c] = gamma / sqrt(var + eps), bias[c] = beta - mean * scale for cdet_epilogue_f32, and the running statistics (momentum form, unbiased
 * variance; both null: untouched, as for a frozen block). ws: cdet_bn_train_f32_ws_doubles(C) doubles of scratch. */
int64_t cdet_bn_train_f32_ws_doubles(int32_t C);
int cdet_bn_train_f32(const float* z, int32_t z_ld, int32_t z_coff, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, double* ws, float* scale, float* bias, float* mean, float* invstd, void* stream);
/* Backward of y = SiLU(gamma * zhat + beta) with zhat = (z - mean) * invstd (the autograd of models/common.py:57-68 in train mode), all fp32, sums in
 * double with a fixed order: dgamma[c] += sum da * zhat, dbeta[c] += sum da (either may be null), dz = scale * (da - mean(da) - zhat * mean(da * zhat))
 * written as value and / or three terms (row length dz_ld, channel 0 first). z: the convolution's fp32 accumulator [M][z_ld]; scale / bias / mean /
 * invstd: what cdet_bn_train_f32 wrote in the forward; ws as there. */
int cdet_bn_silu_bwd_f32(const float* dy, int32_t dy_ld, int32_t dy_coff, const float* z, int32_t z_ld, const float* scale, const float* bias, const float* mean,
                         const float* invstd, int64_t M, int32_t C, double* ws, float* dgamma, float* dbeta, float* dz, void* dz_hi, void* dz_mid,
                         void* dz_lo, int32_t dz_ld, void* stream);
/* dst (=|+=) src on fp32 channel slices; downsum = 1: dst pixel (y, x) takes the sum of src pixels (2y..2y+1, 2x..2x+1) of a 2H x 2W source (backward of
 * nn.Upsample(None, 2, 'nearest')). N, H, W: destination geometry. Gradient fan-in of shortcuts / Concat slices / Upsample. */
int cdet_add_f32(const float* src, int32_t src_ld, int32_t src_coff, float* dst, int32_t dst_ld, int32_t dst_coff, int32_t N, int32_t H, int32_t W, int32_t C,
                 int32_t downsum, int32_t accumulate, void* stream);
/* out[c] += sum over M pixels of src[:, coff + c] (double, fixed order): bias gradient of the head's projections (models/yolo.py:82-84). ws: as
 * cdet_bn_train_f32. */
int cdet_colsum_f32(const float* src, int32_t ld, int32_t coff, int64_t M, int32_t C, double* ws, float* out, void* stream);
/* Backward of cdet_maxpool_f32: dx += dy routed to the FIRST maximum of every window (row-major scan, like nn.MaxPool2d). */
int cdet_maxpool_bwd_f32(const float* x, int32_t x_ld, int32_t x_coff, const float* dy, int32_t dy_ld, int32_t dy_coff, float* dx, int32_t dx_ld,
                         int32_t dx_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, void* stream);
/* nn.MaxPool2d(k, 1, k / 2) on an fp32 map (SPPF, models/common.py:174-191). */
int cdet_maxpool_f32(const float* x, int32_t x_ld, int32_t x_coff, float* y, void* y_hi, void* y_mid, void* y_lo, int32_t y_ld, int32_t y_coff, int32_t N,
                     int32_t H, int32_t W, int32_t C, int32_t k, void* stream);
/* out[c] (=|+=) sum_r src[r, coff + c], c < C_out <= C: bias gradient of the head's biased 1x1 projections (yolo.py:82-84).
 * `part`: fp32 scratch of cdet_bn_bwd_blocks(M) * C elements. */
int cdet_colsum(const void* src, int32_t ld, int32_t coff, int64_t M, int32_t C, int32_t C_out, int32_t dtype, float* out,
                int32_t accumulate, float* part, void* stream);
/* y (=|+=) a + b elementwise on [M,C] slices (Bottleneck residual backward fan-in, common.py:117) */
int cdet_add_channels(const void* a, int32_t a_ld, int32_t a_coff, const void* b, int32_t b_ld, int32_t b_coff,
                      void* y, int32_t y_ld, int32_t y_coff, int64_t M, int32_t C, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Detect head eval branch (models/yolo.py:93-100 + DFL 57-59 + utils/tal.py:181-205)
 * feats: 3 NHWC maps with channels [64 box-bin logits | nc class logits | layout padding], pixel stride f_ld, fp32/bf16/f16.
 * hw6 / strides3 are HOST arrays.
 * y: [N, 4+nc, A] (reference layout, A = sum H_i*W_i), out_dtype.
 * ---------------------------------------------------------------------------------------------- */
int cdet_detect_decode(const void* f0, const void* f1, const void* f2, const int32_t* hw6, const float* strides3,
                       int32_t N, int32_t nc, int32_t f_ld, int32_t dtype, void* y, int32_t out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss: TAL assignment + BCE + CIoU + DFL, forward AND gradient w.r.t. the raw head maps in one pass
 * (utils/loss.py:133-181, utils/tal.py:56-178, utils/metrics.py:373-412).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, nc, n_max;          /* batch, classes, padded GT count per image                      */
    int32_t hw[6];                 /* h0,w0,h1,w1,h2,w2                                              */
    float stride[3];
    float gain_box, gain_cls, gain_dfl;
    float grad_scale;              /* d(scalar)/d(feat) is multiplied by this (loss weight * world)  */
    int32_t dtype;                 /* dtype of the head maps (CDET_F32 / BF16 / F16)                 */
    int32_t grad_dtype;            /* dtype of the gradient maps df*                                 */
    int32_t f_ld;                  /* channel stride of a map pixel: >= 64+nc (layout padding, the   */
                                   /* pad channels get zero gradient)                                */
    int32_t topk;                  /* 10                                                             */
    float alpha, beta;             /* 0.5, 6.0                                                       */
    const float* grad_scale_dev;   /* NULL, or a DEVICE float the gradient is ALSO multiplied by: the GradScaler's scale, read when the */
                                   /* kernel runs (scaler.scale(loss).backward(), averaging.py:158) -- the loss items are not scaled   */
} cdet_loss_desc;
/* Loss.preprocess (utils/loss.py:111-124): n label rows -> gt [N, n_max, 5] fp32 (cls, x1, y1, x2, y2 px), zero rows = padding.
 * batch_idx / cls: [n] fp32, bboxes: [n, 4] fp32 normalised (cx, cy, w, h). Labels keep their order inside an image. A label
 * that does not fit n_max is dropped and counted in *dropped (device int32, may be NULL; never reset here). */
int cdet_pad_targets(const float* batch_idx, const float* cls, const float* bboxes, int32_t n, int32_t N, int32_t n_max,
                     float img_w, float img_h, float* gt, int32_t* dropped, void* stream);
int64_t cdet_det_loss_ws_bytes(const cdet_loss_desc* d);
/* gt: [N, n_max, 5] fp32 (cls, x1, y1, x2, y2 in pixels; rows with box sum <= 0 are padding, loss.py:155);
 * n_max >= 1 (pass one all-zero row per image when the batch has no labels).
 * out_loss[0..3] = (box, cls, dfl, box+cls+dfl) with gains applied; out_loss[4] = 2*N*total (loss.py:179-181).
 * assign outputs (optional, may be NULL): fg_mask u8 [N,A], target_gt_idx i32 [N,A], target_labels i32 [N,A],
 * target_bboxes f32 [N,A,4] (pixels), target_scores f32 [N,A,nc]. dfeat*: same shape/dtype as feats (may be NULL). */
int cdet_det_loss(const cdet_loss_desc* d, const void* f0, const void* f1, const void* f2, const float* gt,
                  void* df0, void* df1, void* df2, float* out_loss5, uint8_t* fg_mask, int32_t* target_gt_idx,
                  int32_t* target_labels, float* target_bboxes, float* target_scores, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batched NMS (utils/general.py:360-481 incl. torchvision.ops.nms at :464) for ALL images in one call.
 * pred: [N, 4+nc, A] (xywh + class scores), dtype. out_rows: [N, max_det, 6] fp32, out_count: [N] i32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, nc, A;
    int32_t dtype;
    float conf_thres, iou_thres;
    int32_t agnostic, multi_label, max_det;
    int32_t max_nms;               /* 30000 (general.py:414)                                         */
    int32_t max_cand;              /* capacity of the candidate buffers per image (<= A or A*nc)     */
    const int32_t* classes;        /* optional device pointer to class filter list                   */
    int32_t n_classes;
} cdet_nms_desc;
int64_t cdet_nms_ws_bytes(const cdet_nms_desc* d);
int cdet_nms_batched(const cdet_nms_desc* d, const void* pred, float* out_rows, int32_t* out_count, void* ws, void* stream);
/* The same, and out_anchor [N, max_det] i32 (may be NULL) = the anchor index every kept row came from: what the reference's mask
 * branch (general.py:410,443-449, `nm` > 0) needs to carry the mask coefficients x[:, 4+nc:] of the kept boxes along. */
int cdet_nms_batched_idx(const cdet_nms_desc* d, const void* pred, float* out_rows, int32_t* out_count, int32_t* out_anchor, void* ws,
                         void* stream);

/* Cross-task merge after the per-task NMS -- everything between non_max_suppression and the result dicts of
 * CerberusDetInference.predict (cerberusdet_inference.py:72-83, 140-177; utils/general.py:484-554 nms_between_tasks, 313-357
 * scale_boxes / clip_boxes): local class ids -> global ids (task-order offsets), rows grouped by task, IoU between boxes of
 * DIFFERENT tasks only (box_iou of utils/metrics.py:415-433, eps 1e-7), greedy row scan: a live row with hits keeps only the
 * best-scoring box of hits U {row} (first maximum in the order [hits ascending, row]); "nothing survives" keeps everything,
 * as the reference does. Then optionally scale_boxes(net -> original shape) and round-half-even, all in fp32 like torch.
 *   rows[t]   [N, max_det, 6] fp32 (x1,y1,x2,y2,conf,local cls) and counts[t] [N] i32: outputs of cdet_nms_batched per task
 *   scale     [N, 5] fp32 (gain, pad_x, pad_y, h0, w0) or NULL (boxes stay in network coordinates, not rounded)
 *   out_rows  [N, T*max_det, 6] fp32 (global class ids), out_count [N] i32.  T <= 8, T*max_det <= 2048. */
typedef struct {
    int32_t N, T, max_det;
    float iou_thres;
    const float* rows[8];
    const int32_t* counts[8];
    int32_t cls_offset[8];
} cdet_merge_desc;
int cdet_merge_tasks(const cdet_merge_desc* d, const float* scale, float* out_rows, int32_t* out_count, void* stream);

/* Validation matcher, batched (val.py:32-54 process_batch + utils/metrics.py:415-433 box_iou): which predictions count as correct
 * at T IoU levels. Every prediction proposes its best class-matching label (highest IoU; exact IoU ties -> the higher label
 * index); at level t a label accepts, among the predictions proposing it with IoU >= iouv[t], the one with the LOWEST index.
 *   det_rows [N, max_det, 6] fp32 (x1,y1,x2,y2,conf,cls) + det_count [N] i32   (cdet_nms_batched / cdet_merge_tasks output)
 *   labels [L,5] fp32 (cls,x1,y1,x2,y2), grouped by image: image n owns rows label_start[n] .. label_start[n+1]-1
 *   iouv [T] fp32, T <= 16; max_labels = upper bound of labels per image (sizes the LDS table, max_labels*T*4 <= 160 KiB)
 *   correct [N, max_det, T] uint8 (rows >= det_count[n] are zeroed) */
typedef struct {
    int32_t N, max_det, T, max_labels;
} cdet_match_desc;
int cdet_match_predictions(const cdet_match_desc* d, const float* det_rows, const int32_t* det_count, const float* labels,
                           const int32_t* label_start, const float* iouv, uint8_t* correct, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor optimizer step (trainers/averaging.py:205-223, utils/torch_utils.py:302-312)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float* p; float* g; float* mom; float* ema;   /* ema may be NULL */
    int64_t n;
    int32_t group;                                /* index into the step's lr array (the reference's 3 param groups) */
    float weight_decay, inv_div;                  /* inv_div = 1 / #tasks served by the block */
    int32_t first_step;                           /* momentum buffer uninitialised            */
} cdet_param_slot;
/* GradScaler state (torch.cuda.amp.GradScaler as the reference drives it, trainers/averaging.py:61, 158, 207, 219-220), DEVICE float[4]:
 * {scale, growth_tracker, skipped steps so far, found_inf of the last step}. Every `scaler` argument below may be NULL (no loss scaling: the
 * found-inf SKIP still applies -- it is decided by the norm alone). The scale multiplies the loss gradient (cdet_loss_desc.grad_scale_dev). */
/* sum of squares of all UNSCALED gradients (g / scale) -> out[0]; `out` must hold 1 + 32*n_slots floats (block partials, fixed-order reduce).
 * A slot with g == NULL takes part in the EMA only (BatchNorm running statistics). A non-finite gradient anywhere makes out[0] non-finite:
 * the found-inf flag of the step. */
int cdet_grad_sqnorm(const cdet_param_slot* slots_dev, int32_t n_slots, float* out, const float* scaler, void* stream);
/* unscale, clip by global norm (coef = min(1, max_norm/(sqrt(*sqnorm)+1e-6))), per-block division, SGD-nesterov, EMA, zero grad.
 * *sqnorm not finite (found_inf): the step is SKIPPED like scaler.step() does -- weights and momentum keep their bits, the gradients are zeroed,
 * the EMA lerp still runs (the reference calls ema.update regardless, averaging.py:222-223).
 * lrs: HOST array of n_groups (<= 4) learning rates, passed by value with the launch -- the slot table does not change while the
 * warm-up / schedule moves the rates (trainers/averaging.py:160-180 of the reference recomputes them every iteration). */
int cdet_sgd_ema_step(const cdet_param_slot* slots_dev, int32_t n_slots, const float* sqnorm, float max_norm,
                      const float* lrs, int32_t n_groups, float momentum, float ema_decay, const float* scaler, float* skip_count, void* stream);
/* skip_count (may be NULL): DEVICE float[2] = {skipped steps so far, found_inf of this step}, written by this launch -- the bookkeeping of plans
 * WITHOUT loss scaling (bf16: scale fixed at 1), which then need no cdet_scaler_update launch; pass elements 2..3 of the scaler state. */
/* scaler.update() (averaging.py:220): found_inf (*sqnorm not finite) -> scale *= backoff_factor, tracker = 0, skipped += 1; otherwise tracker += 1
 * and scale *= growth_factor every growth_interval good steps (GradScaler defaults 2.0 / 0.5 / 2000, init scale 65536). growth_interval <= 0: the
 * scale stays fixed and only the skip counter / found_inf flag are kept (bf16 plans). Enqueue BEHIND the update launches of the step. */
int cdet_scaler_update(float* scaler, const float* sqnorm, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream);
/* dst[i] += src[i]; src[i] = 0 (fp32, 16-byte aligned). The reference accumulates the gradients of a block that several tasks share in one
 * .grad tensor, task after task (trainers/averaging.py:140-160: backward per task, one optimizer step). Here every task but the first owns a
 * gradient bucket of its own for those blocks, so that the task passes are independent launch streams; the buckets are folded into the block's
 * bucket in task order before the norm / all-reduce -- the same sums (fp32 addition of the same two addends). */
int cdet_accumulate_clear(float* dst, float* src, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SyncBatchNorm statistics exchange by peer writes (csrc/peer_exchange.hip, round 4; reference train.py:140-143 converts every BatchNorm to
 * SyncBatchNorm, i.e. one small all-reduce per layer and direction). One exchange buffer per rank, mapped by all peers of the node through HIP IPC:
 *   cdet_peer_alloc / _free          device buffer for the exchange (uncached where the runtime offers it), zeroed
 *   cdet_peer_export / _import       64-byte IPC handle of a buffer / map a peer's buffer (cdet_peer_close unmaps)
 *   cdet_peer_allreduce              vec[0..n) <- sum over ranks, in rank order (bit-identical on every rank): one single-workgroup kernel that writes
 *                                    the rank's row into every peer's slot, publishes an epoch flag, waits for all ranks' flags in its own buffer
 *                                    and sums. The wait is bounded in wall time (CDET_PEER_SPIN_MS, default 10 min): on time-out vec is filled with
 *                                    NaN and *err keeps 1 + the first missing rank (sticky: never cleared by a later launch). peer_table: device array of `world` buffer addresses (own included);
 *                                    the slot = 2 x world x n floats at data_off (floats) + 2 x world flags at flag_off (32-bit words); epoch > 0
 *                                    grows by one per use of the slot. phase 0 = the whole exchange; 1 / 2 = its publish / collect halves as separate
 *                                    launches (ranks sharing one GPU are time-sliced, not concurrent: they need a host barrier between the halves).
 * ---------------------------------------------------------------------------------------------- */
int cdet_peer_alloc(int64_t bytes, void** out);
int cdet_peer_free(void* p);
int cdet_peer_export(void* p, void* handle64);
int cdet_peer_import(const void* handle64, void** out);
int cdet_peer_close(void* p);
int cdet_peer_allreduce(float* vec, int32_t n, const void* peer_table, int32_t world, int32_t rank, int64_t data_off, int64_t flag_off, uint32_t epoch,
                        void* err, int32_t phase, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CERBERUS_HIP_H */
