"""CPU restatement (numpy) of the PIXEL side of the reference's training augmentation -- TEST INFRASTRUCTURE, never imported by the product.

Follows, image by image, what a reference DataLoader worker does with cv2 (data/datasets.py:470-477 load_image, 483-527 load_mosaic,
data/augmentations.py:152-155 cv2.warpPerspective / cv2.warpAffine inside random_perspective, 205-211 mixup, 43-57 augment_hsv, datasets.py:420-438 flips and the
final transpose), materialising every intermediate image the way the reference does -- the HIP kernel (csrc/augment.hip) computes the
same values on the fly and must equal this bit for bit.

cv2 is not installable here: its 8-bit algorithms are restated from OpenCV's published sources (resize.cpp INTER_LINEAR, imgwarp.cpp
warpAffine / remapBilinear, color_hsv.cpp RGB2HSV_b / HSV2RGB_f). **Parity unpinned** at the pixel level; the random parameters and the label
geometry are pinned separately against the reference's own functions (tests/golden/augment.json).
"""
import numpy as np

from .preprocess import resize_linear_u8


def resized_hw(hw0, s):  # datasets.py:470-477
    h0, w0 = hw0
    r = s / max(h0, w0)
    return (int(h0 * r), int(w0 * r)) if r != 1 else (h0, w0)


def build_canvas(tiles, images, canvas):
    """load_mosaic's img4 (canvas = 2s), or the letterboxed single image (canvas = s: letterbox's border colour is the same 114):
    tiles = [(index, (h, w) after the resize, (x1a, y1a, x2a, y2a), (x1b, y1b))], images[index] uint8 HWC BGR originals."""
    img4 = np.full((canvas, canvas, 3), 114, np.uint8)
    for idx, (h, w), (x1a, y1a, x2a, y2a), (x1b, y1b) in tiles:
        im = resize_linear_u8(images[idx], (w, h))
        img4[y1a:y2a, x1a:x2a] = im[y1b:y1b + (y2a - y1a), x1b:x1b + (x2a - x1a)]
    return img4


def _round_sat(x):
    return np.clip(np.rint(x), -2147483648.0, 2147483647.0).astype(np.int64)


def _remap_bilinear_u8(src, X, Y, border):
    """remapBilinear for 8-bit 3-channel images, BORDER_CONSTANT: X, Y fixed-point source coordinates with 5 fractional bits."""
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    H, W = src.shape[:2]

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        v[~ok] = border
        return v

    w00, w01, w10, w11 = (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32
    out = (tap(sx, sy) * w00[..., None] + tap(sx + 1, sy) * w01[..., None] + tap(sx, sy + 1) * w10[..., None] + tap(sx + 1, sy + 1) * w11[..., None] + (1 << 14)) >> 15
    return out.astype(np.uint8)


def invert3x3(M):
    """cv::invert of a 3x3 double matrix (the closed form OpenCV takes for sizes <= 3: determinant by the first row, cofactors / det)."""
    m = np.asarray(M, np.float64)
    d = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
         + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if d == 0:
        return np.zeros((3, 3))
    d = 1.0 / d
    return np.array([[(m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d, (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d, (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d],
                     [(m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d, (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d, (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d],
                     [(m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d, (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d, (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d]])


def warp_perspective_u8(src, M, dsize, border=114):
    """cv2.warpPerspective(src, M, dsize, borderValue=(114,)*3) (data/augmentations.py:152-153): INTER_LINEAR, BORDER_CONSTANT, 8-bit
    3-channel. OpenCV (imgwarp.cpp WarpPerspectiveInvoker) inverts M, walks the output in blocks 64 columns wide, evaluates the homogeneous
    coordinate of a block's first column in double, adds the column offset, divides per pixel (32 / w, 0 for w == 0), saturates and rounds
    to 5 fractional bits, then runs the same remapBilinear as warpAffine."""
    m = invert3x3(M).reshape(9)
    width, height = dsize
    xs = np.arange(width, dtype=np.int64)
    xb, x1 = ((xs // 64) * 64).astype(np.float64)[None, :], (xs % 64).astype(np.float64)[None, :]
    ys = np.arange(height, dtype=np.float64)[:, None]
    X0, Y0, W0 = m[0] * xb + m[1] * ys + m[2], m[3] * xb + m[4] * ys + m[5], m[6] * xb + m[7] * ys + m[8]
    W = W0 + m[6] * x1
    with np.errstate(divide="ignore"):
        W = np.where(W != 0, 32.0 / np.where(W != 0, W, 1.0), 0.0)
    fX = np.maximum(-2147483648.0, np.minimum(2147483647.0, (X0 + m[0] * x1) * W))
    fY = np.maximum(-2147483648.0, np.minimum(2147483647.0, (Y0 + m[3] * x1) * W))
    return _remap_bilinear_u8(src, _round_sat(fX), _round_sat(fY), border)


def warp_affine_u8(src, M, dsize, border=114):
    """cv2.warpAffine(src, M[:2], dsize, borderValue=(114,)*3): INTER_LINEAR, BORDER_CONSTANT, 8-bit 3-channel."""
    A = np.asarray(M, np.float64)[:2]
    D = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = A[1, 1] * D, A[0, 0] * D
    a12, a21 = -A[0, 1] * D, -A[1, 0] * D
    b1 = -a11 * A[0, 2] - a12 * A[1, 2]
    b2 = -a21 * A[0, 2] - a22 * A[1, 2]
    width, height = dsize
    xs, ys = np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64)
    adelta, bdelta = _round_sat(a11 * xs * 1024.0), _round_sat(a21 * xs * 1024.0)
    X0 = _round_sat((a12 * ys + b1) * 1024.0) + 16
    Y0 = _round_sat((a22 * ys + b2) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    return _remap_bilinear_u8(src, X, Y, border)


def bgr2hsv_u8(im):
    b, g, r = (im[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(b, np.maximum(g, r))
    diff = v - np.minimum(b, np.minimum(g, r))
    sdiv = np.where(v > 0, _round_sat((255 << 12) / np.maximum(v, 1).astype(np.float64)), 0)
    hdiv = np.where(diff > 0, _round_sat((180 << 12) / (6.0 * np.maximum(diff, 1))), 0)
    s = (diff * sdiv + (1 << 11)) >> 12
    vr, vg = v == r, v == g
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv + (1 << 11)) >> 12
    h = np.where(h < 0, h + 180, h)
    return h, s, v


def hsv2bgr_u8(h, s, v):
    f = np.float32
    sf, vf = s.astype(f) * f(1.0 / 255.0), v.astype(f) * f(1.0 / 255.0)
    hf = h.astype(f) * f(6.0 / 180.0)
    hf = np.where(hf >= f(6), hf - f(6), hf)
    sector = np.floor(hf).astype(np.int64)
    hf = (hf - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hf = np.where(bad, f(0), hf).astype(f)
    one = f(1)
    tab = np.stack((vf, (vf * (one - sf)).astype(f), (vf * (one - (sf * hf).astype(f)).astype(f)).astype(f),
                    (vf * (one - (sf * (one - hf).astype(f)).astype(f)).astype(f)).astype(f)), -1)
    sb, sg, sr = np.array([1, 1, 3, 0, 0, 2]), np.array([3, 0, 0, 2, 1, 1]), np.array([0, 2, 1, 1, 3, 0])
    pick = lambda sel: np.take_along_axis(tab, sel[sector][..., None], -1)[..., 0]  # noqa: E731
    b, g, r = pick(sb), pick(sg), pick(sr)
    gray = s == 0
    b, g, r = np.where(gray, vf, b), np.where(gray, vf, g), np.where(gray, vf, r)
    out = np.stack([np.clip(np.rint((c * f(255)).astype(f)), 0, 255) for c in (b, g, r)], -1)
    return out.astype(np.uint8)


def augment_hsv(im, lut):
    """augmentations.py:43-57 with the three lookup tables already drawn (lut [3, 256] uint8)."""
    h, s, v = bgr2hsv_u8(im)
    return hsv2bgr_u8(lut[0][h].astype(np.int64), lut[1][s].astype(np.int64), lut[2][v].astype(np.int64))


def render(mosaics, mix_ratio, lut, flipud, fliplr, images, s, perspective=False):
    """mosaics: [(tiles, M, canvas)] (one, or two with mixup) -> uint8 [3, s, s] RGB exactly as `__getitem__` returns it."""
    warp = warp_perspective_u8 if perspective else warp_affine_u8  # augmentations.py:152-155
    ims = [warp(build_canvas(tiles, images, canvas), M, (s, s)) for tiles, M, canvas in mosaics]
    im = ims[0]
    if len(ims) > 1:
        im = (im * mix_ratio + ims[1] * (1 - mix_ratio)).astype(np.uint8)
    if lut is not None:
        im = augment_hsv(im, lut)
    if flipud:
        im = np.flipud(im)
    if fliplr:
        im = np.fliplr(im)
    return np.ascontiguousarray(im.transpose((2, 0, 1))[::-1])
