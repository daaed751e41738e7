"""CPU oracle of the tile pre-processing that feeds the hot path.  TEST INFRASTRUCTURE ONLY: imported by
tests/ and tests/golden/make_golden.py, never by the product path (which is csrc/preprocess.hip and fails loudly without
the HIP library).

Reference: RoiBuilder.py:193-210 —
    train: ToPILImage -> Pad(100) -> RandomCrop(roi_size) -> Resize(resolution) -> RandomHorizontalFlip ->
           RandomVerticalFlip -> ToTensor -> Normalize((.5,.5,.5),(.5,.5,.5))
    flat:  ToPILImage -> Resize(resolution) -> ToTensor -> Normalize
The arithmetic lives in third-party code that is not part of /root/reference: torchvision's PIL backend (absent from this
image) and Pillow's `Image.resize(BILINEAR)` (Pillow 12.2.0 here; the reference pins no version).  Restated from Pillow's
published two-pass resampling (src/libImaging/Resample.c): support-scaled triangle filter, per-output-pixel normalised
float64 coefficients rounded to 22-bit fixed point, horizontal pass to uint8, vertical pass to uint8, each with a +0.5 ulp
bias and saturation.  PARITY PINNED for the arithmetic: tests/golden/prep_*.npz hold outputs of Pillow itself
(tests/golden/make_golden.py: run_prep_case) and this file reproduces them bit for bit.  The ORDER in which torchvision
draws the crop offsets and flip coins from the torch RNG is not pinned (torchvision is not importable here): the kernel takes
the drawn parameters as input.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_coeffs(in_size, out_size):
    """(bounds [out,2] int32 (first input index, count), kk [out,ksize] int32 fixed-point weights) of Pillow's bilinear
    resampling from `in_size` to `out_size` samples over the whole axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        kk[xx] = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS))).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    """uint8 [N, ...] -> uint8 [out_size, ...] along axis 0."""
    bounds, kk = resize_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for k in range(cnt):
            acc += src[xmin + k] * int(kk[xx, k])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow `Image.resize((out_w, out_h), BILINEAR)` of a uint8 [H,W,C] image: horizontal pass first, then vertical."""
    h, w, _ = img.shape
    t = img
    if out_w != w:
        t = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(t.transpose(1, 0, 2)), out_w).transpose(1, 0, 2))
    if out_h != h:
        t = _resample_axis0(t, out_h)
    return t


def to_tensor_normalize(img_u8):
    """ToTensor + Normalize(0.5, 0.5): uint8 [H,W,3] -> fp32 [3,H,W], computed in fp32 as torch does."""
    t = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return (t - np.float32(0.5)) / np.float32(0.5)


def finalize_tile(roi_u8, resolution, params=None, pad=100):
    """One tile through the chain.  params = (top, left, hflip, vflip) of the train-time transform, or None for the
    `img_finalize_flat` (validation) chain."""
    img = roi_u8
    if params is not None:
        top, left, hflip, vflip = (int(v) for v in params)
        s = roi_u8.shape[0]
        padded = np.zeros((s + 2 * pad, roi_u8.shape[1] + 2 * pad, 3), dtype=np.uint8)
        padded[pad:pad + s, pad:pad + roi_u8.shape[1]] = roi_u8
        img = padded[top:top + s, left:left + roi_u8.shape[1]]
    out = resize_bilinear_u8(img, resolution, resolution)
    if params is not None:
        if hflip:
            out = out[:, ::-1]
        if vflip:
            out = out[::-1]
    return to_tensor_normalize(np.ascontiguousarray(out))
