"""ORACLE (test infrastructure only) -- CPU restatement of the reference's pre-processing:
letterbox + BGR->RGB + /255 + (x-mean)/std + HWC->CHW  (/root/reference/tools/infer.py:121-131, 446-453).

``cv2.resize(..., interpolation=cv2.INTER_LINEAR)`` is third-party (opencv-python>=4.9,
/root/reference/requirements.txt:5; absent from this image).  Its 8-bit bilinear path is restated
here FROM RECOLLECTION of OpenCV's ``resize.cpp`` (fixed-point, INTER_RESIZE_COEF_BITS = 11):

    scale_x = w0 / nw (double)          fx = float((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx
    sx < 0 -> (sx, fx) = (0, 0)         sx >= w0-1 -> (sx, fx) = (w0-1, 0)
    alpha = (round_half_even((1-fx)*2048), round_half_even(fx*2048))  as int16        (same for y: beta)
    horizontal pass (int32):  H[dx] = S[sx]*alpha0 + S[min(sx+1, w0-1)]*alpha1
    vertical pass   (uint8):  (((beta0 * (H0 >> 4)) >> 16) + ((beta1 * (H1 >> 4)) >> 16) + 2) >> 2

PARITY UNPINNED: no reference test, no cv2 here.  The HIP kernel is held bit-exact to THIS statement.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def letterbox_geometry(h: int, w: int, new_size: int):
    """tools/infer.py:121-131: (scale, nh, nw, top, left); Python round() = round-half-even."""
    scale = min(new_size / h, new_size / w)
    nh, nw = int(round(h * scale)), int(round(w * scale))
    return scale, nh, nw, (new_size - nh) // 2, (new_size - nw) // 2


def _coeffs(src: int, dst: int):
    scale = src / dst                                          # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0; s[hi] = src - 1
    a1 = np.rint(f * np.float32(2048)).astype(np.int32)        # cvRound: half to even
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
    return s, a0, a1


def resize_linear_u8(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    h0, w0 = img.shape[:2]
    if (h0, w0) == (nh, nw):
        return img.copy()
    sx, ax0, ax1 = _coeffs(w0, nw)
    sy, by0, by1 = _coeffs(h0, nh)
    src = img.astype(np.int32)
    x1 = np.minimum(sx + 1, w0 - 1)
    H = src[:, sx, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]      # [h0, nw, 3] int32
    y1 = np.minimum(sy + 1, h0 - 1)
    H0, H1 = H[sy], H[y1]                                                           # [nh, nw, 3]
    out = (((by0[:, None, None] * (H0 >> 4)) >> 16) + ((by1[:, None, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img_bgr: np.ndarray, new_size: int = 640, color=(114, 114, 114)):
    h, w = img_bgr.shape[:2]
    scale, nh, nw, top, left = letterbox_geometry(h, w, new_size)
    r = resize_linear_u8(img_bgr, nw, nh)
    out = np.empty((new_size, new_size, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + nh, left:left + nw] = r
    return out, scale, (left, top)


def preprocess(img_bgr: np.ndarray, new_size: int):
    """-> (x [3,S,S] fp32 normalised RGB, (padx, pady, scale, w0, h0))   tools/infer.py:446-453."""
    lb, scale, (padx, pady) = letterbox(img_bgr, new_size)
    im = lb[..., ::-1].astype(np.float32) / 255.0
    im = (im - MEAN) / STD
    return np.ascontiguousarray(im.transpose(2, 0, 1)), (padx, pady, scale, img_bgr.shape[1], img_bgr.shape[0])


def preprocess_albumentations(img_bgr: np.ndarray, new_size: int, resize: bool = False):
    """The evaluate path (tools/evaluate.py:57-72 -> scripts/data/augment.py:153-171, images read by
    scripts/data/dataset.py:88-92 as RGB): A.Resize(p=resize) | A.LongestMaxSize(S) + A.PadIfNeeded(S, S, constant
    114, centred: top = int((S - nh) / 2.0)) -> the geometry of letterbox() above -- then A.Normalize restated FROM
    RECOLLECTION of albumentations' functional.normalize (albumentations is third-party, un-pinned in
    requirements.txt and absent here: PARITY UNPINNED):
        mean = float32(mean) * 255; std = float32(std) * 255; denominator = np.reciprocal(std, dtype=float32)
        img = img.astype(float32); img -= mean; img *= denominator
    -> (x [3,S,S] fp32, (padx, pady, scale, w0, h0))."""
    h, w = img_bgr.shape[:2]
    if resize:
        lb, scale, padx, pady = resize_linear_u8(img_bgr, new_size, new_size), min(new_size / h, new_size / w), 0, 0
    else:
        lb, scale, (padx, pady) = letterbox(img_bgr, new_size)
    mean = MEAN * np.float32(255.0)
    den = np.reciprocal(STD * np.float32(255.0), dtype=np.float32)
    im = lb[..., ::-1].astype(np.float32)
    im -= mean
    im *= den
    return np.ascontiguousarray(im.transpose(2, 0, 1)), (padx, pady, scale, w, h)
