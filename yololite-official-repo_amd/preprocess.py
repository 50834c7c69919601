"""Batched GPU pre-processing: uint8 BGR HWC images -> letterboxed, normalised
[B,3,S,S] fp32 on the device with one H2D copy of the packed bytes and one kernel (yl_preprocess).

Mirrors letterbox() + the normalisation block of the reference (/root/reference/tools/infer.py:121-131,
446-453); the letterbox GEOMETRY is computed here exactly like the reference does (Python floats,
round-half-even), the pixels by the HIP kernel (OpenCV-style 11-bit fixed-point bilinear)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

_DESC = np.dtype([("offset", "<i8"), ("h0", "<i4"), ("w0", "<i4"), ("nh", "<i4"), ("nw", "<i4"),
                  ("top", "<i4"), ("left", "<i4")])
assert _DESC.itemsize == 32


def letterbox_geometry(h: int, w: int, new_size: int):
    """tools/infer.py:121-131 -> (scale, nh, nw, top, left)."""
    scale = min(new_size / h, new_size / w)
    nh, nw = int(round(h * scale)), int(round(w * scale))
    return scale, nh, nw, (new_size - nh) // 2, (new_size - nw) // 2


def preprocess_batch(ctx, images: Sequence[np.ndarray], letterbox: bool = True,
                     norm: str = "infer") -> Tuple[torch.Tensor, np.ndarray]:
    """images: BGR uint8 HWC arrays (any sizes).  Returns (x [B,3,S,S] fp32 on ctx.device,
    backmap [B,5] float64 = padx, pady, scale, w0, h0 as tools/infer.py:442-447 defines them).
    norm: "infer" = tools/infer.py:449-450 arithmetic; "albumentations" = the evaluate path's pipeline
    (tools/evaluate.py:57-72 -> scripts/data/augment.py:153-171: A.LongestMaxSize + A.PadIfNeeded produce the same
    geometry as letterbox(), A.Normalize a differently rounded normalisation; letterbox=False = its A.Resize(p=1))."""
    S = ctx.img_size
    ctx.set_option("pre_norm", 1 if norm == "albumentations" else 0)
    desc = np.zeros(len(images), _DESC)
    backmap = np.zeros((len(images), 5), np.float64)
    off = 0
    chunks: List[np.ndarray] = []
    for i, im in enumerate(images):
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError("images must be uint8 HxWx3 (BGR)")
        h, w = im.shape[:2]
        if letterbox:
            scale, nh, nw, top, left = letterbox_geometry(h, w, S)
        else:                                                   # --no_letterbox: plain resize (tools/infer.py:442-444)
            scale, nh, nw, top, left = min(S / h, S / w), S, S, 0, 0
        desc[i] = (off, h, w, nh, nw, top, left)
        backmap[i] = (left, top, scale, w, h)
        flat = np.ascontiguousarray(im).reshape(-1)
        chunks.append(flat)
        off += (flat.size + 15) & ~15                           # 16-byte aligned image starts
    packed = np.zeros(off, np.uint8)
    for d, c in zip(desc, chunks):
        packed[d["offset"]:d["offset"] + c.size] = c
    dev = ctx.device
    p_dev = torch.from_numpy(packed).to(dev, non_blocking=False)
    d_dev = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
    x = torch.empty((len(images), 3, S, S), device=dev, dtype=torch.float32)
    _lib.check(ctx.lib.yl_preprocess(ctx.handle, p_dev.data_ptr(), d_dev.data_ptr(), len(images), x.data_ptr(),
                                     int(torch.cuda.current_stream(dev).cuda_stream)), ctx.handle, "yl_preprocess")
    x.record_stream(torch.cuda.current_stream(dev))
    torch.cuda.current_stream(dev).synchronize()                # p_dev / d_dev may be freed after return
    return x, backmap
