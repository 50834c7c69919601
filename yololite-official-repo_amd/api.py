"""`YoloLite(path).predict(...)` -- the call surface of the reference's pip package
(/root/reference/README.md:20-42, /root/reference/benchmark.py:73-129): one dict per image with
`boxes` (xyxy), `scores`, `classes`, `masks` (None for detectors) and `speed`.

Everything from the uint8 image bytes to the final detections runs in the HIP library: letterbox +
normalise (yl_preprocess, tools/infer.py:121-131,442-453), forward, decode, NMS, back-map (yl_predict).
`preprocess_bgr` below is the host (numpy/PIL) variant kept for tests and tools."""
from __future__ import annotations

import time
from typing import List, Sequence, Union

import numpy as np
import torch

from . import _lib
from .model import load_model_names_imgsize_from_ckpt
from .preprocess import preprocess_batch

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _resize_bilinear_u8(im: np.ndarray, nw: int, nh: int) -> np.ndarray:
    if im.shape[0] == nh and im.shape[1] == nw:
        return im
    from PIL import Image
    return np.asarray(Image.fromarray(im).resize((nw, nh), Image.BILINEAR))


def letterbox(im: np.ndarray, new_size: int = 640, color=(114, 114, 114)):
    """tools/infer.py:121-131 (cv2.resize INTER_LINEAR replaced by PIL bilinear: cv2 is absent here)."""
    h, w = im.shape[:2]
    scale = min(new_size / h, new_size / w)
    nh, nw = int(round(h * scale)), int(round(w * scale))
    r = _resize_bilinear_u8(im, nw, nh)
    top, left = (new_size - nh) // 2, (new_size - nw) // 2
    out = np.empty((new_size, new_size, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + nh, left:left + nw] = r
    return out, scale, (left, top)


def preprocess_bgr(img_bgr: np.ndarray, img_size: int):
    """BGR uint8 HWC -> (normalised CHW fp32, (padx, pady, scale, w0, h0))  (tools/infer.py:446-453)."""
    lb, scale, (padx, pady) = letterbox(img_bgr, img_size)
    im = lb[..., ::-1].astype(np.float32) / 255.0
    im = (im - MEAN) / STD
    h0, w0 = img_bgr.shape[:2]
    return np.ascontiguousarray(im.transpose(2, 0, 1)), (padx, pady, scale, w0, h0)


class YoloLite:
    def __init__(self, weights: str, device: Union[str, int] = "cuda:0"):
        dev = torch.device(device if isinstance(device, str) else f"cuda:{device}")
        self.model, self.names, self.img_size = load_model_names_imgsize_from_ckpt(weights, dev)
        self.device = dev

    @torch.no_grad()
    def predict(self, source: Union[np.ndarray, Sequence[np.ndarray]], device=None, draw: bool = False,
                conf: float = 0.4, iou: float = 0.5) -> List[dict]:
        """source: one BGR uint8 image (HWC) or a sequence of them.  Main-path semantics of
        tools/infer.py:460-516 (conf 0.4 / iou 0.5 defaults :403-404, 300 per class)."""
        imgs = [source] if isinstance(source, np.ndarray) else list(source)
        t0 = time.perf_counter()
        ctx = self.model._ctx_for(self.img_size)
        x, bmap = preprocess_batch(ctx, imgs)                   # letterbox + normalise on the GPU
        bm = bmap.copy()
        bm[:, 2] = np.maximum(bm[:, 2], 1e-6)
        bm = bm.astype(np.float32)
        t1 = time.perf_counter()
        masks = None
        if ctx.NM:                                              # build-defined seg model: masks at prototype resolution
            dets, counts, idx = ctx.predict(x, _lib.POST_MAIN, conf, iou, per_class_cap=300,
                                            backmap=torch.from_numpy(bm), want_idx=True)
            masks = ctx.masks(counts, idx, dets.shape[1]).cpu().numpy()
        else:
            dets, counts = ctx.predict(x, _lib.POST_MAIN, conf, iou, per_class_cap=300, backmap=torch.from_numpy(bm))
        cn = counts.cpu().numpy()
        d = dets.cpu().numpy()
        t2 = time.perf_counter()
        out = []
        for b in range(len(imgs)):
            r = d[b, :min(int(cn[b]), d.shape[1])]
            out.append({"boxes": r[:, :4].copy(), "scores": r[:, 4].copy(), "classes": r[:, 5].astype(np.int64),
                        "masks": (masks[b, :len(r)].copy() if masks is not None else None),
                        "speed": {"pre_ms": (t1 - t0) * 1e3 / len(imgs), "infer_post_ms": (t2 - t1) * 1e3 / len(imgs),
                                  "total_ms": (t2 - t0) * 1e3 / len(imgs)}})
        return out
