"""`YoloLite(path).predict(...)` -- the call surface of the reference's pip package
(/root/reference/README.md:20-42, /root/reference/benchmark.py:73-129): one dict per image with
`boxes` (xyxy), `scores`, `classes`, `masks` (None for detectors) and `speed`.

Everything from the uint8 image bytes to the final detections runs in the HIP library: letterbox +
normalise (yl_preprocess, tools/infer.py:121-131,442-453), forward, decode, NMS, back-map (yl_predict).
There is ONE pre-processing implementation in the product (preprocess.preprocess_batch -> yl_preprocess); the CLIs
use it too."""
from __future__ import annotations

import time
from typing import List, Sequence, Union

import numpy as np
import torch

from . import _lib
from .model import load_model_names_imgsize_from_ckpt
from .preprocess import preprocess_batch

class YoloLite:
    def __init__(self, weights: str, device: Union[str, int] = "cuda:0"):
        dev = torch.device(device if isinstance(device, str) else f"cuda:{device}")
        self.model, self.names, self.img_size = load_model_names_imgsize_from_ckpt(weights, dev)
        self.device = dev
        # serving loop: the launches of a call are replayed from cached hipGraphs (keyed on the buffers of the call; a
        # batch-1 forward is 30-odd launches of a few microseconds each -- eager launch overhead would dominate it)
        # small batches are latency-bound: the 20x20-stage depthwise layers in their split-K form (-9 % batch-1 forward).
        # split_k is another fp32 summation order: detections of this API differ in low-order bits from YOLOLiteHIP(...) /
        # the CLIs on the same checkpoint (both are held to the oracle by the same tolerances).  Applied to every
        # context of the model (any input size), not only the checkpoint's img_size.
        self.model.set_context_options(graph=1, split_k=1)

    @torch.no_grad()
    def predict(self, source: Union[np.ndarray, Sequence[np.ndarray]], device=None, draw: bool = False,
                conf: float = 0.4, iou: float = 0.5, profile: bool = True) -> List[dict]:
        """source: one BGR uint8 image (HWC) or a sequence of them.  Main-path semantics of
        tools/infer.py:460-516 (conf 0.4 / iou 0.5 defaults :403-404, 300 per class).
        `speed` (ms per image, like the reference's pre / infer / post / total split, README.md:182-189):
        pre_ms = host packing + H2D copy + letterbox/normalise kernel (wall clock around a stream sync), infer_ms /
        post_ms = HIP-event intervals on the launch stream (yl_last_timing), total_ms = their sum.  profile=False
        skips the event split (the batch then runs as overlapped chunks) and reports infer_post_ms from the wall clock."""
        imgs = [source] if isinstance(source, np.ndarray) else list(source)
        t0 = time.perf_counter()
        ctx = self.model._ctx_for(self.img_size)
        x, bmap = preprocess_batch(ctx, imgs)                   # letterbox + normalise on the GPU
        bm = bmap.copy()
        bm[:, 2] = np.maximum(bm[:, 2], 1e-6)
        bm = bm.astype(np.float32)
        t1 = time.perf_counter()
        # "time_split" is a measurement aid on the model's SHARED context (single chunk, eager launches, no overlap):
        # set for this call only and put back, so that model(x) / ctx.predict callers never inherit it
        prev_split = ctx.get_option("time_split", 0)
        ctx.set_option("time_split", 1 if profile else 0)
        masks = None
        try:
            if ctx.NM:                                          # build-defined seg model: masks at ORIGINAL image resolution
                dets, counts, idx = ctx.predict(x, _lib.POST_MAIN, conf, iou, per_class_cap=300,
                                                backmap=torch.from_numpy(bm), want_idx=True)
                masks = [m.cpu().numpy() for m in ctx.masks_image(dets, counts, idx, backmap=torch.from_numpy(bm))]
            else:
                dets, counts = ctx.predict(x, _lib.POST_MAIN, conf, iou, per_class_cap=300, backmap=torch.from_numpy(bm))
            timing = ctx.last_timing() if profile else None
        finally:
            ctx.set_option("time_split", prev_split)
        cn = counts.cpu().numpy()
        d = dets.cpu().numpy()
        t2 = time.perf_counter()
        n = len(imgs)
        speed = {"pre_ms": (t1 - t0) * 1e3 / n}
        if profile:
            infer_ms, post_ms = timing
            speed.update(infer_ms=infer_ms / n, post_ms=post_ms / n)
            speed["total_ms"] = speed["pre_ms"] + speed["infer_ms"] + speed["post_ms"]
        else:
            speed.update(infer_post_ms=(t2 - t1) * 1e3 / n, total_ms=(t2 - t0) * 1e3 / n)
        out = []
        for b in range(n):
            r = d[b, :min(int(cn[b]), d.shape[1])]
            out.append({"boxes": r[:, :4].copy(), "scores": r[:, 4].copy(), "classes": r[:, 5].astype(np.int64),
                        "masks": (masks[b] if masks is not None else None), "speed": dict(speed)})
        return out
