"""Post-processing with the reference's Python call surface, executed by the HIP kernels.

Mirrors:
  decode_preds_anchorfree(levels, img_size, center_mode, wh_mode) -> {"box","obj","cls"}
                                                /root/reference/scripts/helpers/utils_ms.py:26-123
  _decode_batch_to_coco_dets(preds, img_size, conf_th, iou_th, add_one) -> list[list[dict]]
                                                /root/reference/scripts/helpers/helpers.py:87-153
  decode_anchorfree_like_train(preds, img_size, conf_th, iou_th, topk, ...) -> {"boxes","scores","classes"}
                                                /root/reference/tools/infer.py:247-389
  nms(boxes, scores, iou_th, max_det)           /root/reference/tools/infer.py:134-152
  infer_main_postprocess(...)                   the inline main-path block tools/infer.py:460-516
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .model import HipContext

_CTX_CACHE: Dict[tuple, HipContext] = {}


def _levels_list(preds):
    lv = list(preds) if isinstance(preds, (list, tuple)) else [preds]
    return [t if t.dim() == 5 else t.unsqueeze(1) for t in lv]


def context_for(levels: Sequence[torch.Tensor], img_size: int, num_masks: int = 0) -> HipContext:
    """A (cached) post-processing-only context matching the level geometry (num_masks: trailing mask
    coefficients per row of a build-defined seg model)."""
    lv = _levels_list(levels)
    dev = lv[0].device if lv[0].is_cuda else torch.device("cuda", torch.cuda.current_device()
                                                           if torch.cuda.is_available() else 0)
    key = (int(img_size), int(lv[0].shape[-1]) - 5 - int(num_masks), tuple(int(t.shape[2]) for t in lv),
           tuple(int(t.shape[1]) for t in lv), dev.index or 0, int(num_masks))
    if key not in _CTX_CACHE:
        _CTX_CACHE[key] = HipContext(key[0], key[1], key[2], key[3], None, key[4], num_masks=key[5])
    return _CTX_CACHE[key]


@torch.no_grad()
def decode_preds_anchorfree(preds_levels, img_size: int, center_mode: str = "v8", wh_mode: str = "softplus"):
    lv = _levels_list(preds_levels)
    return context_for(lv, img_size).decode(lv, center_mode, wh_mode)


def _split(dets: torch.Tensor, counts: torch.Tensor, max_out: int):
    """device [B,max_out,6] + counts -> per-image host arrays (one D2H copy)."""
    cn = counts.cpu().numpy()
    if (cn > max_out).any():
        raise _lib.YoloLiteHipError(f"detections ({int(cn.max())}) exceed max_out ({max_out})")
    d = dets.cpu().numpy()
    return [d[b, :cn[b]] for b in range(d.shape[0])]


@torch.no_grad()
def decode_anchorfree_like_train(preds, img_size: int, conf_th: float = 0.35, iou_th: float = 0.60, topk: int = 300,
                                 center_mode: str = "v8", wh_mode: str = "softplus",
                                 nms_impl: str = "torchvision") -> Dict[str, List[torch.Tensor]]:
    """nms_impl: the primitive behind the reference's nms() (tools/infer.py:134-152) -- "torchvision" (what it
    runs in an install that has torchvision) or "greedy" (its pure-torch loop, IoU + 1e-6, taken when the import
    fails; the reference's CLI only reaches this function in that situation)."""
    lv = _levels_list(preds)
    ctx = context_for(lv, img_size)
    max_out = ctx.default_max_out(_lib.POST_FALLBACK, 300, topk)
    dets, counts = ctx.postprocess(lv, _lib.POST_FALLBACK, conf_th, iou_th, per_class_cap=300, topk=topk,
                                   max_out=max_out, center_mode=center_mode, wh_mode=wh_mode,
                                   fallback_nms=_lib.NMS_TORCHVISION if nms_impl == "torchvision" else _lib.NMS_GREEDY)
    rows = _split(dets, counts, max_out)
    dev = lv[0].device
    return {"boxes": [torch.from_numpy(r[:, :4].copy()).to(dev) for r in rows],
            "scores": [torch.from_numpy(r[:, 4].copy()).to(dev) for r in rows],
            "classes": [torch.from_numpy(r[:, 5].astype(np.int64)).to(dev) for r in rows]}


def _rows_to_coco(rows, add_one=True):
    """packed detection rows (x1,y1,x2,y2,score,class) per image -> the reference's COCO dicts (helpers.py:139-151)"""
    out = []
    for r in rows:
        # helpers.py:58-83 `_xyxy_to_xywh` returns [cx, cy, w, h] (sic)
        w = np.maximum(r[:, 2] - r[:, 0], np.float32(0))
        h = np.maximum(r[:, 3] - r[:, 1], np.float32(0))
        cx = r[:, 0] + np.float32(0.5) * w
        cy = r[:, 1] + np.float32(0.5) * h
        cid = r[:, 5].astype(np.int64) + (1 if add_one else 0)
        out.append([{"category_id": int(c), "bbox": [float(a), float(b), float(c_), float(d)], "score": float(s)}
                    for a, b, c_, d, s, c in zip(cx, cy, w, h, r[:, 4], cid)])
    return out


@torch.no_grad()
def _decode_batch_to_coco_dets(preds, img_size, conf_th=0.001, iou_th=0.65, add_one=True):
    lv = _levels_list(preds)
    ctx = context_for(lv, img_size)
    dets, counts = ctx.postprocess(lv, _lib.POST_EVAL, conf_th, iou_th, per_class_cap=0, topk=0, max_out=ctx.N)
    return _rows_to_coco(_split(dets, counts, ctx.N), add_one)


@torch.no_grad()
def predict_coco_dets(ctx: HipContext, x: torch.Tensor, conf_th=0.001, iou_th=0.65, add_one=True):
    """model(x) + _decode_batch_to_coco_dets in ONE call (yl_predict: decode inside the head-output convs, no raw level
    tensors) -- what tools/evaluate.py runs; same rows as the two-call form (test_fused_decode_epilogue_...)."""
    dets, counts = ctx.predict(x, _lib.POST_EVAL, conf_th, iou_th, per_class_cap=0, topk=0, max_out=ctx.N)
    return _rows_to_coco(_split(dets, counts, ctx.N), add_one)


def _backmap_tensor(backmap):
    if backmap is None:
        return None
    arr = np.asarray(backmap, dtype=np.float64).reshape(len(backmap), 5).copy()
    arr[:, 2] = np.maximum(arr[:, 2], 1e-6)
    return torch.from_numpy(arr.astype(np.float32))


@torch.no_grad()
def predict_main(ctx: HipContext, x: torch.Tensor, conf: float = 0.4, iou: float = 0.5, per_class_cap: int = 300,
                 backmap: Optional[Sequence[Sequence[float]]] = None):
    """model(x) + infer_main_postprocess in ONE call (yl_predict) -- what tools/infer.py runs, the path bench.py
    measures.  Same return value as infer_main_postprocess."""
    max_out = ctx.default_max_out(_lib.POST_MAIN, per_class_cap, 0)
    dets, counts = ctx.predict(x, _lib.POST_MAIN, conf, iou, per_class_cap=per_class_cap, max_out=max_out,
                               backmap=_backmap_tensor(backmap))
    rows = _split(dets, counts, max_out)
    return {"boxes": [r[:, :4].copy() for r in rows], "scores": [r[:, 4].copy() for r in rows],
            "classes": [r[:, 5].astype(np.int64) for r in rows]}


@torch.no_grad()
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_th: float = 0.5, max_det: int = 300,
        impl: str = "torchvision") -> torch.Tensor:
    ctx = context_for([torch.empty((1, 1, 1, 1, 6), device=boxes.device if boxes.is_cuda else "cuda")], 64)
    k = ctx.nms(boxes, scores, iou_th, max_det, _lib.NMS_TORCHVISION if impl == "torchvision" else _lib.NMS_GREEDY)
    return k.to(boxes.device)


@torch.no_grad()
def infer_main_postprocess(preds, img_size: int, conf: float = 0.4, iou: float = 0.5, per_class_cap: int = 300,
                           backmap: Optional[Sequence[Sequence[float]]] = None, num_masks: int = 0):
    """The main-path block of tools/infer.py:460-516 for a whole batch: decode, score, `> conf`,
    per-class NMS (cap 300/class, the nms() default -- the CLI's --max_det is not forwarded there),
    optional back-map to the original image (padx, pady, scale, w0, h0 per image).
    Returns {"boxes","scores","classes"} lists of numpy arrays (classes int64)."""
    lv = _levels_list(preds)
    ctx = context_for(lv, img_size, num_masks)
    bm = _backmap_tensor(backmap)
    max_out = ctx.default_max_out(_lib.POST_MAIN, per_class_cap, 0)
    dets, counts = ctx.postprocess(lv, _lib.POST_MAIN, conf, iou, per_class_cap=per_class_cap, max_out=max_out, backmap=bm)
    rows = _split(dets, counts, max_out)
    return {"boxes": [r[:, :4].copy() for r in rows], "scores": [r[:, 4].copy() for r in rows],
            "classes": [r[:, 5].astype(np.int64) for r in rows]}
