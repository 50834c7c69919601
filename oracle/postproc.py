"""ORACLE (test infrastructure only) -- CPU restatement of the reference post-processing.

Reference sources restated here:
  decode                 /root/reference/scripts/helpers/utils_ms.py:26-123   (pinned, fixtures)
  score + threshold      /root/reference/tools/infer.py:463-475 (main), :310-327 (fallback),
                         /root/reference/scripts/helpers/helpers.py:106-123 (eval)
  per-class NMS loops    tools/infer.py:476-493, :356-366 ; helpers.py:126-136
  greedy fallback NMS    tools/infer.py:134-163                                (pinned, fixtures)
  global top-k           tools/infer.py:368-379
  xyxy->"xywh" (sic)     helpers.py:58-83
  back-map               tools/infer.py:508-516
  torchvision.ops.nms    third-party (torchvision>=0.17, requirements.txt:2; absent here).
                         Restated from its published CPU kernel (``nms_kernel_impl``):
                         areas=(x2-x1)*(y2-y1); order=stable sort desc; for each unsuppressed i,
                         every later j with inter/(area_i+area_j-inter) > thr is suppressed, where
                         inter=max(0,xx2-xx1)*max(0,yy2-yy1).  PARITY UNPINNED for this primitive.

All arithmetic is fp32 in the same association order as the reference expressions.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------ decode
@torch.no_grad()
def decode_levels(levels: Sequence[torch.Tensor], img_size: int, center_mode: str = "v8",
                  wh_mode: str = "softplus") -> Dict[str, torch.Tensor]:
    """utils_ms.py:26-123.  levels: list of [B,A,S,S,5+C] (or [B,S,S,5+C]).
    Returns box [B,N,4] (xyxy px, clamped to [0,img_size-1]), obj [B,N,1] logits, cls [B,N,C] logits."""
    levels = list(levels) if isinstance(levels, (list, tuple)) else [levels]
    boxes, objs, clss = [], [], []
    for t in levels:
        if t.dim() == 4:
            t = t.unsqueeze(1)
        B, A, S, _, D = t.shape
        stride = img_size / float(S)                                     # :71
        ar = torch.arange(S, device=t.device)
        gy, gx = torch.meshgrid(ar, ar, indexing="ij")
        gx, gy = gx.view(1, 1, S, S), gy.view(1, 1, S, S)
        sx, sy = torch.sigmoid(t[..., 0]), torch.sigmoid(t[..., 1])
        if center_mode == "v8":                                          # :83-85
            px = ((sx * 2.0 - 0.5) + gx) * stride
            py = ((sy * 2.0 - 0.5) + gy) * stride
        else:                                                            # :86-88
            px = (sx + gx) * stride
            py = (sy + gy) * stride
        tw, th = t[..., 2], t[..., 3]
        if wh_mode == "v8":                                              # :91-93
            pw = (torch.sigmoid(tw) * 2.0).pow(2.0) * stride
            ph = (torch.sigmoid(th) * 2.0).pow(2.0) * stride
        elif wh_mode == "softplus":                                      # :94-96
            pw = F.softplus(tw) * stride
            ph = F.softplus(th) * stride
        else:                                                            # :97-99
            pw = tw.clamp(-4, 4).exp() * stride
            ph = th.clamp(-4, 4).exp() * stride
        hi = img_size - 1
        xyxy = torch.stack([(px - pw * 0.5).clamp(0, hi), (py - ph * 0.5).clamp(0, hi),
                            (px + pw * 0.5).clamp(0, hi), (py + ph * 0.5).clamp(0, hi)], dim=-1)
        n = A * S * S
        boxes.append(xyxy.reshape(B, n, 4))
        objs.append(t[..., 4].reshape(B, n, 1))
        clss.append(t[..., 5:].reshape(B, n, D - 5))
    return {"box": torch.cat(boxes, 1), "obj": torch.cat(objs, 1), "cls": torch.cat(clss, 1)}


@torch.no_grad()
def decode_unclamped_image(levels, b: int, img_size, center_mode="v8", wh_mode="softplus"):
    """Pre-clamp centre/size, objectness and class logits of image ``b`` (tools/infer.py:268-308).
    Sliced per image and per level exactly like the reference (``p_b = pred[b]``): torch's
    vectorised sigmoid differs by 1 ulp between its vector body and scalar tail, so the slicing
    is part of the bit-exact contract.  Returns px,py,pw,ph,obj_logit [N] and cls_logit [N,C]."""
    out = [[], [], [], [], [], []]
    for t in (list(levels) if isinstance(levels, (list, tuple)) else [levels]):
        p = t[b] if t.dim() == 5 else t[b].unsqueeze(0)                  # [A,S,S,D]
        S = p.shape[1]
        cell = img_size / S
        ar = torch.arange(S, device=t.device)
        gy, gx = torch.meshgrid(ar, ar, indexing="ij")
        gx, gy = gx.float(), gy.float()
        if center_mode == "v8":
            px = ((torch.sigmoid(p[..., 0]) * 2.0 - 0.5) + gx) * cell
            py = ((torch.sigmoid(p[..., 1]) * 2.0 - 0.5) + gy) * cell
        else:
            px = (torch.sigmoid(p[..., 0]) + gx) * cell
            py = (torch.sigmoid(p[..., 1]) + gy) * cell
        if wh_mode == "v8":
            pw = (torch.sigmoid(p[..., 2]) * 2).pow(2) * cell
            ph = (torch.sigmoid(p[..., 3]) * 2).pow(2) * cell
        elif wh_mode == "softplus":
            pw = F.softplus(p[..., 2]) * cell
            ph = F.softplus(p[..., 3]) * cell
        else:
            pw = p[..., 2].clamp(-4, 4).exp() * cell
            ph = p[..., 3].clamp(-4, 4).exp() * cell
        obj = torch.sigmoid(p[..., 4])
        cls_p = torch.sigmoid(p[..., 5:])
        for lst, v in zip(out, (px, py, pw, ph, obj)):
            lst.append(v.reshape(-1))
        out[5].append(cls_p.reshape(-1, p.shape[-1] - 5))
    return [torch.cat(v, 0) for v in out]


# ------------------------------------------------------------------------------ score
@torch.no_grad()
def score_candidates(obj_logit: torch.Tensor, cls_logit: torch.Tensor, c1_uses_cls: bool = False):
    """obj_logit [N], cls_logit [N,C] -> (score [N] fp32, class [N] int64).
    C>1: score = sigmoid(obj) * max_c sigmoid(cls_c), class = first argmax.
    C==1: main/eval path score = sigmoid(obj) (tools/infer.py:470-472, helpers.py:113-115);
          fallback path (c1_uses_cls) score = sigmoid(obj)*sigmoid(cls_0) (tools/infer.py:316-320)."""
    obj = obj_logit.sigmoid()
    C = cls_logit.shape[-1]
    if C > 1:
        conf, idx = cls_logit.sigmoid().max(dim=-1)
        return obj * conf, idx
    idx = torch.zeros_like(obj, dtype=torch.long)
    if c1_uses_cls and C == 1:
        return obj * cls_logit.sigmoid().squeeze(-1), idx
    return obj, idx


# ------------------------------------------------------------------------------ NMS primitives
def _stable_desc_order(scores: np.ndarray) -> np.ndarray:
    # descending by score, ties keep the lower index first (stable)
    return np.argsort(-scores.astype(np.float32), kind="stable")


def nms_torchvision(boxes: np.ndarray, scores: np.ndarray, iou_thr: float) -> np.ndarray:
    """Restatement of torchvision.ops.nms (CPU kernel).  Returns kept indices, score-descending."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = _stable_desc_order(scores)
    suppressed = np.zeros(n, dtype=bool)
    thr = np.float32(iou_thr)
    keep = []
    zero = np.float32(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            if rest.size == 0:
                break
            xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(zero, xx2 - xx1); h = np.maximum(zero, yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr > thr]] = True                           # NaN > thr is False
    return np.asarray(keep, dtype=np.int64)


def nms_greedy_fallback(boxes: np.ndarray, scores: np.ndarray, iou_thr: float) -> np.ndarray:
    """tools/infer.py:139-149,155-163 -- pure-torch greedy NMS used when torchvision is missing:
    IoU = inter / (a1 + a2 - inter + 1e-6) with clamp(min=0) on both extents; keep iff IoU <= thr."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    idxs = _stable_desc_order(scores)
    thr = np.float32(iou_thr)
    eps = np.float32(1e-6)
    zero = np.float32(0)
    keep = []
    while idxs.size > 0:
        i = idxs[0]
        keep.append(i)
        if idxs.size == 1:
            break
        r = idxs[1:]
        bx = boxes[i]
        xa = np.maximum(bx[0], boxes[r, 0]); ya = np.maximum(bx[1], boxes[r, 1])
        xb = np.minimum(bx[2], boxes[r, 2]); yb = np.minimum(bx[3], boxes[r, 3])
        inter = np.maximum(xb - xa, zero) * np.maximum(yb - ya, zero)
        a1 = (bx[2] - bx[0]) * (bx[3] - bx[1])
        a2 = (boxes[r, 2] - boxes[r, 0]) * (boxes[r, 3] - boxes[r, 1])
        iou = inter / (a1 + a2 - inter + eps)
        idxs = r[iou <= thr]
    return np.asarray(keep, dtype=np.int64)


def nms(boxes, scores, iou_thr: float = 0.5, max_det: int = 300, impl: str = "torchvision") -> np.ndarray:
    """tools/infer.py:134-152: NMS then keep[:max_det]."""
    keep = nms_torchvision(boxes, scores, iou_thr) if impl == "torchvision" else \
        nms_greedy_fallback(boxes, scores, iou_thr)
    return keep[:max_det] if (max_det is not None and keep.size > max_det) else keep


def _per_class(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, iou_thr: float,
               cap, impl: str):
    ob, os_, oc = [], [], []
    for c in np.unique(classes):                                        # ascending, like Tensor.unique()
        m = classes == c
        bb, ss = boxes[m], scores[m]
        k = nms(bb, ss, iou_thr, cap, impl)
        if k.size:
            ob.append(bb[k]); os_.append(ss[k]); oc.append(np.full(k.size, int(c), np.int64))
    if not ob:
        return (np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), np.zeros((0,), np.int64))
    return np.concatenate(ob), np.concatenate(os_), np.concatenate(oc)


# ------------------------------------------------------------------------------ pipelines
@torch.no_grad()
def pipeline_main(levels, img_size: int, conf: float = 0.4, iou: float = 0.5, per_class_cap: int = 300,
                  nms_impl: str = "torchvision") -> Dict[str, List[np.ndarray]]:
    """tools/infer.py:460-493 (torchvision importable): decode -> score -> `> conf` -> per-class
    NMS with the nms() default cap of 300 per class -> concat in ascending class order."""
    dec = decode_levels(levels, img_size)
    out = {"boxes": [], "scores": [], "classes": []}
    for b in range(dec["box"].shape[0]):
        sc, ci = score_candidates(dec["obj"][b].squeeze(-1), dec["cls"][b])
        m = sc > conf
        bb, ss, cc = _per_class(dec["box"][b][m].numpy(), sc[m].numpy(), ci[m].numpy(), iou,
                                per_class_cap, nms_impl)
        out["boxes"].append(bb); out["scores"].append(ss); out["classes"].append(cc)
    return out


@torch.no_grad()
def pipeline_eval(levels, img_size: int, conf_th: float = 0.001, iou_th: float = 0.65, add_one: bool = True,
                  nms_impl: str = "torchvision"):
    """helpers.py:87-153: as main but no cap, then [cx,cy,w,h] (sic, helpers.py:78-83) and
    category_id = class + 1.  Returns (list[B] of list[dict], raw arrays per image)."""
    dec = decode_levels(levels, img_size)
    dets, raw = [], []
    for b in range(dec["box"].shape[0]):
        sc, ci = score_candidates(dec["obj"][b].squeeze(-1), dec["cls"][b])
        m = sc > conf_th
        bb, ss, cc = _per_class(dec["box"][b][m].numpy(), sc[m].numpy(), ci[m].numpy(), iou_th, None, nms_impl)
        raw.append((bb, ss, cc))
        w = np.maximum(bb[:, 2] - bb[:, 0], np.float32(0))
        h = np.maximum(bb[:, 3] - bb[:, 1], np.float32(0))
        cx = bb[:, 0] + np.float32(0.5) * w
        cy = bb[:, 1] + np.float32(0.5) * h
        dets.append([{"category_id": int(c) + (1 if add_one else 0),
                      "bbox": [float(a), float(b_), float(c_), float(d)], "score": float(s)}
                     for a, b_, c_, d, s, c in zip(cx, cy, w, h, ss, cc)])
    return dets, raw


@torch.no_grad()
def pipeline_fallback(levels, img_size: int, conf_th: float = 0.35, iou_th: float = 0.60, topk: int = 300,
                      center_mode: str = "v8", wh_mode: str = "softplus", nms_impl: str = "fallback") -> Dict[str, List[np.ndarray]]:
    """tools/infer.py:247-389: score (C==1 -> obj*cls), `> conf`, min-side >= 2 px on the
    pre-clamp size, clamp, per-class NMS through nms() (cap 300), global top-k by score.
    nms_impl: "fallback" = the pure-torch greedy loop of nms() (:139-149; what runs when torchvision is absent,
    the only situation in which the reference's CLI reaches this function), "torchvision" = nms() with the
    import succeeding (:136-137)."""
    lv = list(levels) if isinstance(levels, (list, tuple)) else [levels]
    out = {"boxes": [], "scores": [], "classes": []}
    hi = img_size - 1
    for b in range(lv[0].shape[0]):
        px, py, pw, ph, obj, cls_p = decode_unclamped_image(lv, b, img_size, center_mode, wh_mode)
        C = cls_p.shape[-1]
        if C > 1:
            conf, ci = cls_p.max(dim=-1)
            sc = obj * conf
        elif C == 1:                                                     # :316-320 obj * cls
            sc, ci = obj * cls_p.squeeze(-1), torch.zeros_like(obj, dtype=torch.long)
        else:
            sc, ci = obj, torch.zeros_like(obj, dtype=torch.long)
        m = (sc > conf_th) & (pw >= 2.0) & (ph >= 2.0)
        x, y, w, h = px[m], py[m], pw[m], ph[m]
        xyxy = torch.stack([x - w * 0.5, y - h * 0.5, x + w * 0.5, y + h * 0.5], 1).clamp(0, hi)
        bb, ss, cc = _per_class(xyxy.numpy().reshape(-1, 4), sc[m].numpy(), ci[m].numpy(), iou_th, 300, nms_impl)
        if ss.size > topk:
            top = _stable_desc_order(ss)[:topk]
            bb, ss, cc = bb[top], ss[top], cc[top]
        out["boxes"].append(bb); out["scores"].append(ss); out["classes"].append(cc)
    return out


def backmap(boxes: np.ndarray, padx: float, pady: float, scale: float, w0: int, h0: int) -> np.ndarray:
    """tools/infer.py:508-516: remove letterbox padding, undo scale, clip to the original image."""
    b = np.array(boxes, dtype=np.float32, copy=True).reshape(-1, 4)
    b[:, [0, 2]] -= np.float32(padx)
    b[:, [1, 3]] -= np.float32(pady)
    b /= np.float32(max(scale, 1e-6))
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, w0 - 1)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, h0 - 1)
    return b


# ------------------------------------------------------------------------------ instance masks (build-defined)
@torch.no_grad()
def split_mask_levels(levels, num_classes: int):
    """rows [.., 5+C+NM] -> (detection levels [.., 5+C], coefficient levels [.., NM])."""
    return [t[..., :5 + num_classes] for t in levels], [t[..., 5 + num_classes:] for t in levels]


@torch.no_grad()
def masks_for(levels, protos: torch.Tensor, num_classes: int, img_size: int, keep_idx, boxes_letterbox, thr: float = 0.5):
    """BUILD-DEFINED (no reference code exists; parity unpinned).  For image b and its kept candidates
    keep_idx[b] (indices into the concatenated candidate axis) with decoded boxes in network-input pixels:
        m = sigmoid(coef . proto[b])  (proto [NM,PH,PW]);  mask = (m > thr) inside the box scaled to the
    prototype grid (x1*PW/S <= x < x2*PW/S, same for y), 0 outside.  Returns list of uint8 [Ni,PH,PW]."""
    _, coef_lv = split_mask_levels(levels, num_classes)
    B, NM, PH, PW = protos.shape
    coefs = torch.cat([c.reshape(B, -1, NM) for c in coef_lv], 1)
    out = []
    xs = torch.arange(PW, dtype=torch.float32)[None, None, :]
    ys = torch.arange(PH, dtype=torch.float32)[None, :, None]
    for b in range(B):
        idx = torch.as_tensor(np.asarray(keep_idx[b], dtype=np.int64))
        if idx.numel() == 0:
            out.append(np.zeros((0, PH, PW), np.uint8))
            continue
        m = torch.sigmoid(torch.einsum("nk,kyx->nyx", coefs[b][idx], protos[b]))
        bx = torch.as_tensor(np.asarray(boxes_letterbox[b], dtype=np.float32)).reshape(-1, 4)
        x1, y1 = bx[:, 0] * (PW / img_size), bx[:, 1] * (PH / img_size)
        x2, y2 = bx[:, 2] * (PW / img_size), bx[:, 3] * (PH / img_size)
        inside = (xs >= x1[:, None, None]) & (xs < x2[:, None, None]) & (ys >= y1[:, None, None]) & (ys < y2[:, None, None])
        out.append(((m > thr) & inside).numpy().astype(np.uint8))
    return out


@torch.no_grad()
def masks_image_for(levels, protos: torch.Tensor, num_classes: int, img_size: int, keep_idx, det_boxes, out_hw,
                    backmap=None, thr: float = 0.5):
    """BUILD-DEFINED image-resolution masks (include/yololite_hip.h: yl_masks_image; no reference code exists, parity
    unpinned).  For image b: kept candidates keep_idx[b], their OUTPUT boxes det_boxes[b] (back-mapped when `backmap`
    = [(padx, pady, scale, w0, h0)] is given), output grid out_hw[b] = (h, w).  float32 throughout:
        xs = (x + 0.5) * scale + padx;  u = max(max(xs, 0) * (PW / S) - 0.5, 0);  u0 = min(floor(u), PW-1),
        u1 = min(u0 + 1, PW-1), lu = u - u0  (rows likewise);  m = lerp(lerp(p00, p01, lu), lerp(p10, p11, lu), lv)
        with p = sigmoid(coef . proto);  mask = m > thr inside x1 <= x < x2, y1 <= y < y2.
    Returns list of uint8 [Ni, h, w]."""
    _, coef_lv = split_mask_levels(levels, num_classes)
    B, NM, PH, PW = protos.shape
    coefs = torch.cat([c.reshape(B, -1, NM) for c in coef_lv], 1)
    f32 = np.float32
    out = []
    for b in range(B):
        h, w = int(out_hw[b][0]), int(out_hw[b][1])
        idx = torch.as_tensor(np.asarray(keep_idx[b], dtype=np.int64))
        if idx.numel() == 0:
            out.append(np.zeros((0, h, w), np.uint8))
            continue
        padx, pady, sc = (f32(0), f32(0), f32(1)) if backmap is None else (f32(backmap[b][0]), f32(backmap[b][1]), f32(backmap[b][2]))
        prob = torch.sigmoid(torch.einsum("nk,kyx->nyx", coefs[b][idx], protos[b])).numpy()      # [N, PH, PW]

        def taps(n, pad, P):
            c = (np.arange(n, dtype=f32) + f32(0.5)) * sc + pad
            u = np.maximum(np.maximum(c, f32(0)) * (f32(P) / f32(img_size)) - f32(0.5), f32(0))
            i0 = np.minimum(np.floor(u).astype(np.int64), P - 1)
            return i0, np.minimum(i0 + 1, P - 1), (u - i0.astype(f32)).astype(f32)
        u0, u1, lu = taps(w, padx, PW)
        v0, v1, lv_ = taps(h, pady, PH)
        p00, p01 = prob[:, v0][:, :, u0], prob[:, v0][:, :, u1]
        p10, p11 = prob[:, v1][:, :, u0], prob[:, v1][:, :, u1]
        top = p00 + (p01 - p00) * lu[None, None, :]
        bot = p10 + (p11 - p10) * lu[None, None, :]
        m = top + (bot - top) * lv_[None, :, None]
        bx = np.asarray(det_boxes[b], dtype=f32).reshape(-1, 4)
        xs, ys = np.arange(w, dtype=f32)[None, None, :], np.arange(h, dtype=f32)[None, :, None]
        inside = (xs >= bx[:, 0, None, None]) & (xs < bx[:, 2, None, None]) & (ys >= bx[:, 1, None, None]) & (ys < bx[:, 3, None, None])
        out.append(((m > f32(thr)) & inside).astype(np.uint8))
    return out
