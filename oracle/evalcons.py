"""TEST INFRASTRUCTURE -- CPU restatement of the reference's evaluate-path consumers (SURVEY.md 8(f)
row f3).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates, loop for loop (python floats / numpy float32 exactly as the reference computes them):

  build_curves_from_coco     scripts/data/p_r_f1.py:6-162      (summary dict, no files)
  create_confusion_matrix    scripts/helpers/evaluate.py:59-238 (matrix + per-class stats, no plot)

PINNED: tests/golden/eval_consumers.json was produced by running the reference's own two functions
(tests/golden/make_eval_fixtures.py; seaborn is the only stubbed import and only draws) -- this
restatement is checked against it in tests/test_oracle_golden.py.
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np


# ------------------------------------------------------------------------------ p_r_f1.py
def iou_xywh(a, b):
    """p_r_f1.py:31-41."""
    ax, ay, aw, ah = a
    bx, by, bw, bh = b
    ax2, ay2 = ax + aw, ay + ah
    bx2, by2 = bx + bw, by + bh
    ix1, iy1 = max(ax, bx), max(ay, by)
    ix2, iy2 = min(ax2, bx2), min(ay2, by2)
    iw, ih = max(0.0, ix2 - ix1), max(0.0, iy2 - iy1)
    inter = iw * ih
    ua = max(0.0, aw * ah) + max(0.0, bw * bh) - inter
    return inter / ua if ua > 0 else 0.0


def _greedy(preds, gts, flags, iou):
    """One score-ordered pass over `preds` of a key (p_r_f1.py:64-78 / :107-118): list of 0/1."""
    out = []
    for d in preds:
        best_j, best_iou = -1, 0.0
        for j, g in enumerate(gts):
            if flags[j]:
                continue
            v = iou_xywh(d["bbox"], g)
            if v > best_iou:
                best_iou, best_j = v, j
        if best_iou >= iou and best_j >= 0:
            flags[best_j] = True
            out.append(1)
        else:
            out.append(0)
    return out


def build_curves_from_coco(coco_images, coco_anns, coco_dets, out_dir=None, iou=0.50, steps=201):
    """p_r_f1.py:6-162 -- same summary dict (numpy arrays for the curves)."""
    gt_index = {}
    for a in coco_anns:                                                     # :43-49
        gt_index.setdefault((int(a["image_id"]), int(a["category_id"])), []).append(a["bbox"])
    matched_flags = {k: np.zeros(len(v), dtype=bool) for k, v in gt_index.items()}
    total_gt = sum(len(v) for v in gt_index.values())
    dets_sorted = sorted(coco_dets, key=lambda x: float(x.get("score", 0.0)), reverse=True)   # :56

    tps, fps = [], []
    for d in dets_sorted:                                                   # :58-78
        key = (int(d["image_id"]), int(d["category_id"]))
        gts = gt_index.get(key, [])
        if len(gts) == 0:
            fps.append(1.0); tps.append(0.0)
            continue
        hit = _greedy([d], gts, matched_flags[key], iou)[0]
        tps.append(float(hit)); fps.append(float(1 - hit))
    if len(tps) == 0:                                                       # :80-89
        return {"iou": float(iou), "best_f1": 0.0, "best_conf": 0.0, "precision_at_best": 0.0,
                "recall_at_best": 0.0}
    tps = np.array(tps); fps = np.array(fps)
    cum_tp, cum_fp = np.cumsum(tps), np.cumsum(fps)
    recalls_rank = cum_tp / max(1, total_gt)                                # :94
    precisions_rank = cum_tp / np.maximum(1, cum_tp + cum_fp)               # :95

    det_index = {}
    for d in coco_dets:                                                     # :100-103
        det_index.setdefault((int(d["image_id"]), int(d["category_id"])), []).append(d)
    confs = np.linspace(0.0, 1.0, steps)
    P_curve, R_curve, F1_curve = [], [], []
    for thr in confs:                                                       # :109-127
        TP = FP = 0
        for key, gts in gt_index.items():
            preds = [d for d in det_index.get(key, []) if float(d.get("score", 0.0)) >= thr]
            preds.sort(key=lambda x: float(x.get("score", 0.0)), reverse=True)
            hits = _greedy(preds, gts, np.zeros(len(gts), dtype=bool), iou)
            TP += sum(hits); FP += len(hits) - sum(hits)
        FN = total_gt - TP
        P = TP / (TP + FP) if (TP + FP) > 0 else 0.0
        R = TP / (TP + FN) if (TP + FN) > 0 else 0.0
        F1 = 2 * P * R / (P + R) if (P + R) > 0 else 0.0
        P_curve.append(P); R_curve.append(R); F1_curve.append(F1)
    P_curve = np.array(P_curve); R_curve = np.array(R_curve); F1_curve = np.array(F1_curve)
    best_idx = int(np.argmax(F1_curve))
    fixed_conf = 0.50                                                       # :132
    idx = int(np.argmin(np.abs(confs - fixed_conf)))
    return {
        "iou": float(iou), "best_f1": float(F1_curve[best_idx]), "best_conf": float(confs[best_idx]),
        "precision_at_best": float(P_curve[best_idx]), "recall_at_best": float(R_curve[best_idx]),
        "fixed_conf": fixed_conf, "precision_at_fixed_conf": float(P_curve[idx]),
        "recall_at_fixed_conf": float(R_curve[idx]), "f1_at_fixed_conf": float(F1_curve[idx]),
        "P_curve": P_curve, "R_curve": R_curve, "F1_curve": F1_curve, "confs": confs, "best_idx": best_idx,
        # not in the reference's summary (computed and dropped there, :94-95); kept for the parity tests
        "recalls_rank": recalls_rank, "precisions_rank": precisions_rank,
    }


# ------------------------------------------------------------------------------ evaluate.py
def xywh_to_xyxy(box):
    """evaluate.py:23-25: python-float sums rounded to float32."""
    x, y, w, h = box
    return np.array([x, y, x + w, y + h], dtype=np.float32)


def iou_matrix(boxes1, boxes2):
    """evaluate.py:27-57 (float32 throughout)."""
    if len(boxes1) == 0 or len(boxes2) == 0:
        return np.zeros((len(boxes1), len(boxes2)), dtype=np.float32)
    b1, b2 = boxes1[:, None, :], boxes2[None, :, :]
    ix1 = np.maximum(b1[..., 0], b2[..., 0]); iy1 = np.maximum(b1[..., 1], b2[..., 1])
    ix2 = np.minimum(b1[..., 2], b2[..., 2]); iy2 = np.minimum(b1[..., 3], b2[..., 3])
    iw = np.clip(ix2 - ix1, a_min=0, a_max=None); ih = np.clip(iy2 - iy1, a_min=0, a_max=None)
    inter = iw * ih
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    union = area1[:, None] + area2[None, :] - inter
    union = np.clip(union, a_min=1e-6, a_max=None)
    return inter / union


def confusion_matrix_counts(coco_anns, coco_dets, num_classes, iou_thresh=0.5, score_thresh=0.20):
    """The counting part of create_confusion_matrix (evaluate.py:80-156): int64 [(C+1),(C+1)],
    row = true class, column = predicted class, last index = background."""
    bg = num_classes
    cat_id_to_idx = {cid: cid - 1 for cid in range(1, num_classes + 1)}
    anns_by_img, dets_by_img = defaultdict(list), defaultdict(list)
    for a in coco_anns:
        anns_by_img[a["image_id"]].append(a)
    for d in coco_dets:
        dets_by_img[d["image_id"]].append(d)
    cm = np.zeros((num_classes + 1, num_classes + 1), dtype=np.int64)
    for img_id, gts in anns_by_img.items():                                  # images WITH ground truth only
        dets = [d for d in dets_by_img.get(img_id, []) if d.get("score", 0.0) >= score_thresh]
        dets = sorted(dets, key=lambda d: d.get("score", 0.0), reverse=True)
        gt_boxes = np.array([xywh_to_xyxy(a["bbox"]) for a in gts], dtype=np.float32)
        gt_labels = np.array([cat_id_to_idx[a["category_id"]] for a in gts], dtype=np.int64)
        det_boxes = (np.array([xywh_to_xyxy(d["bbox"]) for d in dets], dtype=np.float32)
                     if dets else np.zeros((0, 4), dtype=np.float32))
        det_labels = (np.array([cat_id_to_idx[d["category_id"]] for d in dets], dtype=np.int64)
                      if dets else np.zeros((0,), dtype=np.int64))
        matched = np.zeros(len(gts), dtype=bool)
        ious = iou_matrix(det_boxes, gt_boxes) if len(dets) > 0 and len(gts) > 0 else \
            np.zeros((len(dets), len(gts)), dtype=np.float32)
        for di in range(len(dets)):
            row = ious[di]
            bj = int(np.argmax(row))
            if row[bj] >= iou_thresh and not matched[bj]:
                cm[int(gt_labels[bj]), int(det_labels[di])] += 1
                matched[bj] = True
            else:
                cm[bg, int(det_labels[di])] += 1
        for gi, was in enumerate(matched):
            if not was:
                cm[int(gt_labels[gi]), bg] += 1
    return cm


def confusion_stats(cm):
    """evaluate.py:158-200: per-class TP / FP / FN / precision / recall from the raw matrix."""
    C = cm.shape[0] - 1
    tp = np.diag(cm)[:-1]
    fn = cm[:-1, C]
    fp = cm[C, :-1]
    prec = np.divide(tp, tp + fp, out=np.zeros_like(tp, dtype=float), where=(tp + fp) != 0)
    rec = np.divide(tp, tp + fn, out=np.zeros_like(tp, dtype=float), where=(tp + fn) != 0)
    return {"tp": tp, "fp": fp, "fn": fn, "precision": prec, "recall": rec,
            "total_fp": int(fp.sum()), "total_fn": int(fn.sum())}
