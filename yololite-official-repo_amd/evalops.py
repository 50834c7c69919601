"""Evaluate-path consumers on the device behind the reference's own signatures.

    build_curves_from_coco(coco_images, coco_anns, coco_dets, out_dir, iou, steps)
        <- scripts/data/p_r_f1.py:6-162 (same summary dict)
    create_confusion_matrix(coco_anns, coco_dets, class_names, SAVE_PATH, ...)
        <- scripts/helpers/evaluate.py:59-238 (same *_stats.txt; returns the raw matrix as well;
           the heat-map PNG is not drawn)

The reference runs both as nested python loops over detections x ground truths -- and repeats the
P/R/F1 matching once per confidence step (201 times).  Here the host only groups and orders the rows
(numpy sort / unique); the matching runs in yl_eval_match / yl_eval_confusion, one wavefront per
(image, category) key or per image, and the 0..1 sweep is ONE matching pass plus a histogram
(yl_eval_sweep): greedy matching of a score-descending list is prefix-stable, so the per-threshold
re-matching of the reference yields exactly the prefix of the full pass.

No CPU fallback: the kernels live in libyololite_hip.so and a HIP device is required.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib


def _dev(device):
    if not torch.cuda.is_available():
        raise _lib.YoloLiteHipError("evalops needs a HIP device (no CPU fallback)")
    return torch.device(device if device is not None else "cuda:0")


def _up(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


# device time of the matching kernels (HIP events around every yl_eval_* launch), accumulated for the benchmark line:
# bench.py --workload eval reports how much of a step is kernels and how much host list -> array conversion
DEVICE_MS = {"total": 0.0, "launches": 0}


class _timed_launch:
    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        self.e1.record(torch.cuda.current_stream(self.dev))
        self.e1.synchronize()
        DEVICE_MS["total"] += self.e0.elapsed_time(self.e1)
        DEVICE_MS["launches"] += 1
        return False


def _offsets(sorted_keys, num_keys):
    off = np.zeros(num_keys + 1, dtype=np.int32)
    if len(sorted_keys):
        off[1:] = np.cumsum(np.bincount(sorted_keys, minlength=num_keys))
    return off


# ------------------------------------------------------------------------------ per-class matching
def match_per_class(det_img, det_cat, det_xywh, det_score, gt_img, gt_cat, gt_xywh, iou=0.5, device=None):
    """Greedy per-(image, category) matching of p_r_f1.py:58-78 for array inputs.
    Returns (tp uint8 [Nd] in the ORIGINAL detection order, has_gt bool [Nd], matched_gt bool [Ng])."""
    lib = _lib.load()
    dev = _dev(device)
    det_img = np.asarray(det_img, dtype=np.int64); det_cat = np.asarray(det_cat, dtype=np.int64)
    gt_img = np.asarray(gt_img, dtype=np.int64); gt_cat = np.asarray(gt_cat, dtype=np.int64)
    det_xywh = np.asarray(det_xywh, dtype=np.float64).reshape(-1, 4)
    gt_xywh = np.asarray(gt_xywh, dtype=np.float64).reshape(-1, 4)
    det_score = np.asarray(det_score, dtype=np.float64)
    nd, ng = len(det_img), len(gt_img)
    if nd == 0:
        return np.zeros(0, np.uint8), np.zeros(0, bool), np.zeros(ng, bool)
    pairs = np.concatenate([np.stack([det_img, det_cat], 1), np.stack([gt_img, gt_cat], 1)], 0)
    _, inv = np.unique(pairs, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    nk = int(inv.max()) + 1
    dkey, gkey = inv[:nd], inv[nd:]
    # key-major, score descending, original list order on ties (python's stable sorted(reverse=True))
    dorder = np.lexsort((np.arange(nd), -det_score, dkey))
    gorder = np.argsort(gkey, kind="stable")
    det_off = _offsets(dkey[dorder], nk)
    gt_off = _offsets(gkey[gorder], nk)

    t_det = _up(det_xywh[dorder], dev)
    t_doff, t_goff = _up(det_off, dev), _up(gt_off, dev)
    t_gt = _up(gt_xywh[gorder], dev) if ng else None
    t_tp = torch.empty(nd, dtype=torch.uint8, device=dev)
    t_gm = torch.empty(max(ng, 1), dtype=torch.uint8, device=dev)
    with _timed_launch(dev):
        _lib.check(lib.yl_eval_match(_ptr(t_det), _ptr(t_doff), _ptr(t_gt), _ptr(t_goff), nk, ng, float(iou),
                                     _ptr(t_tp), None, _ptr(t_gm), _stream(dev)), what="yl_eval_match")
    tp_sorted = t_tp.cpu().numpy()
    tp = np.empty(nd, np.uint8); tp[dorder] = tp_sorted
    has_gt = (gt_off[1:] - gt_off[:-1])[dkey] > 0
    gm = np.zeros(ng, bool)
    if ng:
        gm[gorder] = t_gm[:ng].cpu().numpy().astype(bool)
    return tp, has_gt, gm


def sweep_counts(score, tp, counted, thresholds, device=None):
    """TP / FP counts among `counted` detections with score >= thr, for every thr (p_r_f1.py:100-118)."""
    lib = _lib.load()
    dev = _dev(device)
    thr = np.asarray(thresholds, dtype=np.float64)
    steps = len(thr)
    n = len(score)
    t_thr = _up(thr, dev)
    t_tpg = torch.empty(steps, dtype=torch.int32, device=dev)
    t_fpg = torch.empty(steps, dtype=torch.int32, device=dev)
    t_s = _up(np.asarray(score, np.float64), dev) if n else None
    t_tp = _up(np.asarray(tp, np.uint8), dev) if n else None
    t_c = _up(np.asarray(counted, np.uint8), dev) if n else None
    with _timed_launch(dev):
        _lib.check(lib.yl_eval_sweep(_ptr(t_s), _ptr(t_tp), _ptr(t_c), n, _ptr(t_thr), steps, _ptr(t_tpg), _ptr(t_fpg),
                                     _stream(dev)), what="yl_eval_sweep")
    return t_tpg.cpu().numpy().astype(np.int64), t_fpg.cpu().numpy().astype(np.int64)


def build_curves_from_coco(coco_images, coco_anns, coco_dets, out_dir=None, iou=0.50, steps=201, device=None):
    """Drop-in for scripts/data/p_r_f1.py:6-162: same arguments, same summary dict (`out_dir` is
    accepted and, as in the reference, nothing is written by this function)."""
    total_gt = len(coco_anns)
    if len(coco_dets) == 0:                                                  # p_r_f1.py:80-89
        return {"iou": float(iou), "best_f1": 0.0, "best_conf": 0.0, "precision_at_best": 0.0,
                "recall_at_best": 0.0}
    d_img = np.array([int(d["image_id"]) for d in coco_dets], dtype=np.int64)
    d_cat = np.array([int(d["category_id"]) for d in coco_dets], dtype=np.int64)
    d_box = np.array([d["bbox"] for d in coco_dets], dtype=np.float64).reshape(-1, 4)
    d_sc = np.array([float(d.get("score", 0.0)) for d in coco_dets], dtype=np.float64)
    g_img = np.array([int(a["image_id"]) for a in coco_anns], dtype=np.int64)
    g_cat = np.array([int(a["category_id"]) for a in coco_anns], dtype=np.int64)
    g_box = np.array([a["bbox"] for a in coco_anns], dtype=np.float64).reshape(-1, 4)

    tp, has_gt, _ = match_per_class(d_img, d_cat, d_box, d_sc, g_img, g_cat, g_box, iou=iou, device=device)
    confs = np.linspace(0.0, 1.0, steps)
    tp_ge, fp_ge = sweep_counts(d_sc, tp, has_gt, confs, device=device)

    P_curve, R_curve, F1_curve = [], [], []
    for TP, FP in zip(tp_ge.tolist(), fp_ge.tolist()):                       # p_r_f1.py:120-124
        FN = total_gt - TP
        P = TP / (TP + FP) if (TP + FP) > 0 else 0.0
        R = TP / (TP + FN) if (TP + FN) > 0 else 0.0
        F1 = 2 * P * R / (P + R) if (P + R) > 0 else 0.0
        P_curve.append(P); R_curve.append(R); F1_curve.append(F1)
    P_curve = np.array(P_curve); R_curve = np.array(R_curve); F1_curve = np.array(F1_curve)
    best_idx = int(np.argmax(F1_curve))
    fixed_conf = 0.50
    idx = int(np.argmin(np.abs(confs - fixed_conf)))

    # score-ranked PR curve (p_r_f1.py:56-95; computed by the reference, not part of its summary)
    order = np.lexsort((np.arange(len(d_sc)), -d_sc))
    cum_tp = np.cumsum(tp[order].astype(np.float64))
    cum_fp = np.cumsum(1.0 - tp[order].astype(np.float64))
    return {
        "iou": float(iou), "best_f1": float(F1_curve[best_idx]), "best_conf": float(confs[best_idx]),
        "precision_at_best": float(P_curve[best_idx]), "recall_at_best": float(R_curve[best_idx]),
        "fixed_conf": fixed_conf, "precision_at_fixed_conf": float(P_curve[idx]),
        "recall_at_fixed_conf": float(R_curve[idx]), "f1_at_fixed_conf": float(F1_curve[idx]),
        "P_curve": P_curve, "R_curve": R_curve, "F1_curve": F1_curve, "confs": confs, "best_idx": best_idx,
        "recalls_rank": cum_tp / max(1, total_gt), "precisions_rank": cum_tp / np.maximum(1, cum_tp + cum_fp),
    }


# ------------------------------------------------------------------------------ confusion matrix
def confusion_matrix_counts(coco_anns, coco_dets, num_classes, iou_thresh=0.5, score_thresh=0.20, device=None):
    """Raw detection confusion matrix of evaluate.py:80-156: int64 [(C+1),(C+1)], row = true class,
    column = predicted class, last index = background.  category_id must be 1..C (KeyError otherwise,
    as in the reference)."""
    lib = _lib.load()
    dev = _dev(device)
    C = int(num_classes)

    def idx(cids):
        cids = np.asarray(cids, dtype=np.int64)
        bad = (cids < 1) | (cids > C)
        if bad.any():
            raise KeyError(int(cids[bad][0]))
        return (cids - 1).astype(np.int32)

    W = C + 1
    if len(coco_anns) == 0:
        return np.zeros((W, W), dtype=np.int64)
    g_img = np.array([a["image_id"] for a in coco_anns], dtype=np.int64)
    g_cls = idx([a["category_id"] for a in coco_anns])
    g_b = np.array([a["bbox"] for a in coco_anns], dtype=np.float64).reshape(-1, 4)
    img_ids, g_key = np.unique(g_img, return_inverse=True)                   # images WITH ground truth
    g_key = g_key.reshape(-1)
    ni = len(img_ids)
    gorder = np.argsort(g_key, kind="stable")

    def xyxy32(b):                                                           # evaluate.py:23-25
        return np.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1).astype(np.float32)

    if len(coco_dets):
        d_img = np.array([d["image_id"] for d in coco_dets], dtype=np.int64)
        d_sc = np.array([d.get("score", 0.0) for d in coco_dets], dtype=np.float64)
        pos = np.searchsorted(img_ids, d_img)
        pos_c = np.minimum(pos, ni - 1)
        keep = (img_ids[pos_c] == d_img) & (d_sc >= score_thresh)            # :101-104
        sel = np.nonzero(keep)[0]
    else:
        sel = np.zeros(0, dtype=np.int64)
    if len(sel):
        d_key = pos_c[sel]
        d_cls = idx([coco_dets[i]["category_id"] for i in sel])
        d_b = np.array([coco_dets[i]["bbox"] for i in sel], dtype=np.float64).reshape(-1, 4)
        dorder = np.lexsort((np.arange(len(sel)), -d_sc[sel], d_key))        # :107 stable, score descending
        det_off = _offsets(d_key[dorder], ni)
        t_det, t_dcls = _up(xyxy32(d_b[dorder]), dev), _up(d_cls[dorder], dev)
    else:
        det_off = np.zeros(ni + 1, dtype=np.int32)
        t_det = t_dcls = None
    gt_off = _offsets(g_key[gorder], ni)
    t_gt, t_gcls = _up(xyxy32(g_b[gorder]), dev), _up(g_cls[gorder], dev)
    t_doff, t_goff = _up(det_off, dev), _up(gt_off, dev)
    t_cm = torch.empty(W * W, dtype=torch.int32, device=dev)
    t_gm = torch.empty(len(g_img), dtype=torch.uint8, device=dev)
    with _timed_launch(dev):
        _lib.check(lib.yl_eval_confusion(_ptr(t_det), _ptr(t_dcls), _ptr(t_doff), _ptr(t_gt), _ptr(t_gcls), _ptr(t_goff),
                                         ni, len(g_img), C, float(np.float32(iou_thresh)), _ptr(t_cm), _ptr(t_gm),
                                         _stream(dev)), what="yl_eval_confusion")
    return t_cm.cpu().numpy().reshape(W, W).astype(np.int64)


def confusion_stats(cm):
    """evaluate.py:158-200."""
    C = cm.shape[0] - 1
    tp = np.diag(cm)[:-1]
    fn = cm[:-1, C]
    fp = cm[C, :-1]
    prec = np.divide(tp, tp + fp, out=np.zeros_like(tp, dtype=float), where=(tp + fp) != 0)
    rec = np.divide(tp, tp + fn, out=np.zeros_like(tp, dtype=float), where=(tp + fn) != 0)
    return {"tp": tp, "fp": fp, "fn": fn, "precision": prec, "recall": rec,
            "total_fp": int(fp.sum()), "total_fn": int(fn.sum())}


def create_confusion_matrix(coco_anns, coco_dets, class_names, SAVE_PATH, filename="confusion_matrix.png",
                            title="Detection Confusion Matrix", iou_thresh=0.5, score_thresh=0.20, device=None):
    """Drop-in for scripts/helpers/evaluate.py:59-238: writes
    <SAVE_PATH>/confusion_matrices/<filename>_stats.txt exactly like the reference.  Returns the raw
    matrix (the reference returns None); the seaborn heat map is not drawn."""
    save_dir = os.path.join(SAVE_PATH, "confusion_matrices")
    os.makedirs(save_dir, exist_ok=True)
    cm = confusion_matrix_counts(coco_anns, coco_dets, len(class_names), iou_thresh, score_thresh, device=device)
    st = confusion_stats(cm)
    with open(os.path.join(save_dir, filename.replace(".png", "_stats.txt")), "w") as f:
        f.write(f"Total FP: {st['total_fp']}\n")
        f.write(f"Total FN: {st['total_fn']}\n\n")
        f.write("Class\tTP\tFP\tFN\tPrecision\tRecall\n")
        for i, cls in enumerate(class_names):
            f.write(f"{cls}\t{int(st['tp'][i])}\t{int(st['fp'][i])}\t{int(st['fn'][i])}\t"
                    f"{st['precision'][i]:.3f}\t{st['recall'][i]:.3f}\n")
    return cm
