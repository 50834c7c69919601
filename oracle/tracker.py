"""TEST INFRASTRUCTURE -- CPU restatement of the reference's Kalman-SORT tracker
(/root/reference/tools/tracker.py:9-326).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.

numpy float32 throughout, the same `@` / `np.linalg.inv` calls (so the same BLAS / LAPACK rounding as
the reference in the same environment).  State is kept as arrays per track instead of KalmanFilter
objects.  PINNED: tests/golden/tracker.json holds frame-by-frame outputs of the reference's own
KalmanSortTracker (tests/golden/make_tracker_fixtures.py; the file is pure numpy and imports as is).
"""
from __future__ import annotations

import numpy as np

_F = np.eye(7, dtype=np.float32)
_F[0, 4] = _F[1, 5] = _F[2, 6] = 1.0                          # tracker.py:93-97
_Q = np.eye(7, dtype=np.float32) * 0.01                       # :100
_H = np.zeros((4, 7), dtype=np.float32)
_H[0, 0] = _H[1, 1] = _H[2, 2] = _H[3, 3] = 1.0               # :103-107
_R = np.eye(4, dtype=np.float32)                              # :110
_I = np.eye(7, dtype=np.float32)


def xyxy_to_z(b):
    """tracker.py:9-24."""
    x1, y1, x2, y2 = b
    w = x2 - x1
    h = y2 - y1
    return np.array([x1 + w / 2.0, y1 + h / 2.0, w * h, w / (h + 1e-6)], dtype=np.float32)


def z_to_xyxy(x):
    """tracker.py:27-39."""
    cx, cy, s, r = x[0], x[1], x[2], x[3]
    w = np.sqrt(s * r)
    h = s / (w + 1e-6)
    return np.array([cx - w / 2.0, cy - h / 2.0, cx + w / 2.0, cy + h / 2.0], dtype=np.float32)


def iou_xyxy(A, B):
    """tracker.py:42-71."""
    if A.size == 0 or B.size == 0:
        return np.zeros((A.shape[0], B.shape[0]), dtype=np.float32)
    a, b = A[:, None, :], B[None, :, :]
    iw = np.maximum(0.0, np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]))
    ih = np.maximum(0.0, np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]))
    inter = iw * ih
    union = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0.0, inter / union, 0.0).astype(np.float32)


class SortOracle:
    """KalmanSortTracker (tracker.py:157-326) with the same constructor and update()."""

    def __init__(self, iou_threshold=0.3, max_age=15, min_hits=2, match_by_class=True):
        self.iou_threshold, self.max_age, self.min_hits = iou_threshold, max_age, min_hits
        self.match_by_class = match_by_class
        self.reset()

    def reset(self):
        self.tracks = []            # dicts: x [7,1], P [7,7], id, cls, score, hits, age, tsu
        self._next_id = 1

    def update(self, boxes, scores, classes):
        boxes = np.zeros((0, 4), np.float32) if boxes is None or len(boxes) == 0 else np.asarray(boxes, np.float32)
        scores = (np.zeros((boxes.shape[0],), np.float32) if scores is None or len(scores) == 0
                  else np.asarray(scores, np.float32))
        classes = (np.zeros((boxes.shape[0],), np.int32) if classes is None or len(classes) == 0
                   else np.asarray(classes, np.int32))
        for t in self.tracks:                                                   # :223-227
            t["x"] = _F @ t["x"]
            t["P"] = _F @ t["P"] @ _F.T + _Q
            t["age"] += 1
            t["tsu"] += 1
        if boxes.shape[0] == 0:                                                 # :230-232
            self.tracks = [t for t in self.tracks if t["tsu"] <= self.max_age]
            return []
        if self.tracks:
            tb = np.array([z_to_xyxy(t["x"][:4, 0]) for t in self.tracks], dtype=np.float32)
            iou = iou_xyxy(tb, boxes)
        else:
            iou = np.zeros((0, boxes.shape[0]), np.float32)
        mt, md, matches = set(), set(), []
        if self.tracks and boxes.shape[0] > 0:                                  # :251-283
            D = iou.shape[1]
            if self.match_by_class:
                same = (np.array([t["cls"] for t in self.tracks])[:, None] == classes[None, :]).astype(np.float32)
                iou = iou * same
            for idx in np.argsort(-iou.reshape(-1)):
                i, j = idx // D, idx % D
                if iou[i, j] < self.iou_threshold:
                    break
                if i in mt or j in md:
                    continue
                mt.add(i); md.add(j); matches.append((i, j))
        for ti, dj in matches:                                                  # :286-296
            t = self.tracks[ti]
            z = xyxy_to_z(boxes[dj]).reshape(4, 1).astype(np.float32)
            y = z - (_H @ t["x"])
            S = _H @ t["P"] @ _H.T + _R
            K = t["P"] @ _H.T @ np.linalg.inv(S)
            t["x"] = t["x"] + K @ y
            t["P"] = (_I - K @ _H) @ t["P"]
            t["score"] = max(t["score"], float(scores[dj]))
            if not self.match_by_class:
                t["cls"] = int(classes[dj])
            t["hits"] += 1
            t["tsu"] = 0
        for j in range(boxes.shape[0]):                                         # :299-301
            if j in md:
                continue
            x = np.zeros((7, 1), np.float32)
            x[:4, 0] = xyxy_to_z(boxes[j])
            self.tracks.append(dict(x=x, P=np.eye(7, dtype=np.float32) * 10.0, id=self._next_id, cls=int(classes[j]),
                                    score=float(scores[j]), hits=1, age=1, tsu=0))
            self._next_id += 1
        self.tracks = [t for t in self.tracks if t["tsu"] <= self.max_age]      # :304
        return [{"track_id": t["id"], "bbox": z_to_xyxy(t["x"][:4, 0]), "cls": t["cls"], "score": t["score"]}
                for t in self.tracks if t["tsu"] == 0 and t["hits"] >= self.min_hits]
