"""Kalman-SORT tracking on the device behind the reference's class surface.

    KalmanSortTracker(iou_threshold=0.3, max_age=15, min_hits=2, match_by_class=True)
        .update(boxes, scores, classes) -> [{"track_id", "bbox", "cls", "score"}, ...]     one stream
        .reset()
    <- tools/tracker.py:157-326

    TrackerBank(num_streams, ...).update(dets, counts) -> (ids, boxes, cls, scores, counts) device tensors
        S independent streams advanced by ONE launch from the packed detections yl_predict leaves on the
        device ([S, max_out, 6] + counts) -- the multi-camera serving case, no host round trip.

No CPU fallback: the kernels live in libyololite_hip.so (csrc/yl_track.hip).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


class TrackerBank:
    def __init__(self, num_streams: int, max_tracks: int = 512, iou_threshold: float = 0.3, max_age: int = 15,
                 min_hits: int = 2, match_by_class: bool = True, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.YoloLiteHipError("TrackerBank needs a HIP device (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.S, self.T = int(num_streams), int(max_tracks)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else 0
        _lib.check(self.lib.yl_track_create(idx, self.S, self.T, float(iou_threshold), int(max_age), int(min_hits),
                                            int(bool(match_by_class)), C.byref(h)), what="yl_track_create")
        self._h = h
        self._alloc_outputs()

    def _alloc_outputs(self):
        dev = self.device
        self.out_id = torch.zeros((self.S, self.T), dtype=torch.int32, device=dev)
        self.out_cls = torch.zeros((self.S, self.T), dtype=torch.int32, device=dev)
        self.out_box = torch.zeros((self.S, self.T, 4), dtype=torch.float32, device=dev)
        self.out_score = torch.zeros((self.S, self.T), dtype=torch.float32, device=dev)
        self.out_count = torch.zeros((self.S,), dtype=torch.int32, device=dev)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.yl_track_destroy(h)
            self._h = None

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, stream_index: int = -1):
        _lib.check(self.lib.yl_track_reset(self._h, int(stream_index), self._stream()), what="yl_track_reset")

    def update(self, dets: torch.Tensor, counts: torch.Tensor):
        """dets [S, max_out, 6] fp32 (x1,y1,x2,y2,score,class), counts [S] int32, both on the device.
        Returns views of the bank's output tensors (overwritten by the next call)."""
        assert dets.is_cuda and counts.is_cuda and dets.dtype == torch.float32 and counts.dtype == torch.int32
        assert dets.dim() == 3 and dets.shape[0] == self.S and dets.shape[2] == 6 and dets.is_contiguous()
        _lib.check(self.lib.yl_track_update(self._h, dets.data_ptr() if dets.numel() else None, counts.data_ptr(),
                                            int(dets.shape[1]), self.out_id.data_ptr(), self.out_box.data_ptr(),
                                            self.out_cls.data_ptr(), self.out_score.data_ptr(),
                                            self.out_count.data_ptr(), self._stream()), what="yl_track_update")
        return self.out_id, self.out_box, self.out_cls, self.out_score, self.out_count

    def grow(self, max_tracks: int):
        """larger per-stream capacity, state kept (yl_track_grow; synchronises the device)."""
        if int(max_tracks) > self.T:
            _lib.check(self.lib.yl_track_grow(self._h, int(max_tracks)), what="yl_track_grow")
            self.T = int(max_tracks)
            self._alloc_outputs()

    def stats(self):
        """(tracks alive, tracks LOST to the capacity limit) per stream -- the bank drops new tracks beyond
        `max_tracks` (the reference's list is unbounded); a non-zero second array means results have diverged."""
        n = (C.c_int32 * self.S)()
        o = (C.c_int32 * self.S)()
        _lib.check(self.lib.yl_track_stats(self._h, n, o), what="yl_track_stats")
        return np.array(n[:]), np.array(o[:])


class KalmanSortTracker:
    """Single-stream drop-in for tools/tracker.py:157-326 (same constructor, update(), reset()).  The reference's
    track list is unbounded; the device bank has a capacity, so update() grows it (x2, up to 4096 tracks) whenever
    tracks alive + detections of the frame could exceed it, and raises if the device still reports lost tracks."""

    def __init__(self, iou_threshold: float = 0.3, max_age: int = 15, min_hits: int = 2, match_by_class: bool = True,
                 device="cuda:0", max_tracks: int = 512):
        self.iou_threshold, self.max_age, self.min_hits = iou_threshold, max_age, min_hits
        self.match_by_class = match_by_class
        self._bank = TrackerBank(1, max_tracks, iou_threshold, max_age, min_hits, match_by_class, device=device)

    def reset(self):
        self._bank.reset(-1)
        self._alive = 0

    def stats(self):
        return self._bank.stats()

    def update(self, boxes, scores, classes):
        n = 0 if boxes is None else len(boxes)
        b = np.zeros((n, 4), np.float32) if n == 0 else np.asarray(boxes, np.float32).reshape(-1, 4)
        s = np.zeros((n,), np.float32) if scores is None or len(scores) == 0 else np.asarray(scores, np.float32)
        c = np.zeros((n,), np.int32) if classes is None or len(classes) == 0 else np.asarray(classes, np.int32)
        d = np.zeros((1, max(n, 1), 6), np.float32)
        d[0, :n, :4], d[0, :n, 4], d[0, :n, 5] = b, s, c
        dev = self._bank.device
        self._alive = getattr(self, "_alive", 0)
        if self._alive + n > self._bank.T:                  # every detection may open a track (tracker.py:299-305)
            need = self._alive + n
            if need > 4096:
                raise _lib.YoloLiteHipError(f"KalmanSortTracker: {need} tracks exceed the device bank limit (4096)")
            self._bank.grow(min(4096, max(2 * self._bank.T, need)))
        ids, box, cls, sc, cnt = self._bank.update(torch.from_numpy(d).to(dev),
                                                   torch.tensor([n], dtype=torch.int32, device=dev))
        k = int(cnt[0].item())
        alive, lost = self._bank.stats()
        self._alive = int(alive[0])
        if int(lost[0]):
            raise _lib.YoloLiteHipError(f"KalmanSortTracker: {int(lost[0])} tracks lost to the bank capacity")
        ids, box, cls, sc = ids[0, :k].cpu().numpy(), box[0, :k].cpu().numpy(), cls[0, :k].cpu().numpy(), \
            sc[0, :k].cpu().numpy()
        return [{"track_id": int(ids[i]), "bbox": box[i].astype(np.float32), "cls": int(cls[i]),
                 "score": float(sc[i])} for i in range(k)]
