#!/usr/bin/env python3
"""Generate tests/golden/tracker.json by RUNNING THE REFERENCE's own KalmanSortTracker
(/root/reference/tools/tracker.py -- pure numpy, imported unmodified) on seeded synthetic sequences.
The fixture stores the per-frame detections and the reference's per-frame outputs.

Scenes are built so that assignment decisions have margins: objects move on smooth paths with modest
noise, IoUs of competing pairs are well separated and far from the threshold (the device kernel sums the
Kalman products in a different order than BLAS, so decisions that hinge on the last bit are not pinned).
Run:  python tests/golden/make_tracker_fixtures.py
"""
import importlib.util
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np

OUT = os.environ.get("YL_FIXTURE_OUT") or os.path.dirname(os.path.abspath(__file__))


def scene(seed, n_obj, n_frames, n_cls, miss=0.1, clutter=0.5, appear_late=True):
    """list over frames of (boxes [N,4] f32, scores [N] f32, classes [N] i32)."""
    r = np.random.RandomState(seed)
    pos = r.uniform(60, 540, (n_obj, 2)); vel = r.uniform(-6, 6, (n_obj, 2))
    wh = r.uniform(30, 90, (n_obj, 2)); cls = r.randint(0, n_cls, n_obj)
    t0 = r.randint(0, n_frames // 3, n_obj) if appear_late else np.zeros(n_obj, int)
    t1 = n_frames - r.randint(0, n_frames // 3, n_obj)
    frames = []
    for f in range(n_frames):
        bx, sc, cl = [], [], []
        for o in range(n_obj):
            if not (t0[o] <= f < t1[o]) or r.rand() < miss:
                continue
            c = pos[o] + vel[o] * f + r.normal(0, 0.8, 2)
            w = wh[o] * (1 + 0.002 * f) + r.normal(0, 0.5, 2)
            bx.append([c[0] - w[0] / 2, c[1] - w[1] / 2, c[0] + w[0] / 2, c[1] + w[1] / 2])
            sc.append(r.uniform(0.4, 0.99)); cl.append(cls[o])
        for _ in range(r.poisson(clutter)):                       # short-lived false positives
            c = r.uniform(0, 600, 2); w = r.uniform(10, 40, 2)
            bx.append([c[0], c[1], c[0] + w[0], c[1] + w[1]]); sc.append(r.uniform(0.4, 0.6)); cl.append(r.randint(0, n_cls))
        p = r.permutation(len(bx))
        frames.append((np.asarray(bx, np.float32).reshape(-1, 4)[p], np.asarray(sc, np.float32)[p],
                       np.asarray(cl, np.int32)[p]))
    return frames


def main():
    spec = importlib.util.spec_from_file_location("ref_tracker", "/root/reference/tools/tracker.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cases = {
        "default": (dict(), scene(1, n_obj=6, n_frames=40, n_cls=3)),
        "crowd": (dict(), scene(2, n_obj=25, n_frames=30, n_cls=4, miss=0.15, clutter=2.0)),
        "anyclass": (dict(match_by_class=False, min_hits=1, max_age=3, iou_threshold=0.2),
                     scene(3, n_obj=8, n_frames=30, n_cls=5, miss=0.25)),
        "gaps": (dict(max_age=2, min_hits=3), scene(4, n_obj=5, n_frames=36, n_cls=2, miss=0.35, clutter=0.2)),
    }
    out = {}
    for name, (kw, frames) in cases.items():
        trk = ref.KalmanSortTracker(**kw)
        rec = {"kw": kw, "frames": []}
        for i, (b, s, c) in enumerate(frames):
            if name == "gaps" and i in (10, 11, 12, 25):          # frames without any detection
                b, s, c = b[:0], s[:0], c[:0]
            o = trk.update(b, s, c)
            rec["frames"].append({"boxes": b.tolist(), "scores": s.tolist(), "classes": c.tolist(),
                                  "out": [{"track_id": int(t["track_id"]), "bbox": [float(v) for v in t["bbox"]],
                                           "cls": int(t["cls"]), "score": float(t["score"])} for t in o],
                                  "n_tracks": len(trk.tracks)})
        out[name] = rec
        print(name, "frames", len(frames), "final tracks", len(trk.tracks), "next id", trk._next_id,
              "outputs/frame", np.mean([len(f["out"]) for f in rec["frames"]]))
    with open(os.path.join(OUT, "tracker.json"), "w") as f:
        json.dump(out, f)
    print("wrote tracker.json", os.path.getsize(os.path.join(OUT, "tracker.json")))


if __name__ == "__main__":
    main()
