"""Shared comparisons (CPU oracle tests and GPU parity tests): P/R/F1 summary dicts, tracker sequences."""
import numpy as np

CURVE_KEYS = ("best_f1", "best_conf", "precision_at_best", "recall_at_best", "fixed_conf",
              "precision_at_fixed_conf", "recall_at_fixed_conf", "f1_at_fixed_conf", "best_idx")


def assert_curves_equal(got, want):
    """Bit-exact: the curves are ratios of integer counts evaluated in float64 on both sides."""
    assert got["iou"] == want["iou"]
    if "P_curve" not in want:                      # the reference's "no predictions" early return
        assert "P_curve" not in got and got["best_f1"] == 0.0 and got["best_conf"] == 0.0
        return
    for k in CURVE_KEYS:
        assert got[k] == want[k], k
    for k in ("P_curve", "R_curve", "F1_curve", "confs"):
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k


def check_tracker_sequence(make_tracker, rec, box_tol):
    """Replay a fixture sequence through `make_tracker(**kw)`; ids / classes / counts exact, boxes within
    `box_tol` pixels, scores exact (a running max of the inputs)."""
    trk = make_tracker(**rec["kw"])
    for fi, fr in enumerate(rec["frames"]):
        out = trk.update(np.asarray(fr["boxes"], np.float32).reshape(-1, 4), np.asarray(fr["scores"], np.float32),
                         np.asarray(fr["classes"], np.int32))
        want = fr["out"]
        assert [t["track_id"] for t in out] == [t["track_id"] for t in want], f"frame {fi}"
        assert [t["cls"] for t in out] == [t["cls"] for t in want], f"frame {fi}"
        for a, b in zip(out, want):
            assert np.abs(np.asarray(a["bbox"], np.float64) - np.asarray(b["bbox"])).max() <= box_tol, f"frame {fi}"
            assert a["score"] == b["score"]
