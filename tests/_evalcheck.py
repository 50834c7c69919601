"""Shared comparison for the P/R/F1 summary dicts (CPU oracle tests and GPU parity tests)."""
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
