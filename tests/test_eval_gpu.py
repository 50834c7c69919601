"""Evaluate-path consumers on the device, through the C ABI, against (1) the golden
vectors produced by the reference's own functions and (2) the CPU oracle on larger seeded inputs.
Integer / index work and ratios of integer counts: the bar is bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import yololite_amd as ya                      # noqa: F401  (import shim)
from yololite_amd import evalops
from oracle import evalcons as oeval
from _evalcheck import assert_curves_equal

CASES = ["mixed", "ties", "crowded", "no_dets"]


def _fix(golden_dir):
    with open(os.path.join(golden_dir, "eval_consumers.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", CASES)
def test_curves_match_reference_fixture(golden_dir, case):
    fx = _fix(golden_dir)[case]
    for tag, want in fx["curves"].items():
        iou, steps = tag.split("_")
        got = evalops.build_curves_from_coco(fx["images"], fx["anns"], fx["dets"], None, iou=float(iou),
                                             steps=int(steps))
        assert_curves_equal(got, want)


@pytest.mark.parametrize("case", CASES)
def test_confusion_matches_reference_fixture(golden_dir, case, tmp_path):
    fx = _fix(golden_dir)[case]
    names = [f"c{i}" for i in range(fx["num_classes"])]
    for rec in fx["confusion"].values():
        cm = evalops.create_confusion_matrix(fx["anns"], fx["dets"], names, SAVE_PATH=str(tmp_path),
                                             iou_thresh=rec["iou_thresh"], score_thresh=rec["score_thresh"])
        assert np.array_equal(cm, np.asarray(rec["cm"]))
        txt = open(os.path.join(tmp_path, "confusion_matrices", "confusion_matrix_stats.txt")).read()
        assert txt == rec["stats_txt"]


def _random_coco(seed, n_img, n_cls, gt_per_img, det_per_img):
    r = np.random.RandomState(seed)
    anns, dets = [], []
    for img in range(1, n_img + 1):
        g = r.uniform(0, 500, (gt_per_img, 2))
        wh = r.uniform(10, 140, (gt_per_img, 2))
        gc = r.randint(1, n_cls + 1, gt_per_img)
        for k in range(gt_per_img):
            anns.append({"image_id": img, "category_id": int(gc[k]),
                         "bbox": [float(g[k, 0]), float(g[k, 1]), float(wh[k, 0]), float(wh[k, 1])]})
        for k in range(det_per_img):
            j = r.randint(gt_per_img)
            near = r.rand() < 0.6
            b = (np.r_[g[j], wh[j]] * (1 + r.normal(0, 0.08, 4))) if near else \
                np.r_[r.uniform(0, 500, 2), r.uniform(10, 140, 2)]
            dets.append({"image_id": img, "category_id": int(gc[j] if near else r.randint(1, n_cls + 1)),
                         "bbox": [float(np.float32(v)) for v in b],
                         "score": float(np.float32(round(r.rand(), 2) if k % 3 == 0 else r.rand()))})
    return anns, dets


def test_curves_and_confusion_match_oracle_large():
    anns, dets = _random_coco(7, n_img=40, n_cls=6, gt_per_img=25, det_per_img=60)
    want = oeval.build_curves_from_coco([], anns, dets, None, iou=0.5, steps=51)
    got = evalops.build_curves_from_coco([], anns, dets, None, iou=0.5, steps=51)
    assert_curves_equal(got, {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in want.items()})
    assert np.array_equal(got["recalls_rank"], want["recalls_rank"])
    assert np.array_equal(got["precisions_rank"], want["precisions_rank"])
    for iou_t, sc_t in ((0.5, 0.2), (0.45, 0.0)):
        cm_o = oeval.confusion_matrix_counts(anns, dets, 6, iou_t, sc_t)
        cm_g = evalops.confusion_matrix_counts(anns, dets, 6, iou_t, sc_t)
        assert np.array_equal(cm_o, cm_g)
        assert cm_g[:-1].sum() == len(anns)            # every ground truth lands in exactly one cell


def test_match_properties_full_scale():
    """Size-independent properties at evaluation scale (5000 images x 100 detections): a ground truth
    is matched at most once, TP count == matched-GT count, detections identical to a ground truth of
    their own key are all true positives when listed first."""
    r = np.random.RandomState(11)
    n_img, gpi = 5000, 8
    g_img = np.repeat(np.arange(n_img), gpi)
    g_cat = r.randint(1, 81, n_img * gpi)
    g_box = np.c_[r.uniform(0, 500, (n_img * gpi, 2)), r.uniform(10, 120, (n_img * gpi, 2))]
    # one exact copy of every ground truth (score 1.0) + 92 random detections per image
    d_img = np.r_[g_img, np.repeat(np.arange(n_img), 92)]
    d_cat = np.r_[g_cat, r.randint(1, 81, n_img * 92)]
    d_box = np.r_[g_box, np.c_[r.uniform(0, 500, (n_img * 92, 2)), r.uniform(10, 120, (n_img * 92, 2))]]
    d_sc = np.r_[np.ones(len(g_img)), r.uniform(0, 0.99, n_img * 92)]
    tp, has_gt, gm = evalops.match_per_class(d_img, d_cat, d_box, d_sc, g_img, g_cat, g_box, iou=0.5)
    assert tp[:len(g_img)].all()                        # exact copies, ranked first, each takes a ground truth
    assert gm.all() and int(tp.sum()) == int(gm.sum())  # ... so every ground truth is taken exactly once
    assert not tp[len(g_img):].any()                    # nothing left for the random detections
    tp_ge, fp_ge = evalops.sweep_counts(d_sc, tp, has_gt, np.linspace(0, 1, 201))
    assert tp_ge[0] == tp[has_gt].sum() and fp_ge[0] == (has_gt & (tp == 0)).sum()
    assert (np.diff(tp_ge) <= 0).all() and (np.diff(fp_ge) <= 0).all()
    assert tp_ge[-1] == len(g_img) and fp_ge[-1] == 0   # only the score-1.0 copies survive thr = 1.0
