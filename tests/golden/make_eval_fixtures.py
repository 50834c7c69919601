#!/usr/bin/env python3
"""Generate tests/golden/eval_consumers.json by RUNNING THE REFERENCE's own evaluation consumers
(/root/reference, read-only) in this container:

  scripts/data/p_r_f1.py          build_curves_from_coco      (imported as is: numpy + matplotlib)
  scripts/helpers/evaluate.py     create_confusion_matrix     (seaborn stubbed -- it only draws the
                                  heat map; torchvision stubbed for the unrelated helpers import;
                                  sklearn.metrics.confusion_matrix is the real one, wrapped only to
                                  RECORD the matrix the reference computes, which it otherwise just plots)

The fixture stores the synthetic COCO-style inputs (images / annotations / detections) and the
reference's outputs.  Run:  cd /root/repo && python tests/golden/make_eval_fixtures.py
"""
import json
import os
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

REF = "/root/reference"
OUT = os.environ.get("YL_FIXTURE_OUT") or os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    sns = types.ModuleType("seaborn")
    sns.heatmap = lambda *a, **k: None
    sys.modules["seaborn"] = sns
    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvo.nms = tvo.box_iou = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not used here"))
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo


def f32(v):
    """python float holding an fp32-representable value (what .tolist() of an fp32 tensor gives)."""
    return float(np.float32(v))


def make_case(seed, n_img, n_cls, max_gt, fp_rate, tie_scores, empty_gt_imgs=(), no_det_imgs=(), min_gt=1):
    """COCO-style lists.  Ground truths are arbitrary doubles; detections are fp32-valued [cx,cy,w,h]
    rows (sic: the reference's _xyxy_to_xywh emits centres and every consumer treats them as x,y)."""
    r = np.random.RandomState(seed)
    images, anns, dets = [], [], []
    ann_id = 1
    for img_id in range(1, n_img + 1):
        images.append({"id": img_id, "file_name": f"val_{img_id}.jpg", "width": 640, "height": 640})
        n_gt = 0 if img_id in empty_gt_imgs else int(r.randint(min_gt, max_gt + 1))
        gts = []
        for _ in range(n_gt):
            w, h = r.uniform(20, 200, 2)
            x, y = r.uniform(0, 640 - w), r.uniform(0, 640 - h)
            c = int(r.randint(1, n_cls + 1))
            box = [float(x), float(y), float(w), float(h)]
            if r.rand() < 0.05:
                box[2] = 0.0                                   # zero-area ground truth
            gts.append((box, c))
            anns.append({"id": ann_id, "image_id": img_id, "category_id": c, "bbox": box,
                         "area": float(max(0.0, box[2] * box[3])), "iscrowd": 0})
            ann_id += 1
        if gts and r.rand() < 0.3:                             # an exact duplicate ground truth (IoU ties)
            box, c = gts[int(r.randint(len(gts)))]
            anns.append({"id": ann_id, "image_id": img_id, "category_id": c, "bbox": list(box),
                         "area": float(max(0.0, box[2] * box[3])), "iscrowd": 0})
            ann_id += 1
        if img_id in no_det_imgs:
            continue
        for box, c in gts:                                     # 0..3 jittered detections per ground truth
            for _ in range(int(r.randint(0, 4))):
                j = r.normal(0, 0.12, 4) * [box[2] + 1, box[3] + 1, box[2] + 1, box[3] + 1]
                cls = c if r.rand() < 0.85 else int(r.randint(1, n_cls + 1))
                s = r.rand() ** 0.7
                if tie_scores:
                    s = round(s * 20) / 20.0                   # many equal scores, some exactly on a threshold
                dets.append({"image_id": img_id, "category_id": cls,
                             "bbox": [f32(box[0] + j[0]), f32(box[1] + j[1]), f32(max(box[2] + j[2], 0.0)),
                                      f32(max(box[3] + j[3], 0.0))], "score": f32(s)})
            if r.rand() < 0.15:                                # a detection identical to the ground truth
                dets.append({"image_id": img_id, "category_id": c, "bbox": [f32(v) for v in box],
                             "score": f32(r.rand())})
        for _ in range(int(r.poisson(fp_rate))):               # unrelated false positives
            w, h = r.uniform(5, 150, 2)
            dets.append({"image_id": img_id, "category_id": int(r.randint(1, n_cls + 1)),
                         "bbox": [f32(r.uniform(0, 600)), f32(r.uniform(0, 600)), f32(w), f32(h)],
                         "score": f32(r.rand() * 0.6)})
    order = r.permutation(len(dets))                           # list order matters for score ties
    return images, anns, [dets[i] for i in order]


def main():
    install_stubs()
    os.chdir(REF)                                              # evaluate.py appends os.getcwd() to sys.path
    sys.path.append(REF)
    from scripts.data.p_r_f1 import build_curves_from_coco
    import scripts.helpers.evaluate as ev

    recorded = {}
    real_cm = ev.confusion_matrix

    def recording_cm(y_true, y_pred, labels=None):
        m = real_cm(y_true, y_pred, labels=labels)
        recorded["cm"] = np.asarray(m).tolist()
        return m
    ev.confusion_matrix = recording_cm

    cases = {
        "mixed": dict(seed=1, n_img=24, n_cls=4, max_gt=9, fp_rate=3.0, tie_scores=False,
                      empty_gt_imgs=(5, 17), no_det_imgs=(9,)),
        "ties": dict(seed=2, n_img=16, n_cls=3, max_gt=6, fp_rate=2.0, tie_scores=True, empty_gt_imgs=(3,)),
        "crowded": dict(seed=3, n_img=2, n_cls=1, min_gt=70, max_gt=100, fp_rate=30.0, tie_scores=False),   # > 64 GT per key
        "no_dets": dict(seed=4, n_img=4, n_cls=2, max_gt=3, fp_rate=0.0, tie_scores=False,
                        no_det_imgs=(1, 2, 3, 4)),
    }
    out = {}
    for name, kw in cases.items():
        images, anns, dets = make_case(**kw)
        n_cls = kw["n_cls"]
        rec = {"num_classes": n_cls, "images": images, "anns": anns, "dets": dets, "curves": {}, "confusion": {}}
        for iou, steps in ((0.5, 201), (0.75, 41)):
            s = build_curves_from_coco(images, anns, dets, out_dir=None, iou=iou, steps=steps)
            rec["curves"][f"{iou}_{steps}"] = {k: (v.tolist() if isinstance(v, np.ndarray) else v)
                                               for k, v in s.items()}
        for iou_t, score_t in ((0.5, 0.20), (0.5, rec["curves"]["0.5_201"]["best_conf"]), (0.3, 0.0)):
            with tempfile.TemporaryDirectory() as td:
                recorded.clear()
                ev.create_confusion_matrix(anns, dets, [f"c{i}" for i in range(n_cls)], SAVE_PATH=td,
                                           iou_thresh=iou_t, score_thresh=score_t)
                stats = open(os.path.join(td, "confusion_matrices", "confusion_matrix_stats.txt")).read()
            rec["confusion"][f"{iou_t}_{score_t!r}"] = {"iou_thresh": iou_t, "score_thresh": score_t,
                                                        "cm": recorded["cm"], "stats_txt": stats}
        out[name] = rec
        print(name, "images", len(images), "anns", len(anns), "dets", len(dets),
              "best_f1", rec["curves"]["0.5_201"]["best_f1"])
    with open(os.path.join(OUT, "eval_consumers.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(OUT, "eval_consumers.json"), os.path.getsize(os.path.join(OUT, "eval_consumers.json")))


if __name__ == "__main__":
    main()
