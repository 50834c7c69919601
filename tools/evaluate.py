#!/usr/bin/env python3
"""CLI with the reference's tools/evaluate.py surface (/root/reference/tools/evaluate.py:37-79) for
the part of evaluate_model that is on the hot path (/root/reference/scripts/helpers/evaluate.py:421-429,
253-303): batched forward -> _decode_batch_to_coco_dets(conf 0.001, iou 0.65) -> detections JSON, plus
the forward-only latency bench (2 warm-up + 10 timed batches, ms/img = sum ms / sum images).
COCOeval / curves / confusion matrix / summary image are unchanged CPU consumers of the detections
list and are out of scope.

    python tools/evaluate.py --weights W.pt --test_folder D [--img_size S] [--batch_size 8] [--device 0]
D holds images/ (or the images directly); labels are not needed for this part."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # also the package default (yololite_amd._lib); here before torch is imported

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--test_folder", required=True)
    ap.add_argument("--img_size", type=int, default=0)
    ap.add_argument("--device", default="0")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--no_letterbox", action="store_true")
    ap.add_argument("--out", default="runs/evaluate")
    ap.add_argument("--debug-levels", action="store_true", help="model(x) then _decode_batch_to_coco_dets (raw level "
                    "tensors materialised, as the reference does) instead of ONE fused yl_predict call; same detections")
    args = ap.parse_args()

    import yololite_amd as ya
    from tools.infer import imread_bgr, next_run_dir
    device = torch.device(f"cuda:{int(args.device)}")
    model, names, meta_img_size = ya.load_model_names_imgsize_from_ckpt(args.weights, device)
    S = int(args.img_size) if int(args.img_size) > 0 else int(meta_img_size)
    root = Path(args.test_folder)
    img_dir = root / "images" if (root / "images").exists() else root
    paths = sorted(str(p) for p in img_dir.glob("*") if p.suffix.lower() in (".jpg", ".jpeg", ".png", ".bmp"))
    if not paths:
        raise ValueError(f"no images under {img_dir}")
    run_dir = next_run_dir(args.out)
    ctx = model._ctx_for(S)
    lab_dir = root / "labels"
    coco_dets, coco_anns, fwd_ms, fwd_imgs = [], [], [], 0
    for i in range(0, len(paths), args.batch_size):
        chunk = paths[i:i + args.batch_size]
        imgs = [imread_bgr(p) for p in chunk]
        # the evaluate path's pre-processing (scripts/data/augment.py:153-171 through YoloDataset), on the device:
        # LongestMaxSize + PadIfNeeded (= the letterbox geometry) or Resize with --no_letterbox, A.Normalize arithmetic
        x, bms = ya.preprocess_batch(ctx, imgs, letterbox=not args.no_letterbox, norm="albumentations")
        for j, p in enumerate(chunk):
            padx, pady, scale, w0, h0 = bms[j]
            lab = lab_dir / (Path(p).stem + ".txt")
            if lab.exists():
                # YOLO rows "cls xc yc w h" (normalised, scripts/data/dataset.py:94-112) -> letterbox pixels ->
                # the reference's "[cx,cy,w,h]" rows of _xyxy_to_xywh (helpers.py:58-83), category_id = cls+1
                rows = np.loadtxt(str(lab), ndmin=2, dtype=np.float64)
                sx = (S / w0) if args.no_letterbox else scale
                sy = (S / h0) if args.no_letterbox else scale
                for r in rows.reshape(-1, 5) if rows.size else []:
                    x1 = (r[1] - r[3] / 2) * w0 * sx + padx; x2 = (r[1] + r[3] / 2) * w0 * sx + padx
                    y1 = (r[2] - r[4] / 2) * h0 * sy + pady; y2 = (r[2] + r[4] / 2) * h0 * sy + pady
                    bb = [float(np.float32(v)) for v in ((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1)]
                    coco_anns.append({"id": len(coco_anns) + 1, "image_id": i + j, "category_id": int(r[0]) + 1,
                                      "bbox": bb, "area": float(max(0.0, bb[2] * bb[3])), "iscrowd": 0})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.debug_levels:       # the reference's two calls: raw level tensors, then _decode_batch_to_coco_dets
            preds = model(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dets = ya._decode_batch_to_coco_dets(preds, S, conf_th=0.001, iou_th=0.65, add_one=True)
        else:                       # ONE fused yl_predict call (forward + decode + NMS), the path bench.py measures
            dets = ya.predict_coco_dets(ctx, x, conf_th=0.001, iou_th=0.65, add_one=True)
            t1 = time.perf_counter()
        if i // args.batch_size >= 2 and len(fwd_ms) < 10:          # evaluate.py:253-303 protocol
            fwd_ms.append((t1 - t0) * 1e3); fwd_imgs += len(chunk)
        for j, dl in enumerate(dets):
            for d in dl:
                coco_dets.append(dict(d, image_id=i + j, file_name=os.path.basename(chunk[j])))
    with open(Path(run_dir) / "detections.json", "w") as f:
        json.dump(coco_dets, f)
    summary = {"images": len(paths), "detections": len(coco_dets), "img_size": S,
               ("gpu_forward_ms_per_img" if args.debug_levels else "gpu_predict_ms_per_img"): (sum(fwd_ms) / fwd_imgs) if fwd_imgs else None}
    if coco_anns:
        # evaluate.py:480-489: P/R/F1 curves, then the confusion matrix at the best-F1 confidence -- on the device
        from yololite_amd import evalops
        cur = evalops.build_curves_from_coco([], coco_anns, coco_dets, Path(run_dir) / "curves", iou=0.50, steps=201)
        evalops.create_confusion_matrix(coco_anns, coco_dets, list(names), SAVE_PATH=str(run_dir),
                                        score_thresh=cur["best_conf"], device=device)
        summary.update({k: cur[k] for k in ("best_f1", "best_conf", "precision_at_best", "recall_at_best",
                                            "precision_at_fixed_conf", "recall_at_fixed_conf", "f1_at_fixed_conf")
                        if k in cur})
        with open(Path(run_dir) / "curves.json", "w") as f:
            json.dump({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in cur.items()}, f)
    with open(Path(run_dir) / "summary.json", "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary))
    print(f"saved to {run_dir}")


if __name__ == "__main__":
    main()
