#!/usr/bin/env python3
"""CLI with the reference's tools/infer.py surface (/root/reference/tools/infer.py:396-559), running
the MI355X-native path: checkpoint -> letterbox/normalise (host) -> HIP forward + decode + per-class
NMS + back-map -> runs/infer/<n>/{labels/*.txt, json/*.json}.

    python tools/infer.py --weights W.pt --img I.png|--img_dir D [--img_size S] [--conf 0.4] [--iou 0.5]
                          [--max_det 300] [--save_txt] [--no_letterbox] [--device 0]

Differences, all documented in DESIGN.md: images are decoded with PIL (cv2 is absent in this environment)
and letterboxed/normalised on the GPU (yl_preprocess, OpenCV-style fixed-point bilinear),
the annotated *_pred.jpg is not drawn (cosmetics, out of scope), --device cpu is refused (no CPU path).
Like the reference's main path, --max_det is NOT forwarded to the per-class NMS (cap 300 per class)."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # also the package default (yololite_amd._lib); here before torch is imported

import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def next_run_dir(base: str) -> str:
    """runs/<kind>/<n> with the lowest free positive n, created atomically (same observable numbering as
    /root/reference/tools/infer.py:108-118; concurrent CLI runs cannot share a directory)."""
    os.makedirs(base, exist_ok=True)
    taken = {int(e) for e in os.listdir(base) if e.isdigit()}
    n = 0
    while True:
        n = next(k for k in range(n + 1, len(taken) + n + 3) if k not in taken)
        path = os.path.join(base, str(n))
        try:
            os.mkdir(path)
        except FileExistsError:                 # lost a race with another process: it is taken now
            taken.add(n)
            n -= 1
            continue
        return os.path.realpath(path)


def imread_bgr(path: str):
    from PIL import Image
    try:
        return np.asarray(Image.open(path).convert("RGB"))[..., ::-1].copy()
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--img", default=None)
    ap.add_argument("--img_dir", default=None)
    ap.add_argument("--img_size", type=int, default=0)
    ap.add_argument("--device", default="0")
    ap.add_argument("--conf", type=float, default=0.4)
    ap.add_argument("--iou", type=float, default=0.50)
    ap.add_argument("--max_det", type=int, default=300)
    ap.add_argument("--save_txt", action="store_true")
    ap.add_argument("--no_letterbox", action="store_true")
    ap.add_argument("--batch", type=int, default=16, help="images per HIP launch (the reference runs one at a time)")
    ap.add_argument("--debug-levels", action="store_true", help="materialise the raw level tensors: model(x) then the "
                    "post-processing call, as the reference does (default: ONE fused yl_predict call, same detections)")
    args = ap.parse_args()

    import yololite_amd as ya
    if args.device == "cpu":
        raise SystemExit("this build has no CPU execution path; use --device <gpu index>")
    device = torch.device(f"cuda:{int(args.device)}")
    model, names, meta_img_size = ya.load_model_names_imgsize_from_ckpt(args.weights, device)
    S = int(args.img_size) if int(args.img_size) > 0 else int(meta_img_size)

    if args.img and Path(args.img).exists():
        paths = [args.img]
    elif args.img_dir and Path(args.img_dir).exists():
        exts = (".jpg", ".jpeg", ".png", ".bmp")
        paths = sorted(str(p) for p in Path(args.img_dir).glob("*") if p.suffix.lower() in exts)
    else:
        raise ValueError("Ange --img eller --img_dir som existerar.")
    run_dir = next_run_dir("runs/infer")
    (Path(run_dir) / "labels").mkdir(parents=True, exist_ok=True)
    (Path(run_dir) / "json").mkdir(parents=True, exist_ok=True)

    ctx = model._ctx_for(S)
    for i in range(0, len(paths), args.batch):
        chunk, imgs, ok = paths[i:i + args.batch], [], []
        for pth in chunk:
            img0 = imread_bgr(pth)
            if img0 is None:
                print(f"Varnar: kunde inte läsa {pth}")
                continue
            imgs.append(img0); ok.append((pth, img0.shape[:2]))
        if not imgs:
            continue
        x, bms = ya.preprocess_batch(ctx, imgs, letterbox=not args.no_letterbox)     # letterbox + normalise on the GPU
        if args.debug_levels:
            outs = model(x)
            res = ya.infer_main_postprocess(outs, S, args.conf, args.iou, backmap=bms)
        else:       # the path bench.py measures: decode inside the head-output convs, no raw level tensors
            res = ya.predict_main(ctx, x, args.conf, args.iou, backmap=bms)
        for j, (pth, (h, w)) in enumerate(ok):
            b, s, c = res["boxes"][j], res["scores"][j], res["classes"][j]
            if args.save_txt and b.size > 0:
                cx, cy = (b[:, 0] + b[:, 2]) / 2.0 / w, (b[:, 1] + b[:, 3]) / 2.0 / h
                bw, bh = (b[:, 2] - b[:, 0]) / w, (b[:, 3] - b[:, 1]) / h
                with open(Path(run_dir) / "labels" / f"{Path(pth).stem}.txt", "w", encoding="utf-8") as f:
                    for k in range(len(s)):
                        f.write(f"{int(c[k])} {cx[k]:.6f} {cy[k]:.6f} {bw[k]:.6f} {bh[k]:.6f} {s[k]:.4f}\n")
            rec = [{"bbox_xyxy": [float(v) for v in bb], "score": float(ss), "class_id": int(cc),
                    "class_name": names[int(cc)] if int(cc) < len(names) else str(int(cc))}
                   for bb, ss, cc in zip(b.tolist(), s.tolist(), c.tolist())]
            with open(Path(run_dir) / "json" / f"{Path(pth).stem}.json", "w", encoding="utf-8") as f:
                json.dump({"image": pth, "detections": rec}, f, ensure_ascii=False, indent=2)
            print(f"✓ {pth}: {len(rec)} detections")
    print(f"Allt sparat i: {run_dir}")


if __name__ == "__main__":
    main()
