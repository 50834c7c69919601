#!/usr/bin/env python3
"""Winograd margin of edge_m + seg measured on MASK BITS.  The prototype branch's two dense 3x3 convs run
as Winograd F(2x2,3x3) by default (option "winograd" 1) and feed sigmoid(coef . proto) > 0.5 -- a thresholded output, so the
score margin files (profiles/r04_winograd_margin*.json) say nothing about it.  For several weight seeds x all 32 images
(BASELINE config 4: edge_m + seg, 640x640, B = 32; up to 48 detections per image): image-resolution masks of the HIP path
under winograd 0 (direct convolution) and 1 against the oracle's masks_image_for on the ORACLE's own levels / prototypes for
the same kept candidates and boxes: pixels that differ (xor) over pixels set in either (union), and the IoU.
    python tools/wino_mask_margin.py [--seeds 1 2 3 4] > profiles/rNN_winograd_mask_margin_edge_m_seg.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import model as omodel, postproc as opost  # noqa: E402
from yololite_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--max-dets", type=int, default=48)
    a = ap.parse_args()
    rows = []
    for seed in a.seeds:
        wl = bench.build_workload("edge_m", 640, a.batch, seed=seed, seg=True, dev="cuda:0", rank=seed)
        ctx, x = wl["ctx"], wl["x"]
        ctx.set_option("graph", 0); ctx.set_option("streams", 1)
        orc = omodel.build_from_meta(wl["meta"]).eval()
        orc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in wl["sd"].items()}, strict=False)
        parts = []
        with torch.no_grad():
            for i in range(0, a.batch, 4):
                parts.append(orc(x[i:i + 4].cpu()))
        ref_lv = [torch.cat([p[0][l] for p in parts]) for l in range(len(parts[0][0]))]
        ref_pr = torch.cat([p[1] for p in parts])
        r = {"seed": seed}
        for w in (0, 1):
            ctx.set_option("winograd", w)
            d, c, idx = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=bench.MAX_OUT, want_idx=True)
            mk = ctx.masks_image(d, c, idx)
            xor = union = inter = ndet = 0
            worst = 1.0
            for b in range(a.batch):
                n = min(int(c[b]), a.max_dets)
                if n == 0:
                    continue
                keep = [idx[b, :n].cpu().numpy()]
                boxes = [d[b, :n, :4].cpu().numpy()]
                exp = opost.masks_image_for([t[b:b + 1] for t in ref_lv], ref_pr[b:b + 1], 80, 640, keep, boxes, [(640, 640)])[0].astype(bool)
                got = mk[b][:n].cpu().numpy().astype(bool)
                xo, un, it = int((got ^ exp).sum()), int((got | exp).sum()), int((got & exp).sum())
                xor += xo; union += un; inter += it; ndet += n
                if un:
                    worst = min(worst, it / un)
            r[f"winograd_{w}"] = {"detections": ndet, "mask_pixels_union": union, "pixels_that_differ": xor,
                                  "flip_fraction": xor / max(union, 1), "iou_all": inter / max(union, 1), "worst_image_iou": worst}
        ctx.set_option("winograd", 1)
        rows.append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
        del wl
        torch.cuda.empty_cache()
    out = {"model": "edge_m + seg", "batch": a.batch, "img": 640, "max_dets_per_image": a.max_dets, "seeds": rows,
           "worst_flip_fraction": {f"winograd_{w}": max(r[f"winograd_{w}"]["flip_fraction"] for r in rows) for w in (0, 1)},
           "worst_image_iou": {f"winograd_{w}": min(r[f"winograd_{w}"]["worst_image_iou"] for r in rows) for w in (0, 1)},
           "bar": "north_star: mask IoU >= 0.999 (flip fraction <= 1e-3)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
