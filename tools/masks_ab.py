#!/usr/bin/env python3
"""Developer A/B of yl_masks_image: prints SHA-256 digests of the mask tensors of a fixed set of cases (config 4
at 640 x 640, back-mapped outputs of odd sizes incl. up-scaling, rows that are not multiples of 16 bytes, small
batches) and the kernel's event-timed duration.  Run under two builds (YOLOLITE_HIP_LIB=...) and diff the lines.

    python tools/masks_ab.py [--time 20]
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from yololite_amd import _lib  # noqa: E402


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = "cuda:0"
    B = args.batch
    wl = bench.build_workload("edge_m", 640, B, seed=1, seg=True, dev=dev)
    ctx, x = wl["ctx"], wl["x"]
    ctx.set_option("graph", 0)
    ctx.set_option("streams", 1)
    mo = bench.MAX_OUT
    d, c, i = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=mo, want_idx=True)
    print("lib", _lib.LIB_PATH, "dets/img", float(c.float().mean()))
    dd = d.cpu().numpy(); cc = c.cpu().numpy()
    bw = np.concatenate([dd[b, :cc[b], 2] - dd[b, :cc[b], 0] for b in range(B)])
    bh = np.concatenate([dd[b, :cc[b], 3] - dd[b, :cc[b], 1] for b in range(B)])
    print("counts min/max", cc.min(), cc.max(), "box w pct 50/90/99/max", np.percentile(bw, [50, 90, 99, 100]).round(1),
          "h", np.percentile(bh, [50, 90, 99, 100]).round(1), "mean area", float((bw * bh).mean()))
    print("640 packed ", digest(ctx.masks_image(d, c, i, packed=True)))
    print("640 uint8  ", digest(ctx.masks_image(d, c, i)))
    # back-mapped outputs: original sizes (h0, w0) of all kinds; dets are scaled like predict() with backmap does
    rng = np.random.RandomState(5)
    sizes = [(480, 640), (1080, 1920), (100, 150), (37, 45), (641, 333), (64, 2000), (1200, 50), (720, 1280)]
    for nb in (1, 3, 8):
        hw = [sizes[(k + nb) % len(sizes)] for k in range(nb)]
        bm = np.zeros((nb, 5), np.float32)
        for k, (h0, w0) in enumerate(hw):
            r = min(640 / h0, 640 / w0)
            nh, nw = int(round(h0 * r)), int(round(w0 * r))
            bm[k] = [(640 - nw) // 2, (640 - nh) // 2, r, w0, h0]
        bmt = torch.from_numpy(bm).to(dev)
        d3, c3, i3 = ctx.predict(x[:nb], _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=mo, backmap=bmt, want_idx=True)
        for packed in (True, False):
            print(f"backmap B={nb} packed={int(packed)}", digest(ctx.masks_image(d3, c3, i3, backmap=bmt, thr=0.3, packed=packed)))
    # timing of the B = 32 packed launch (events around the whole masks_image call minus its host work is not possible
    # from here: use rocprofv3 --kernel-trace --stats on this script for the kernel's own duration)
    d, c, i = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=mo, want_idx=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for packed in (True, False):
        row = 80 if packed else 640
        arena = torch.empty((B * mo * 640 * row,), device=dev, dtype=torch.uint8)
        ts = []
        for _ in range(5 if args.time else 0):       # back-to-back launches between two events: host time is hidden
            v = ctx.masks_image(d, c, i, packed=packed, arena=arena)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.time):
                v = ctx.masks_image(d, c, i, packed=packed, arena=arena)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.time)
        if ts:
            nb = float(c.sum()) * 640 * row
            print("masks_image(packed=%d, arena) launch ms: median %.3f min %.3f  -> %.2f TB/s written" % (
                packed, float(np.median(ts)), float(np.min(ts)), nb / (float(np.median(ts)) * 1e-3) / 1e12))
            ref = ctx.masks_image(d, c, i, packed=packed)
            for b in (0, B - 1):
                assert torch.equal(v[b, :int(c[b])], ref[b]), b
        del arena


if __name__ == "__main__":
    main()
