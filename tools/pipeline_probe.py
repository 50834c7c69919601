#!/usr/bin/env python3
"""Probe: does the benchmark step leave the GPU idle between steps?  The executor forks the batch into two chunk
streams and JOINS them on the caller's stream, so step i+1 cannot start before the slower chunk of step i has
finished.  This script times the same work as bench.py (edge_n 640x640 B=64, yl_predict, hipGraph) as

  serial      one context, steps back to back on one stream              (bench.py's loop)
  lanes=K     K contexts of the same model, step i on context i % K, each on its own torch stream -- K batches in
              flight, every step still a complete yl_predict of one B=64 batch into its own output slot

for several (streams per context, K) pairs, and checks that every lane produces the serial result bit for bit.
    python tools/pipeline_probe.py [--model edge_n --batch 64 --steps 40]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="edge_n")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--combos", default="2x1,2x2,2x3,1x2,1x3,1x4,3x2,4x2")
    args = ap.parse_args()
    from yololite_amd import _lib
    from yololite_amd.model import HipContext
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.build_workload(args.model, 640, args.batch, seed=1, seg=bool(args.seg), dev=dev)
    prog, B = wl["prog"], args.batch
    xs = [wl["x"], bench.synth_images(B, 640, seed=77).to(dev), bench.synth_images(B, 640, seed=78).to(dev),
          bench.synth_images(B, 640, seed=79).to(dev)]
    MO = bench.MAX_OUT

    def mk_ctx(streams):
        c = HipContext(prog.img_size, prog.num_classes, prog.level_size, prog.level_anchors, prog, 0)
        c.set_option("graph", 1)
        c.set_option("streams", streams)
        return c

    def run(ctxs, lanes, steps):
        K = len(ctxs)
        outs = [(torch.empty((B, MO, 6), device=dev), torch.empty((B,), device=dev, dtype=torch.int32)) for _ in range(K)]
        for i in range(steps):
            k = i % K
            with torch.cuda.stream(lanes[k]):
                ctxs[k].predict(xs[k], _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=MO, out=outs[k])
        return outs

    ref = {}
    for combo in args.combos.split(","):
        s, K = (int(v) for v in combo.split("x"))
        ctxs = [mk_ctx(s) for _ in range(K)]
        lanes = [torch.cuda.Stream(device=dev) for _ in range(K)]
        outs = run(ctxs, lanes, 3 * K)
        torch.cuda.synchronize()
        for k in range(K):              # lane k processes input k: compare with the first configuration's result
            key = k
            d, c = outs[k][0].cpu().numpy(), outs[k][1].cpu().numpy()
            if key not in ref:
                ref[key] = (d, c)
            else:
                assert np.array_equal(c, ref[key][1]), (combo, k)
                for b in range(B):
                    assert np.array_equal(d[b, :c[b]], ref[key][0][b, :c[b]]), (combo, k, b)
        rates = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(ctxs, lanes, args.steps)
            torch.cuda.synchronize()
            rates.append(B * args.steps / (time.perf_counter() - t0))
        rates = np.sort(rates)
        print(f"streams/ctx {s}  lanes {K}:  {rates[len(rates) // 2]:9.1f} images/s  (min {rates[0]:.1f} max {rates[-1]:.1f})  "
              f"{1e3 * B / rates[len(rates) // 2]:.4f} ms/step", flush=True)
        del ctxs


if __name__ == "__main__":
    main()
