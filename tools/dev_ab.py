#!/usr/bin/env python3
"""Developer A/B of one "dev_select" bit: detections must be the same bits with the bit set and clear; step time of both
(graph replay, one stream, bench workload) and the eager per-layer table of the layers that changed.
    python tools/dev_ab.py BIT [model] [B] [seg]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from yololite_amd import _lib

bit = 1 << int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "edge_n"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
seg = len(sys.argv) > 4 and sys.argv[4] == "seg"
for S, b in ((640, B), (320, 3)):
    wl = bench.build_workload(name, S, b, seed=1, seg=seg)
    ctx, x, prog = wl["ctx"], wl["x"], wl["prog"]
    ctx.set_option("streams", 1)
    res, lay = {}, {}
    for dv in (bit, 0):
        ctx.set_option("dev_select", dv)
        d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
        res[dv] = (d.cpu().numpy().copy(), c.cpu().numpy().copy())
        lv = wl["model"](x)
        res[(dv, "lv")] = [t.cpu().numpy().copy() for t in (lv[0] if seg else lv)]
        if S == 640:
            lay[dv] = np.median(np.stack([np.asarray(ctx.forward(x, timed=True)[1]) for _ in range(9)]), axis=0)
    d0, c0 = res[bit]; d1, c1 = res[0]
    same = np.array_equal(c0, c1) and all(np.array_equal(d0[i, :c0[i]].view(np.uint32), d1[i, :c1[i]].view(np.uint32)) for i in range(b))
    same_lv = all(np.array_equal(u.view(np.uint32), v.view(np.uint32)) for u, v in zip(res[(bit, "lv")], res[(0, "lv")]))
    print(name, S, b, "dets", int(c0.sum()), "SAME" if same else "DIFFERENT", "levels", "SAME" if same_lv else "DIFFERENT", flush=True)
    if S == 640:
        for i, l in enumerate(prog.layers):
            if abs(lay[bit][i] - lay[0][i]) > 0.02 * lay[0][i]:
                print(f"  layer {i:3d} {l.name:40s} old {lay[bit][i]:.4f} new {lay[0][i]:.4f} ms")
        print("  eager sum old %.4f new %.4f" % (lay[bit].sum(), lay[0].sum()))
        ctx.set_option("graph", 1)
        for dv in (bit, 0, bit, 0):
            ctx.set_option("dev_select", dv)
            for _ in range(10): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
            print("  dev_select", dv, "ms/step %.4f  img/s %.0f" % (dt * 1e3, b / dt), flush=True)
        ctx.set_option("graph", 0)
