#!/usr/bin/env python3
"""Developer A/B of the window-in-LDS head launch (yl_conv_dpw_kernel) against the tap-load one (yl_conv_dpp_kernel,
"dev_select" bit 16): detections must be the same bits; step time of both (graph replay, 1 stream, bench workload).
    python tools/dpw_ab.py [model] [B]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from yololite_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "edge_n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
OFF = _lib.DEV_DPW_OFF
for S, b in ((640, B), (416, 3), (96, 5)):
    wl = bench.build_workload(name, S, b, seed=1)
    ctx, x = wl["ctx"], wl["x"]
    ctx.set_option("streams", 1)
    res = {}
    for dv in (OFF, 0):
        ctx.set_option("dev_select", dv)
        for mode, conf, iou in ((_lib.POST_MAIN, 0.4, 0.5), (_lib.POST_EVAL, 0.001, 0.65)):
            mo = 1024 if mode == _lib.POST_MAIN else ctx.N
            d, c = ctx.predict(x, mode, conf, iou, per_class_cap=300 if mode == _lib.POST_MAIN else 0, max_out=mo)
            res[(dv, mode)] = (d.cpu().numpy().copy(), c.cpu().numpy().copy())
    for mode in (_lib.POST_MAIN, _lib.POST_EVAL):
        d0, c0 = res[(OFF, mode)]; d1, c1 = res[(0, mode)]
        same = np.array_equal(c0, c1) and all(np.array_equal(d0[i, :c0[i]].view(np.uint32), d1[i, :c1[i]].view(np.uint32)) for i in range(b))
        print(name, S, b, "mode", mode, "dets", int(c0.sum()), "SAME" if same else "DIFFERENT", flush=True)
    if S == 640:
        ctx.set_option("graph", 1)
        for dv in (OFF, 0, OFF, 0):
            ctx.set_option("dev_select", dv)
            for _ in range(10): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
            print("dev_select", dv, "ms/step %.4f  img/s %.0f" % (dt * 1e3, b / dt), flush=True)
        ctx.set_option("graph", 0)
