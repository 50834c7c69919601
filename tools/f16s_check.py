#!/usr/bin/env python3
"""Developer check of the fp16-storage mode ("store_f16"): logits against the fp32 run of the same library and against
"mfma_f16" (fp16 operands, fp32 tensors), detections, arena size, step time.
    python tools/f16s_check.py [model] [B] [S] [seg]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from yololite_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "edge_n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 320
seg = len(sys.argv) > 4 and sys.argv[4] == "seg"
wl = bench.build_workload(name, S, B, seed=1, seg=seg)
ctx, x, model = wl["ctx"], wl["x"], wl["model"]
def levels():
    o = model(x)
    return [t.clone() for t in (o[0] if seg else o)]
ref = levels()
out = {}
for mode in ("mfma_f16", "store_f16"):
    ctx.set_option(mode, 1)
    lv = levels()
    worst = max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(lv, ref))
    d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
    out[mode] = (worst, c.cpu().numpy().copy())
    ctx.set_option("graph", 1)
    for _ in range(5): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    ctx.set_option("graph", 0)
    print(f"{name} B={B} S={S} {mode}: max logit error / level max {worst:.2e}, dets {int(c.sum())}, {dt*1e3:.3f} ms/step, {B/dt:.0f} img/s", flush=True)
    ctx.set_option(mode, 0)
d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
print("fp32 dets", int(c.sum()), "back to fp32 bits:", all(torch.equal(a, b) for a, b in zip(levels(), ref)))
