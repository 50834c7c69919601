#!/usr/bin/env python3
"""Developer A/B helper: per-layer eager times + SHA-256 of the raw levels for one model.
    python tools/layers_ab.py yololite_m 32 [seg]      (run under different env / YOLOLITE_HIP_LIB and diff)"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
name, B = sys.argv[1], int(sys.argv[2])
seg = len(sys.argv) > 3 and sys.argv[3] == "seg"
wl = bench.build_workload(name, 640, B, seed=1, seg=seg)
ctx, x, prog = wl["ctx"], wl["x"], wl["prog"]
ctx.set_option("graph", 0); ctx.set_option("streams", 1)
out = wl["model"](x)
lv = out[0] if seg else out
h = hashlib.sha256()
for t in lv:
    h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
print("levels", name, B, h.hexdigest()[:16])
lay = np.median(np.stack([np.asarray(ctx.forward(x, timed=True)[1]) for _ in range(9)]), axis=0)
for i, (l, ms) in enumerate(zip(prog.layers, lay)):
    print(f"{i:3d} {l.name:40s} cin{l.cin:4d} cout{l.cout:4d} k{l.k} dw{l.dw_k} {ms:8.4f} ms {2.0 * l.macs * B / (ms * 1e-3) / 1e12:7.2f} TF")
print("sum ms %.4f" % lay.sum())
