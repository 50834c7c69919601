#!/usr/bin/env python3
"""Developer A/B of the fused entry kernel (yl_stemblock_kernel): SHA-256 of the raw levels of edge_n / edge_m forwards
at a few shapes (border tiles, odd sizes) and the eager duration of layer 0 at B = 64, 640 x 640.  Run under two
builds (YOLOLITE_HIP_LIB=...) and diff."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


print("lib", _lib.LIB_PATH)
for name, S, B in (("edge_n", 640, 8), ("edge_n", 416, 3), ("edge_n", 352, 2), ("edge_m", 320, 2), ("edge_n", 96, 5), ("edge_n", 224, 4)):
    meta = zoo_meta(name, 80, S)
    m = ya.build_model_from_meta(meta)
    m.load_state_dict(synth_state_dict(meta, seed=4))
    m.to("cuda:0")
    x = bench.synth_images(B, S, seed=77).cuda()
    print(name, S, B, digest(m(x)))
wl = bench.build_workload("edge_n", 640, 64, seed=1)
ctx, x = wl["ctx"], wl["x"]
ctx.set_option("graph", 0); ctx.set_option("streams", 1)
print("edge_n 640 B=64 levels", digest(wl["model"](x)))
lay = np.median(np.stack([np.asarray(ctx.forward(x, timed=True)[1]) for _ in range(15)]), axis=0)
print("layer 0 (%s) eager ms: %.4f   sum of layers %.4f" % (wl["prog"].layers[0].name, lay[0], lay.sum()))
