#!/usr/bin/env python3
"""Developer aid: step time of the bench workload (graph replay, one stream) under the library named by YOLOLITE_HIP_LIB
-- for ablation variants (tools/build_variant.sh ... -DDPW_ABL=<bits>: results wrong, timing only).  conf is set above every
score so that the NMS sees no survivor in any variant: differences are the conv launches'.
    YOLOLITE_HIP_LIB=_variants/libyololite_hip_abl1.so python tools/abl_time.py [model] [B] [seg]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YL_BENCH_ALLOW_EMPTY"] = "1"
import bench
from yololite_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "edge_n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.build_workload(name, 640, B, seed=1)
ctx, x = wl["ctx"], wl["x"]
ctx.set_option("streams", 1); ctx.set_option("graph", 1)
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
res = []
for rep in range(3):
    for _ in range(10): ctx.predict(x, _lib.POST_MAIN, 1e30, 0.5, per_class_cap=300, max_out=1024)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.predict(x, _lib.POST_MAIN, 1e30, 0.5, per_class_cap=300, max_out=1024)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 200 * 1e3)
print(os.path.basename(os.environ.get("YOLOLITE_HIP_LIB", "in-tree")), " ".join(sys.argv[3:]), "ms/step", " ".join("%.4f" % r for r in res), flush=True)
