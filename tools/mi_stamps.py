#!/usr/bin/env python3
"""Phase timings of yl_masks_image_kernel from a -DYL_MI_STAMP variant build:
   tools/build_variant.sh mistamp yl_post.hip -DYL_MI_STAMP
   YOLOLITE_HIP_LIB=_variants/libyololite_hip_mistamp.so python tools/mi_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from yololite_amd import _lib
lib = _lib.load()
wl = bench.build_workload("edge_m", 640, 32, seed=1, seg=True)
ctx, x = wl["ctx"], wl["x"]
ctx.set_option("graph", 0); ctx.set_option("streams", 1)
mo = bench.MAX_OUT
d, c, i = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=mo, want_idx=True)
arena = torch.empty((32 * mo * 640 * 80,), device="cuda:0", dtype=torch.uint8)
for _ in range(3):
    ctx.masks_image(d, c, i, packed=True, arena=arena)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 16 * 12))()
lib.yl_debug_mi_stamps.argtypes = [C.c_void_p]
assert lib.yl_debug_mi_stamps(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(2, 16, 12) / 100.0     # us
names = ["geom+det", "fill", "coef..A", "barrier A", "zero+eval", "barrier B", "units", "barrier C", "flush"]
for role, nm in enumerate(("box", "fill")):
    print(nm, "role, workgroup 7: per item, us since the item's start")
    for k in range(16):
        r = t[role, k]
        if r[0] == 0:
            break
        nxt = t[role, k + 1, 0] if k + 1 < 16 and t[role, k + 1, 0] else np.nan
        print(f"  item {k}: start {r[0] - t[role, 0, 0]:8.2f} | " + " ".join(
            f"{names[j]} {r[j + 1] - r[j]:6.2f}" for j in range(9) if r[j + 1] and r[j]) + f" | next item +{nxt - r[0]:.2f}")

bb = (C.c_ulonglong * 8192)()
lib.yl_debug_mi_blocks.argtypes = [C.c_void_p]
assert lib.yl_debug_mi_blocks(bb) == 0
q = np.array(bb[:], dtype=np.float64).reshape(4096, 2) / 100.0
t0 = q[q[:, 0] > 0, 0].min()
for lo, nm in ((2048, "box kernel (first launch)"), (0, "fill kernel (second launch)")):
    r = q[lo:lo + 2048]
    r = r[r[:, 0] > 0]
    st, en = r[:, 0] - t0, r[:, 1] - t0
    print(f"{nm}: {len(r)} workgroups, start pct 0/50/90/100 = {np.percentile(st, [0, 50, 90, 100]).round(1)}, "
          f"end pct 0/50/90/100 = {np.percentile(en, [0, 50, 90, 100]).round(1)}, duration median {np.median(en - st):.1f} max {(en - st).max():.1f} us")
