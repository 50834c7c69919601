#!/usr/bin/env python3
"""Phase timeline of yl_conv_wino2_kernel from a -DYL_WINO_STAMP=<Cin> variant build (second item of every workgroup,
smooth3-sized layers):
   tools/build_variant.sh wstamp yl_convc.hip -DYL_WINO_STAMP=328
   YOLOLITE_HIP_LIB=_variants/libyololite_hip_wstamp.so python tools/wino_stamps.py [dev_select]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta
from bench import synth_images
dv = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib = _lib.load()
MODEL = os.environ.get("STAMP_MODEL", "yololite_m")
meta = zoo_meta(MODEL, 80, 640)
m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=1)); m.to("cuda:0")
ctx = m._ctx_for(640)
ctx.set_option("streams", 1); ctx.set_option("dev_select", dv)
x = synth_images(32, 640).cuda()
for _ in range(3):
    ctx.forward(x)
torch.cuda.synchronize()
n = 256 * 8 * 64
buf = (C.c_ulonglong * n)()
lib.yl_debug_wino_stamps.argtypes = [C.c_void_p]
assert lib.yl_debug_wino_stamps(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 8, 64)
names = (["m-tile 0 (B + 24 MFMAs + U request)", "m-tile 1", "m-tile 2", "m-tile 3: reads, B -> barrier", "barrier", "window request + next reads", "m-tile 3 MFMAs -> next block"]
         if MODEL.startswith("yololite") else ["wait + barrier", "requests", "group 0 (+ taps 0-2)", "group 1 (+ taps 3-5)", "group 2 (+ taps 6-8, act)", "group 3", "-> next block"])
NS = 7
for blk in (0, 100):
    for w in range(8):
        r = t[blk, w]
        k = int((r > 0).sum())
        d = np.diff(r[:k])
        print(f"block {blk} wave {w}: " + " | ".join(" ".join(f"{d[NS*i+j]:.0f}" for j in range(NS) if NS*i+j < len(d)) for i in range(min(k // NS, 4))))
d = np.diff(t[:, :, :63], axis=2)
ok = (t[:, :, :63] > 0).all(axis=2)
print("blocks/waves complete:", int(ok.sum()), "of", ok.size)
dm = d[ok].reshape(-1, 62)
for j in range(NS):
    cols = [NS * i + j for i in range(1, 9) if NS * i + j < 62]
    print(f"{names[j]:28s} mean {dm[:, cols].mean():8.0f}  p50 {np.median(dm[:, cols]):8.0f}  p90 {np.percentile(dm[:, cols], 90):8.0f}")
print("k-block period mean", dm[:, NS:NS * 8].reshape(len(dm), -1, NS).sum(axis=2).mean())
