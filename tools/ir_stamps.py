#!/usr/bin/env python3
"""Phase timeline of yl_ir_kernel (second tile of every workgroup) from a stamp build:
   tools/build_variant.sh wstamp yl_convc.hip -DYL_WINO_STAMP=96 -DYL_STAMP_OH=40      (edge_n's 40x40 blocks: 96 expanded channels)
   YOLOLITE_HIP_LIB=_variants/libyololite_hip_wstamp.so python tools/ir_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta
from bench import synth_images
lib = _lib.load()
meta = zoo_meta("edge_n", 80, 640)
m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=1)); m.to("cuda:0")
ctx = m._ctx_for(640)
ctx.set_option("streams", 1)
x = synth_images(64, 640).cuda()
for _ in range(3):
    ctx.forward(x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8 * 64))()
lib.yl_debug_wino_stamps.argtypes = [C.c_void_p]
assert lib.yl_debug_wino_stamps(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 8, 64)[:, :4, :31]      # 4 waves, 6 slabs x 5 stamps + end
ok = (t > 0).all(axis=2)
print("workgroups/waves with a second tile:", int(ok.sum()), "of", ok.size)
d = np.diff(t, axis=2)[ok]
names = ["E (expansion MFMAs -> LDS)", "wait at barrier", "D (depthwise from LDS)", "P (projection MFMAs)", "loop end -> next slab"]
for j, n in enumerate(names):
    cols = [5 * i + j for i in range(1, 6)]
    print(f"{n:30s} mean {d[:, cols].mean():7.0f}  p50 {np.median(d[:, cols]):7.0f}  p90 {np.percentile(d[:, cols], 90):7.0f}")
print("slab period mean", d[:, 5:25].reshape(len(d), 4, 5).sum(axis=2).mean(), " tile (6 slabs) ", (t[ok][:, 30] - t[ok][:, 0]).mean())
for blk in (0, 100):
    for w in range(4):
        if ok[blk, w]:
            r = np.diff(t[blk, w])
            print(f"block {blk} wave {w}: " + " | ".join(" ".join(f"{r[5*i+j]:.0f}" for j in range(5)) for i in range(6)))
