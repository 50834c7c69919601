#!/usr/bin/env python3
"""Phase timeline of the block-cooperative depthwise kernel from a -DYL_DWC_STAMP=<Cin> variant build:
   tools/build_variant.sh stamp yl_convc.hip -DYL_DWC_STAMP=256
   YOLOLITE_HIP_LIB=_variants/libyololite_hip_stamp.so python tools/dwc_stamps.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta
from bench import synth_images
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.load()
meta = zoo_meta("edge_n", 80, 640)
m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=1)); m.to("cuda:0")
ctx = m._ctx_for(640)
ctx.set_option("streams", 1)
x = synth_images(B, 640).cuda()
for _ in range(3):
    ctx.forward(x)
torch.cuda.synchronize()
n = 1024 * 8 * 32
buf = (C.c_ulonglong * n)()
lib.yl_debug_dwc_stamps.argtypes = [C.c_void_p]
assert lib.yl_debug_dwc_stamps(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(1024, 8, 32)
used = t[:, 0, 0] > 0
nb = int(used.sum())
print(f"blocks stamped {nb}  (waves 0-3: depthwise producers, 4-7: GEMM consumers; cycles)")
for blk in (0, nb // 2, nb - 1):
    for w in (0, 1, 4, 5):
        r = t[blk, w]
        fine = r[20:30].copy()
        r = r.copy(); r[20:] = 0
        k = int((r > 0).sum())
        if w >= 4 and fine[0] > 0:
            print(f"   consumer tile 2: start->acc-init {fine[0]-r[2+2*2+1]:.0f} mfma-loop {fine[1]-fine[0]:.0f} epilogue-> {r[2+2*3]-fine[1]:.0f}")
        if w < 4 and fine[0] > 0:
            f = np.diff(fine)
            print(f"   tile 2 half-steps: " + " | ".join(f"lds-store {f[5*h]:.0f} next-load+fence {f[5*h+1]:.0f} taps {f[5*h+2]:.0f} store {f[5*h+3]:.0f}" for h in range(2)))
        d = np.diff(r[:k])
        print(f"block {blk} wave {w}: prologue {d[0]:.0f} |", " ".join(
            f"[work {d[1+2*i]:.0f} bar {d[2+2*i]:.0f}]" for i in range(min((k - 2) // 2, 9))))
