#!/usr/bin/env python3
"""Print the NMS phase timings of a -DYL_NMS_STAMP variant build (see csrc/yl_post.hip):
   YOLOLITE_HIP_LIB=_variants/libyololite_hip_stamp.so python tools/nms_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta
from bench import synth_images
lib = _lib.load()
meta = zoo_meta("edge_n", 80, 640)
m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=int(os.environ.get('YL_SEED', '2')), head_noise=2.0)); m.to("cuda:0")
ctx = m._ctx_for(640)
x = synth_images(64, 640).cuda()
for _ in range(4):
    ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=300)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.yl_debug_nms_stamps.argtypes = [C.c_void_p]
assert lib.yl_debug_nms_stamps(buf) == 0
t = np.array(buf[:7], dtype=np.float64)
names = ["count", "compact", "sort", "segments-prep", "segments", "scan+write"]
print("survivors", buf[7], "kept", buf[8])
for i, n in enumerate(names):
    print(f"{n:14s} {(t[i + 1] - t[i]) / 100.0:8.2f} us")
print(f"{'total':14s} {(t[6] - t[0]) / 100.0:8.2f} us")
for gy in range(4):
    q = np.array(buf[16 * gy:16 * gy + 16], dtype=np.float64)
    if q[0] == 0: continue
    print(f"group {gy}: survivors {int(q[7])}  hist+assign {(q[1]-q[0])/100:.1f}  compact {(q[2]-q[1])/100:.1f}  sort {(q[3]-q[2])/100:.1f}  prep {(q[4]-q[3])/100:.1f}  segments {(q[5]-q[4])/100:.1f}  publish {(q[6]-q[5])/100:.1f}  merge {max(q[9]-q[6],0)/100:.1f}  | total {(max(q[6],q[9])-q[0])/100:.1f} us")
tb = (C.c_ulonglong * 128)()
if hasattr(lib, "yl_debug_nms_tstamps"):
    lib.yl_debug_nms_tstamps.argtypes = [C.c_void_p]
    lib.yl_debug_nms_tstamps(tb)
    tt = np.array(tb[:], dtype=np.float64)
    for k in range(0, 120, 6):
        if tt[k] == 0: break
        d = [(tt[k + i + 1] - tt[k + i]) / 100 for i in range(5)]
        print(f"chunk {k//6}: phaseA {d[0]:.2f} barrier {d[1]:.2f} phaseB {d[2]:.2f} barrier {d[3]:.2f} scan {d[4]:.2f} us")
# class-segment size distribution of image 0 (survivors per class before NMS)
outs = m(x[:1])
d = ya.decode_preds_anchorfree(outs, 640)
sc = torch.sigmoid(d["obj"][0, :, 0:1]) * torch.sigmoid(d["cls"][0])
s, c = sc.max(-1)
keep = s > 0.4
cnt = torch.bincount(c[keep], minlength=80).cpu().numpy()
print("survivors/class sorted:", sorted(cnt.tolist(), reverse=True)[:24], "sum", cnt.sum())
