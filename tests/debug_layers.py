"""Debug aid (not a test): compares every activation slot of the HIP program with the outputs of the
oracle's modules (forward hooks) and prints which fused layers match.  Usage on the GPU box:
    python tests/debug_layers.py edge_n 128"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd.program import synth_state_dict, zoo_meta, make_meta
from oracle import model as omodel

name, S = sys.argv[1], int(sys.argv[2])
meta = zoo_meta(name, 80, S) if name in ("edge_n", "edge_m", "yololite_m") else \
    make_meta(arch="YOLOLiteMS_CPU", backbone=name, num_classes=3, fpn_channels=16, depth_multiple=0.5, img_size=S)
sd = synth_state_dict(meta, 0)
orc = omodel.build_from_meta(meta).eval()
orc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rec = {}
for n, mod in orc.named_modules():
    mod.register_forward_hook(lambda m, i, o, n=n: rec.__setitem__(n, o) if torch.is_tensor(o) else None)
x = torch.randn(2, 3, S, S, generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    ref = orc(x)
m = ya.build_model_from_meta(meta); m.load_state_dict(sd); m.to("cuda:0")
outs = m(x.cuda())
ctx, prog = m._ctx_for(S), m.program
for i, l in enumerate(prog.layers):
    if l.out_slot < 0:
        continue
    shp = prog.slots[l.out_slot]
    t = ctx.read_slot(l.out_slot, 2, shp).cpu().permute(0, 3, 1, 2)
    best = (1e9, None)
    for n, o in rec.items():
        if tuple(o.shape) == tuple(t.shape):
            e = (o - t).abs().max().item()
            if e < best[0]:
                best = (e, n)
    print(f"{i:3d} {l.name:45s} k{l.k} s{l.stride} dw{l.dw_k} {str(shp):18s} best_err={best[0]:.3e} ~ {best[1]}")
for l, (o, r) in enumerate(zip(outs, ref)):
    print("level", l, (o.cpu() - r).abs().max().item())
