"""Where does yl_conv_wino2_kernel differ from yl_conv_wino_kernel?  (debug aid)  python tools/wino_dbg.py edge_m 320 1"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import zoo_meta, synth_state_dict, _hip_for, _x, DEV, _lib
name, S, seg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
meta = zoo_meta(name, 80, S, **(dict(seg=True) if seg else {}))
sd = synth_state_dict(meta, seed=6)
m = _hip_for(meta, sd)
ctx = m._ctx_for(S)
x = _x(B, S, seed=13).to(DEV)
def run():
    out = m(x)
    return ([t.clone() for t in out[0]] + [out[1].clone()]) if seg else [t.clone() for t in out]
ctx.set_option("winograd", 1)
ctx.set_option("dev_select", _lib.DEV_WINO_V1)
first = run()
for shape in (0, 3):
    ctx.set_option("dev_select", shape << _lib.DEV_WINO_SHAPE_SHIFT)
    got = run()
    for i, (u, v) in enumerate(zip(first, got)):
        d = (u - v).abs()
        bad = (u != v)
        print("shape", shape, "tensor", i, tuple(u.shape), "differ", int(bad.sum()), "of", bad.numel(), "max", float(d.max()), "nan", int(torch.isnan(v).sum()))
        if bad.any() and u.dim() == 4:
            idx = bad.nonzero()
            print("   first", idx[:6].tolist(), "last", idx[-3:].tolist())
            # histogram over last two dims (y, x) if NCHW-like
            ys = idx[:, -2].unique().tolist(); xs = idx[:, -1].unique().tolist()
            print("   dim-2 values", ys[:40], "dim-1 values", xs[:40])
