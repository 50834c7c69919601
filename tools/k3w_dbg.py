"""Where does yl_conv_k3w_kernel differ from the direct kernel?  (debug aid)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import zoo_meta, synth_state_dict, _hip_for, _x, DEV, _lib
meta = zoo_meta("yololite_m_v2", 80, 640)
sd = synth_state_dict(meta, seed=9)
m = _hip_for(meta, sd)
ctx = m._ctx_for(640)
B = 3
x = _x(B, 640, seed=21).to(DEV)
ctx.set_option("winograd", 0); ctx.set_option("reuse_slots", 0); ctx.set_option("streams", 1)
prog = m.program
layers = prog.layers
res = {}
for dev in (_lib.DEV_K3W_OFF, 0):
    ctx.set_option("dev_select", dev)
    m(x)
    outs = []
    for li in (1, 2):
        l = layers[li]
        outs.append(ctx.read_slot(l.out_slot, B, prog.slots[l.out_slot]).clone())
    res[dev] = outs
for li, (u, v) in enumerate(zip(res[_lib.DEV_K3W_OFF], res[0])):
    bad = (u != v)
    print("layer", li + 1, layers[li + 1].name, tuple(u.shape), "differ", int(bad.sum()), "of", bad.numel(), "max", float((u - v).abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print("   first", idx[:8].tolist())
        print("   b", idx[:, 0].unique().tolist(), "y", idx[:, 1].unique().tolist()[:20], "x", idx[:, 2].unique().tolist()[:20], "c", idx[:, 3].unique().tolist())
u, v = res[_lib.DEV_K3W_OFF][0], res[0][0]
bad = (u != v).any(dim=3)                      # [B, 320, 320]
tb = bad.view(B, 80, 4, 80, 4).any(dim=4).any(dim=2).view(-1)     # per 4x4 tile
idx = tb.nonzero().view(-1)
print("wrong tiles", int(tb.sum()), "of", tb.numel(), "first", idx[:48].tolist())
d = (idx[1:] - idx[:-1])
print("gaps histogram", torch.unique(d, return_counts=True))
# inside a wrong tile: which pixels / channels
t0 = int(idx[0]); b0 = t0 // 6400; ty = (t0 % 6400) // 80; tx = t0 % 80
print("tile", t0, "diff map", (u[b0, 4*ty:4*ty+4, 4*tx:4*tx+4] != v[b0, 4*ty:4*ty+4, 4*tx:4*tx+4]).sum(dim=2).tolist())
print("old", u[b0, 4*ty, 4*tx, :6].tolist(), "new", v[b0, 4*ty, 4*tx, :6].tolist())
print("new at tile0 px(0,0),(1,1),(3,3):", v[0,0,0,:4].tolist(), v[0,1,1,:4].tolist(), v[0,3,3,:4].tolist())
t1 = int(idx[9]); b1 = t1 // 6400; ty1 = (t1 % 6400) // 80; tx1 = t1 % 80
print("new at tile", t1, v[b1,4*ty1,4*tx1,:4].tolist(), v[b1,4*ty1+2,4*tx1+1,:4].tolist())
