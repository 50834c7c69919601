"""The full-size parity test's sequence in a fresh process (python tools/flake_dbg.py [model B seg]); on a mismatch between the eager and the
bench schedule (graph replay, two chunk streams: its FIRST launch runs on freshly allocated arenas) say which one moved.  Round 6: caught a
rare first-launch error of the buffer-descriptor LDS-DMA form of yl_conv_dpw_kernel (5 of 100 processes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yololite_amd import _lib
NAME = sys.argv[1] if len(sys.argv) > 1 else "edge_n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
SEG = len(sys.argv) > 3 and sys.argv[3] == "1"
wl = bench.build_workload(NAME, 640, B, seed=1, dev="cuda:0", rank=0, seg=SEG)
ctx, x = wl["ctx"], wl["x"]
mo = bench.MAX_OUT
def pred():
    d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=mo)
    return d.clone(), c.clone()
def eager():
    for k, v in (("graph", 0), ("streams", 1), ("batch_levels", 0), ("fuse_decode", 0)):
        ctx.set_option(k, v)
def sched():
    for k, v in (("graph", 1), ("streams", 2), ("batch_levels", 1), ("fuse_decode", 1)):
        ctx.set_option(k, v)
eager(); e0 = pred()
for g in (4, 2, 1):
    ctx.set_option("nms_groups", g); pred()
ctx.set_option("nms_groups", 0)
sched(); g0 = pred(); g1 = pred()
ok = torch.equal(e0[1], g0[1]) and torch.equal(e0[0], g0[0])
if ok:
    print("OK"); sys.exit(0)
eager(); e1 = pred()
sched(); g2 = pred()
imgs = (g0[1] != e0[1]).nonzero().view(-1).tolist()
print("MISMATCH images", imgs, "counts eager/graph0", [(int(e0[1][b]), int(g0[1][b])) for b in imgs],
      "| e0==e1", bool(torch.equal(e0[0], e1[0]) and torch.equal(e0[1], e1[1])),
      "g0==g1", bool(torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])),
      "g1==e0", bool(torch.equal(g1[0], e0[0]) and torch.equal(g1[1], e0[1])),
      "g2==e0", bool(torch.equal(g2[0], e0[0]) and torch.equal(g2[1], e0[1])))
for b in imgs[:1]:
    n = min(int(e0[1][b]), int(g0[1][b]))
    rows = (e0[0][b, :n] != g0[0][b, :n]).any(dim=1).nonzero().view(-1)
    i = int(rows[0]) if len(rows) else n
    print("first differing row", i, "eager", e0[0][b, max(i - 1, 0):i + 2].tolist(), "graph", g0[0][b, max(i - 1, 0):i + 2].tolist())
