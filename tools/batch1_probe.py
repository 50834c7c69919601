import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from yololite_amd import _lib
wl = bench.build_workload("edge_n", 640, 1, seed=1, dev="cuda:0")
ctx, x = wl["ctx"], wl["x"]
d = torch.empty((1, 1024, 6), device="cuda"); c = torch.empty((1,), device="cuda", dtype=torch.int32)
for graph in (0, 1):
    for ts in (0, 1):
        ctx.set_option("graph", graph); ctx.set_option("time_split", ts)
        for _ in range(20):
            ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024, out=(d, c))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(200):
            ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024, out=(d, c))
        e1.record(); torch.cuda.synchronize()
        print("graph", graph, "time_split", ts, "gpu ms/call", e0.elapsed_time(e1) / 200, "wall", (time.perf_counter() - t0) * 5, ctx.last_timing() if ts else "")
