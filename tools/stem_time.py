import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from yololite_amd import _lib
wl = bench.build_workload("edge_n", 640, 64, seed=1)
ctx, x = wl["ctx"], wl["x"]
ctx.set_option("graph", 0); ctx.set_option("streams", 1)
lay = np.median(np.stack([np.asarray(ctx.forward(x, timed=True)[1]) for _ in range(15)]), axis=0)
print(os.path.basename(_lib.LIB_PATH), "layer0 ms %.4f" % lay[0])
