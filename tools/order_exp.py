import sys, os, gc, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import argparse
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "20"]
# reuse bench's parser by calling main-like pieces
ap = None
import importlib
src = open(bench.__file__).read()
# build args through bench.main's parser: simplest is to copy defaults from a dry parse
def get_args():
    import types
    ns = {}
    code = src[src.index("def main():"):]
    # not robust; instead construct Namespace from known defaults
    return None
from types import SimpleNamespace
args = SimpleNamespace(gpus=1, steps=20, warmup=5, batch=64, model="edge_n", img=640, conf=0.4, iou=0.5, graph=1, no_cpu_baseline=True,
    fuse_dw="auto", fuse_stem=1, fuse_uib=0, seg=0, streams=0, in_flight=2, tile_m=0, layers=False, seed=-1, stress=0, nms_groups=0,
    hybrid=0, batch_levels=1, fuse_decode=1, opt=[], lanes=0, bf16=0, f16=0, store_f16=0, winograd=1, workload="predict", min_seconds=0.5,
    layer_reps=int(os.environ.get("LAYER_REPS", "15")), other_configs=0)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
import ctypes
_hip = ctypes.CDLL("libamdhip64.so")
_pre = []
for _ in range(int(os.environ.get("PRE", "0"))):
    h = ctypes.c_void_p(); _hip.hipStreamCreateWithFlags(ctypes.byref(h), 1); _pre.append(h)
if os.environ.get("PRE_USE") == "1":      # touch them: a HW queue is acquired at first use
    for h in _pre: _hip.hipStreamSynchronize(h)
if os.environ.get("PRE_FREE") == "1":
    for h in _pre: _hip.hipStreamDestroy(h)
order = sys.argv_order = os.environ.get("ORDER", "edge_n,yololite_m,edge_m_seg,v2").split(",")
cfg = {"edge_n": ("edge_n", 64, 0), "yololite_m": ("yololite_m", 32, 0), "edge_m_seg": ("edge_m", 32, 1), "v2": ("yololite_m_v2", 32, 0)}
for name in order:
    m, b, sg = cfg[name]
    o = bench.measure_predict(args, m, b, sg, dev, 0, 1, gather=False, min_seconds=0.5, max_blocks=8)
    print(name, o["value"], o["ms_per_step"], o["one_batch_in_flight"]["value"], "mem GB %.1f" % (torch.cuda.memory_allocated() / 1e9), flush=True)
    del o
    if os.environ.get("GC") == "1":
        gc.collect(); torch.cuda.empty_cache()
