import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta
from bench import synth_images
meta = zoo_meta("edge_n", 80, 640)
x = synth_images(16, 640).cuda()
for seed in range(12):
    m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=seed, head_noise=2.0)); m.to("cuda:0")
    ctx = m._ctx_for(640)
    d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=300)
    d2, c2 = ctx.predict(x, _lib.POST_MAIN, 0.02, 0.5, per_class_cap=300, max_out=300)
    print("seed", seed, "mean dets @0.4", float(c.float().mean()), "@0.02", float(c2.float().mean()), flush=True)
