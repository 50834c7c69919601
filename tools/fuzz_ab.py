#!/usr/bin/env python3
"""Developer fuzz: random image sizes / batches, edge_n (and a few edge_m / yololite_m): raw levels and detections of the
default kernel selection must be the same bits as with the round-3 run-time fusions off ("fuse_head" 0) and the
round-1 kernel set ("tile_m" 6; yololite_m's dense 3x3 differs in summation order there: tolerance).
    python tools/fuzz_ab.py [n_cases] [seed]"""
import os, sys, random
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import synth_state_dict, zoo_meta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    name = rng.choice(["edge_n"] * 6 + ["edge_m", "yololite_m"])
    S = 32 * rng.randint(2, 22 if name == "edge_n" else 12)
    B = rng.randint(1, 6 if S <= 384 else 3)
    meta = zoo_meta(name, 80, S)
    m = ya.build_model_from_meta(meta); m.load_state_dict(synth_state_dict(meta, seed=case + 3, head_noise=2.0)); m.to("cuda:0")
    x = bench.synth_images(B, S, seed=100 + case).cuda()
    ctx = m._ctx_for(S)
    def run():
        lv = [t.clone() for t in m(x)]
        d, c = ctx.predict(x, _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300, max_out=512)
        return lv, d.cpu().numpy().copy(), c.cpu().numpy().copy()
    ref = run()
    ok = True
    for opt, val in (("fuse_head", 0), ("tile_m", 6), ("streams", 1)):
        ctx.set_option(opt, val)
        got = run()
        ctx.set_option(opt, {"fuse_head": 1, "tile_m": 0, "streams": 2}[opt])
        loose = name == "yololite_m" and opt == "tile_m"
        for a, b in zip(ref[0], got[0]):
            same = torch.allclose(a, b, atol=2e-5, rtol=1e-5) if loose else torch.equal(a, b)
            if not same:
                ok = False
                print("  LEVEL MISMATCH", name, S, B, opt, float((a - b).abs().max()))
        if not loose and not (np.array_equal(ref[2], got[2]) and all(np.array_equal(ref[1][i, :ref[2][i]].view(np.uint32), got[1][i, :got[2][i]].view(np.uint32)) for i in range(B))):
            ok = False
            print("  DET MISMATCH", name, S, B, opt, ref[2].tolist(), got[2].tolist())
    bad += 0 if ok else 1
    print(case, name, S, B, "dets", int(ref[2].sum()), "OK" if ok else "FAIL")
    del m, ctx
print("failures:", bad)
