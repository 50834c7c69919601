#!/usr/bin/env python3
"""Developer aid: per-BLOCK comparison of the HIP forward (all slots kept: option reuse_slots 0) with the oracle's
backbone blocks (forward hooks) -- where does a deviation start?   python tools/debug_blocks.py MODEL S B [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import yololite_amd as ya  # noqa: E402
from yololite_amd.program import synth_state_dict, zoo_meta  # noqa: E402
from oracle import model as omodel  # noqa: E402


def main():
    name, S, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=seed)
    rng = np.random.RandomState(1234)
    x = torch.from_numpy(rng.randn(B, 3, S, S).astype(np.float32))
    orc = omodel.build_from_meta(meta).eval()
    orc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    feats = {}
    for n, m in orc.named_modules():
        if n.startswith("backbone.blocks.") and n.count(".") == 3:
            m.register_forward_hook(lambda mod, i, o, n=n: feats.__setitem__(n, o.detach()))
    with torch.no_grad():
        ref = orc(x)
    m = ya.build_model_from_meta(meta)
    m.load_state_dict(sd)
    m.to("cuda:0")
    ctx = m._ctx_for(S)
    ctx.set_option("reuse_slots", 0)
    ctx.set_option("fuse_head", 0)
    outs = m(x.cuda())
    prog = m.program
    for i, L in enumerate(prog.layers):
        if L.out_slot < 0:
            continue
        nm = L.name
        key = None
        for suf in (".conv_pwl", ".conv_pw", ".conv", ".ir", ".uib", ".pw_proj.conv"):
            if nm.endswith(suf) and nm[:-len(suf)] in feats:
                key = nm[:-len(suf)]
        if key is None or (nm.endswith(".conv_pw") and ".conv_pwl" not in nm and any(l.name == key + ".conv_pwl" for l in prog.layers)):
            continue
        t = ctx.read_slot(L.out_slot, B, prog.slots[L.out_slot]).permute(0, 3, 1, 2).cpu()
        r = feats[key]
        if t.shape != r.shape:
            continue
        err = (t - r).abs().max().item()
        print(f"{i:3d} {nm:40s} max|ref| {r.abs().max().item():9.4f}  max err {err:.3e}  rel {err / (r.abs().max().item() + 1e-12):.2e}")
    for l, (o, r) in enumerate(zip(outs, ref)):
        print("level", l, "err", (o.cpu() - r).abs().max().item(), "max", r.abs().max().item())


if __name__ == "__main__":
    main()
