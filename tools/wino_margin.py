#!/usr/bin/env python3
"""measured score error of the Winograd options against the ORACLE on yololite_m 640x640 B=32,
several weight seeds x all 32 images: max |hip score - oracle score| over all 8400 x 32 candidates for winograd 0 (direct
convolution, the parity path), 2 (selective: only the dense 3x3 layers of the finest level, smooth3.{0,3}) and 1 (all six).
Decision rule: an option becomes a default only if its error stays >= 4x inside the 1e-4 score bar (<= 2.5e-5).
    python tools/wino_margin.py [--seeds 1 2 3 4] [--model yololite_m] > profiles/rNN_winograd_margin.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import model as omodel  # noqa: E402


def score(levels, C=80):
    raw = torch.cat([l.reshape(l.shape[0], -1, l.shape[-1])[..., :5 + C] for l in levels], 1)
    return torch.sigmoid(raw[..., 4]) * torch.sigmoid(raw[..., 5:]).max(-1).values


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--model", default="yololite_m")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    rows = []
    for seed in a.seeds:
        wl = bench.build_workload(a.model, 640, a.batch, seed=seed, dev="cuda:0", rank=seed)
        orc = omodel.build_from_meta(wl["meta"]).eval()
        orc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in wl["sd"].items()}, strict=False)
        parts = []
        with torch.no_grad():
            for i in range(0, a.batch, 4):
                parts.append(orc(wl["x"][i:i + 4].cpu()))
        ref = score([torch.cat([p[l] for p in parts]) for l in range(len(parts[0]))])
        r = {"seed": seed}
        for w in (0, 2, 1):
            wl["ctx"].set_option("winograd", w)
            got = score([t.cpu() for t in wl["model"](wl["x"])])
            err = (got - ref).abs()
            r[f"winograd_{w}"] = {"max_abs_score_err": float(err.max()), "p9999": float(err.flatten().kthvalue(int(err.numel() * 0.9999)).values),
                                  "margin_to_1e-4": round(1e-4 / float(err.max()), 2)}
        wl["ctx"].set_option("winograd", 0)
        rows.append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
        del wl
        torch.cuda.empty_cache()
    worst = {f"winograd_{w}": max(r[f"winograd_{w}"]["max_abs_score_err"] for r in rows) for w in (0, 2, 1)}
    print(json.dumps({"model": a.model, "batch": a.batch, "img": 640, "candidates_per_seed": int(ref.numel()), "seeds": rows,
                      "worst_max_abs_score_err": worst,
                      "margin_to_1e-4": {k: round(1e-4 / v, 2) for k, v in worst.items()},
                      "rule": "default only if margin >= 4 (error <= 2.5e-5)"}, indent=1))


if __name__ == "__main__":
    main()
