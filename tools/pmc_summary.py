#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (mean per dispatch)."""
import csv, sys, collections, json

if len(sys.argv) > 1 and sys.argv[1] == "--json":
    # --json [--label TEXT] FETCH.csv WRITE.csv : per-kernel KB per dispatch of both counters (profiles/rNN_pmc_traffic*.json).
    # mean AND median: the benchmark run also dispatches every kernel once on the B=8 head-calibration batch, which the
    # median ignores.  csrc_sha256 ties the pass to the kernel sources it was taken on (bench.csrc_digest).
    import os, statistics
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    argv = sys.argv[2:]
    label = "B=64 edge_n 640x640"
    if argv and argv[0] == "--label":
        label, argv = argv[1], argv[2:]
    out = collections.defaultdict(dict)
    for path in argv:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                out[k][c + "_KB_mean"] = sum(v) / len(v)
                out[k][c + "_KB_median"] = statistics.median(v)
                out[k]["dispatches"] = len(v)
    try:
        from bench import csrc_digest
        sha = csrc_digest()
    except Exception:
        sha = None
    # traffic of ONE step: the passes run predict steps only (bench.py --layer-reps 0), so every full-batch dispatch of a
    # yl_* kernel belongs to a step; dispatches on the 8-image head-calibration batch (< half the kernel's median) are
    # dropped; steps = the full-batch dispatches of the network's entry kernel (one per step)
    per_step = None
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    ndisp, steps = 0, 0
    for path in argv:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "yl_" not in k or "yl_spin_kernel" in k:          # (the stream-overlap probe of context / pipeline creation)
                continue
            for c, v in d.items():
                if c not in tot:
                    continue
                med = statistics.median(v)
                full = [x for x in v if x >= 0.5 * med]
                tot[c] += sum(full)
                if c == "WRITE_SIZE":
                    ndisp += len(full)
                    if "yl_stemblock_kernel" in k or "yl_stem_mfma_kernel" in k or "yl_stemdw_kernel" in k:
                        steps = max(steps, len(full))
    if steps:
        per_step = {"fetch_kb": tot["FETCH_SIZE"] / steps, "write_kb": tot["WRITE_SIZE"] / steps, "steps": steps,
                    "dispatches_per_step": round(ndisp / steps, 1)}
    print(json.dumps({"note": "rocprofv3 --pmc, separate passes, eager launches (--graph 0 --streams 1 --layer-reps 0: predict "
                              "steps only), " + label + "; "
                              "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 of their bytes (MI355X_MICROARCH.md HBM section)",
                      "csrc_sha256": sha, "per_step": per_step, "kernels": out}, indent=1))
    sys.exit(0)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in rows.values() for c in k})
print("kernel".ljust(60), "n".rjust(4), *[c[-18:].rjust(19) for c in names])
for k, d in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    print(k.ljust(60), str(n).rjust(4), *[("%.4g" % (sum(d[c]) / max(len(d[c]), 1))).rjust(19) if c in d else "-".rjust(19) for c in names])
