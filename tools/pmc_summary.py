#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (mean per dispatch)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in rows.values() for c in k})
print("kernel".ljust(60), "n".rjust(4), *[c[-18:].rjust(19) for c in names])
for k, d in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    print(k.ljust(60), str(n).rjust(4), *[("%.4g" % (sum(d[c]) / max(len(d[c]), 1))).rjust(19) if c in d else "-".rjust(19) for c in names])
