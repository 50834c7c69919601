#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calib.hip) on the MI355X box: tools/fetch_calib.sh OUTDIR
OUT=${1:-gpurun_out/fetch_calib}
ROOT=$PWD
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/$OUT/f -o p --output-format csv -- /tmp/fetch_calib > $ROOT/$OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/$OUT/w -o p --output-format csv -- /tmp/fetch_calib > $ROOT/$OUT/w.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
BYTES = 768 << 20
res = {}
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    fn = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[ctr + "_KB_mean"] = sum(v) / len(v)
for k, v in sorted(res.items()):
    rd = k.startswith("calib_x")
    key = "FETCH_SIZE_KB_mean" if rd else "WRITE_SIZE_KB_mean"
    actual = BYTES if k != "calib_w4s" else (BYTES // 384 // 16) * 16 * 384
    v["actual_bytes"] = actual
    v["counter_over_actual"] = round(v[key] * 1024.0 / actual, 4)
    print(f"{k:12s} {key:18s} {v[key] * 1024 / 1e6:10.1f} MB counted / {actual / 1e6:8.1f} MB moved = {v['counter_over_actual']}")
json.dump({"bytes": BYTES, "kernels": res, "note": "counter_over_actual: multiply a measured FETCH_SIZE/WRITE_SIZE (KB*1024) by 1/this to get bytes for that access pattern"},
          open(f"{out}/fetch_calibration.json", "w"), indent=1)
PY
