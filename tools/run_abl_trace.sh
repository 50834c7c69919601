#!/bin/bash
# developer aid on the GPU box: rocprofv3 kernel durations of the head launch for ablation variants
#   tools/run_abl_trace.sh "intree abl1 abl63 ..." [abl_time args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
V=$1; shift
for v in $V; do
  if [ $v == intree ]; then unset YOLOLITE_HIP_LIB; else export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_$v.so; fi
  O=gpurun_out/abltrace_$v; rm -rf $O; mkdir -p $O
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python tools/abl_time.py "$@" > $O/log.txt 2>&1
  echo "== $v"; grep "ms/step" $O/log.txt
  f=$(find $O -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("dpw", "dpp", "stemblock", "nms", "ir_kernel<2")):
        print("   %-60s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $O
done
