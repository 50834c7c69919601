#!/bin/bash
# developer experiment on the GPU box (variant library built with -DYL_VARIANT_SKIP_LAYERS, results WRONG): headline step
# time with a range of layers left out = the upper bound of what speeding those launches up can buy.
#   tools/run_skip_ab.sh "none 12-22 2-9 ..." [bench args]
cd $GRAFT_REPO_ROOT
export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_skip.so YL_BENCH_ALLOW_EMPTY=1
R=$1; shift
for r in $R; do
  if [ $r == none ]; then unset YL_SKIP; else export YL_SKIP=$r; fi
  echo -n "skip $r: "; timeout -k 5 300 python bench.py --no-cpu-baseline --other-configs 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
