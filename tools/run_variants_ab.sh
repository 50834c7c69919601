#!/bin/bash
# developer A/B on the GPU box: one bench configuration over several variant libraries (tools/build_variant.sh NAME ...),
# "base" = the in-tree library:   tools/run_variants_ab.sh "base gbs2 gbs6" --model yololite_m --batch 32 --steps 10
cd $GRAFT_REPO_ROOT
V=$1; shift
for r in 1 2; do
  for v in $V; do
    if [ $v == base ]; then unset YOLOLITE_HIP_LIB; else export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_$v.so; fi
    echo -n "$v: "; timeout -k 5 300 python bench.py --no-cpu-baseline --other-configs 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
