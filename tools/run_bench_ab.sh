#!/bin/bash
# developer A/B on the GPU box, UNTRACED: bench.py headline (no CPU baseline, no other configs) alternating between the
# in-tree library and an older one:  tools/run_bench_ab.sh _variants/libyololite_hip_head.so [rounds] [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OLD=$1; R=${2:-3}; shift 2
for r in $(seq $R); do
  for w in old new; do
    if [ $w == old ]; then export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/$OLD; else unset YOLOLITE_HIP_LIB; fi
    echo -n "$w "; timeout -k 5 200 python bench.py --no-cpu-baseline --other-configs 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['blocks']['images_per_sec_min'], d['blocks']['images_per_sec_max'])"
  done
done
