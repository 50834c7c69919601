#!/bin/bash
# developer aid on the GPU box: kernel durations of the head launches, one stream (tools/head_ab.py HEAD_AB_TRACE)
#   tools/run_head_trace.sh [fuse_head values, default 0,1] [lib]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/hp; rm -rf $O; mkdir -p $O
[ -n "$2" ] && export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/$2
HEAD_AB_TRACE=${1:-0,1} timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python tools/head_ab.py edge_n 64 > $O/log.txt 2>&1
grep "dpp\|dwh_kernel<6, 3\|pwt_kernel<6, 2, true\|Name" $O/t_kernel_stats.csv
