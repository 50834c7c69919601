#!/bin/bash
# developer aid on the GPU box: SQ counters of the fused head launch (tools/head_ab.py HEAD_AB_TRACE=1), two passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/hpmc; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/$1
export HEAD_AB_TRACE=1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $O/p1 -o p --output-format csv -- python tools/head_ab.py edge_n 64 > $O/p1.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $O/p2 -o p --output-format csv -- python tools/head_ab.py edge_n 64 > $O/p2.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/p3 -o p --output-format csv -- python tools/head_ab.py edge_n 64 > $O/p3.log 2>&1
for p in p1 p2 p3; do python tools/pmc_summary.py $(find $O/$p -name '*counter_collection.csv' | head -1) | grep "kernel  \|dpp\|stemblock\|kernel "; echo; done > $O/sq.txt
rm -rf $O/p1 $O/p2 $O/p3
cat $O/sq.txt | cut -c1-250
