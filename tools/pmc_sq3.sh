#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, kernel-trace only) incl. scalar / branch instruction counts and kernel durations,
# eager full-batch launches of one bench configuration:   tools/pmc_sq3.sh OUTDIR --model edge_n --batch 64
OUT=$1; shift
ROOT=$PWD
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
cd /tmp
PM="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --in-flight 1 --min-seconds 0 --other-configs 0 --layer-reps 0 $@"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $ROOT/$OUT/p1 -o p --output-format csv -- $PM > $ROOT/$OUT/p1.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $ROOT/$OUT/p2 -o p --output-format csv -- $PM > $ROOT/$OUT/p2.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SENDMSG -d $ROOT/$OUT/p3 -o p --output-format csv -- $PM > $ROOT/$OUT/p3.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/p4 -o p --output-format csv -- $PM > $ROOT/$OUT/p4.log 2>&1
cd $ROOT
for p in p1 p2 p3; do python tools/pmc_summary.py $(find $OUT/$p -name '*counter_collection.csv' | head -1); echo; done > $OUT/sq_summary.txt
cp $(find $OUT/p4 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/sq_summary.txt | cut -c1-260 | head -90
