#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, kernel-trace only) for one bench configuration, eager full-batch launches:
#   tools/pmc_sq.sh OUTDIR --model yololite_m --batch 32 [--seg 1]
OUT=$1; shift
ROOT=$PWD
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
cd /tmp
PM="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --min-seconds 0 --other-configs 0 $@"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $ROOT/$OUT/p1 -o p --output-format csv -- $PM > $ROOT/$OUT/p1.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $ROOT/$OUT/p2 -o p --output-format csv -- $PM > $ROOT/$OUT/p2.log 2>&1
cd $ROOT
for p in p1 p2; do python tools/pmc_summary.py $(find $OUT/$p -name '*counter_collection.csv' | head -1); echo; done > $OUT/sq_summary.txt
rm -rf $OUT/p1 $OUT/p2
cat $OUT/sq_summary.txt | cut -c1-260 | head -60
