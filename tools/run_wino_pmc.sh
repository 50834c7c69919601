#!/bin/bash
# MFMA microbenchmark + SQ counter passes of the yololite_m configuration (eager, one stream):  tools/run_wino_pmc.sh TAG [dev_select]
TAG=${1:-wino}; DV=${2:-0}
ROOT=$PWD; OUT=gpurun_out/$TAG; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak > $OUT/mfma_peak.jsonl
cat $OUT/mfma_peak.jsonl
export TMPDIR=/tmp
cd /tmp
PM="python $ROOT/bench.py --model yololite_m --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --min-seconds 0 --other-configs 0 --layer-reps 0 --opt dev_select=$DV"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $ROOT/$OUT/p1 -o p --output-format csv -- $PM > $ROOT/$OUT/p1.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $ROOT/$OUT/p2 -o p --output-format csv -- $PM > $ROOT/$OUT/p2.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_WAVES -d $ROOT/$OUT/p3 -o p --output-format csv -- $PM > $ROOT/$OUT/p3.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/st -o s --output-format csv -- $PM > $ROOT/$OUT/st.log 2>&1
cd $ROOT
for p in p1 p2 p3; do python tools/pmc_summary.py $(find $OUT/$p -name '*counter_collection.csv' | head -1) | grep -E "^kernel|wino"; echo; done > $OUT/sq_summary.txt
grep -i wino $(find $OUT/st -name '*kernel_stats.csv' | head -1) >> $OUT/sq_summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/st
cat $OUT/sq_summary.txt | cut -c1-300
tail -3 $OUT/p3.log | cut -c1-200
