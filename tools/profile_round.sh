#!/bin/bash
# Round profile set (run on the MI355X box through gpurun; outputs under gpurun_out/prof_$TAG, summaries are then
# copied into profiles/ by hand):   tools/profile_round.sh r01
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations)
#   2. PMC passes, each in its own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE, SQ set
#   3. un-profiled bench line + per-layer table
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $BENCH > $OUT/stats.log 2>&1
# same, one internal stream: every conv launch is the full-batch one that bench.py's roofline.avg_launch_ms times
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s --output-format csv -- $BENCH --streams 1 > $OUT/stats1.log 2>&1
PM="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph 0 --streams 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $PM > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $PM > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o p --output-format csv -- $PM > $OUT/pmc_sq.log 2>&1
cd $ROOT
python bench.py --steps 30 --warmup 5 --layers > $OUT/bench.json 2> $OUT/layers.txt
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (KB per dispatch, mean; gfx950: wide coalesced reads report 1/2 of the bytes)"
  python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
  echo; echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE (KB per dispatch, mean)"
  python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
  echo; echo "# rocprofv3 --pmc SQ_* (per dispatch, mean)"
  python tools/pmc_summary.py $(find $OUT/pmc_sq -name '*counter_collection.csv' | head -1)
} > $OUT/pmc_summary.txt
python tools/pmc_summary.py --json $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic.json
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_streams1.csv
ls -la $OUT; tail -1 $OUT/bench.json | cut -c1-300
