#!/bin/bash
# Round profile set (run ONCE per round on the MI355X box through gpurun, on the final sources; outputs under
# gpurun_out/prof_$TAG, summaries are then copied into profiles/ with tools/collect_profiles.sh):  tools/profile_round.sh r04
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations), 2 streams and 1 stream, + the
#      full-batch-only statistics of the 1-stream trace (tools/kernel_stats_full.py)
#   2. PMC passes, each in its own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE (predict steps only:
#      --layer-reps 0, so that the per-step traffic is the sum over a step's dispatches), SQ set
#   3. un-profiled bench line + per-layer table; NMS-stress line; the other configs (kernel stats + bench + layer table)
#   4. FETCH_SIZE / WRITE_SIZE calibration on independent kernels (tools/fetch_calib.sh)
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --other-configs 0 --min-seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $BENCH > $OUT/stats.log 2>&1
# same, one internal stream: every conv launch is the full-batch one that bench.py's roofline.avg_launch_ms times
rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o s --output-format csv -- $BENCH --streams 1 > $OUT/stats1.log 2>&1
PM="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --other-configs 0 --min-seconds 0 --layer-reps 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $PM > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $PM > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o p --output-format csv -- $PM > $OUT/pmc_sq.log 2>&1
for cfg in "yololite_m 0" "edge_m 1" "yololite_m_v2 0"; do
  set -- $cfg
  N=$1; [ "$2" == "1" ] && N=${1}_seg
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$N -o s --output-format csv -- python $ROOT/bench.py --model $1 --seg $2 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --min-seconds 0 > $OUT/stats_$N.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcf_$N -o p --output-format csv -- python $ROOT/bench.py --model $1 --seg $2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --min-seconds 0 --layer-reps 0 > $OUT/pmcf_$N.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcw_$N -o p --output-format csv -- python $ROOT/bench.py --model $1 --seg $2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --min-seconds 0 --layer-reps 0 > $OUT/pmcw_$N.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmcs_$N -o p --output-format csv -- python $ROOT/bench.py --model $1 --seg $2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 --min-seconds 0 --layer-reps 0 > $OUT/pmcs_$N.log 2>&1
done
cd $ROOT
python bench.py --steps 30 --warmup 5 --layers > $OUT/bench.json 2> $OUT/layers.txt
python bench.py --steps 30 --warmup 5 --stress 1 --no-cpu-baseline > $OUT/bench_stress.json 2> /dev/null
python bench.py --model yololite_m --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $OUT/bench_yololite_m.json 2> $OUT/layers_yololite_m.txt
python bench.py --model edge_m --seg 1 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $OUT/bench_edge_m_seg.json 2> $OUT/layers_edge_m_seg.txt
python bench.py --model yololite_m_v2 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $OUT/bench_yololite_m_v2.json 2> $OUT/layers_yololite_m_v2.txt
# option "winograd": 1 (every eligible layer) is the library default since round 4; labelled lines for 0 (direct everywhere)
# and 2 (the finest level's >= 64-channel layers only)
python bench.py --model yololite_m --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --winograd 0 --layers > $OUT/bench_yololite_m_winograd0.json 2> $OUT/layers_yololite_m_winograd0.txt
python bench.py --model yololite_m --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --winograd 2 > $OUT/bench_yololite_m_winograd2.json 2> /dev/null
python bench.py --model yololite_m_v2 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --winograd 0 > $OUT/bench_yololite_m_v2_winograd0.json 2> /dev/null
python bench.py --model edge_m --seg 1 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --winograd 0 > $OUT/bench_edge_m_seg_winograd0.json 2> /dev/null
python bench.py --workload eval > $OUT/bench_eval.json 2> /dev/null
python bench.py --workload track > $OUT/bench_track.json 2> /dev/null
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (KB per dispatch, mean; gfx950: reads are counted at ~1/2 of their bytes, see rNN_fetch_calibration.json)"
  python tools/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
  echo; echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE (KB per dispatch, mean)"
  python tools/pmc_summary.py $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
  echo; echo "# rocprofv3 --pmc SQ_* (per dispatch, mean)"
  python tools/pmc_summary.py $(find $OUT/pmc_sq -name '*counter_collection.csv' | head -1)
} > $OUT/pmc_summary.txt
python tools/pmc_summary.py --json --label "B=64 edge_n 640x640" $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic.json
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_streams1.csv
python tools/kernel_stats_full.py $(find $OUT/stats1 -name '*kernel_trace.csv' | head -1) > $OUT/kernel_stats_full_batch_streams1.csv
for N in yololite_m edge_m_seg yololite_m_v2; do
  cp $(find $OUT/stats_$N -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$N.csv
  python tools/kernel_stats_full.py $(find $OUT/stats_$N -name '*kernel_trace.csv' | head -1) > $OUT/kernel_stats_full_batch_$N.csv
  python tools/pmc_summary.py --json --label "B=32 $N 640x640" $(find $OUT/pmcf_$N -name '*counter_collection.csv' | head -1) $(find $OUT/pmcw_$N -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic_$N.json
  { echo "# rocprofv3 --pmc SQ_* (per dispatch, mean), $N B=32, eager one-stream predict steps, csrc_sha256 $(python -c 'import bench; print(bench.csrc_digest())')"
    python tools/pmc_summary.py $(find $OUT/pmcs_$N -name '*counter_collection.csv' | head -1); } > $OUT/sq_$N.txt
done
tools/fetch_calib.sh gpurun_out/prof_$TAG/calib > $OUT/calib.log 2>&1
# keep the merge-back small: raw traces are not needed
rm -rf $OUT/stats $OUT/stats1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/stats_* $OUT/pmcf_* $OUT/pmcw_* $OUT/pmcs_* $OUT/calib/f $OUT/calib/w
ls -la $OUT; tail -1 $OUT/bench.json | cut -c1-300
