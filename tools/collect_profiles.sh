#!/bin/bash
# copy the summaries of tools/profile_round.sh from gpurun_out/prof_$TAG into profiles/ (tracked):  tools/collect_profiles.sh r04
TAG=${1:-r04}
S=gpurun_out/prof_$TAG
D=profiles
cp $S/kernel_stats.csv $D/${TAG}_kernel_stats_bench_edge_n_b64.csv
cp $S/kernel_stats_streams1.csv $D/${TAG}_kernel_stats_bench_edge_n_b64_streams1.csv
cp $S/kernel_stats_full_batch_streams1.csv $D/${TAG}_kernel_stats_full_batch_edge_n_b64_streams1.csv
cp $S/layers.txt $D/${TAG}_layer_table_edge_n_b64.txt
cp $S/bench.json $D/${TAG}_bench_edge_n_b64.json
cp $S/bench_stress.json $D/${TAG}_bench_edge_n_b64_nms_stress.json
cp $S/pmc_summary.txt $D/${TAG}_pmc_summary.txt
cp $S/pmc_traffic.json $D/${TAG}_pmc_traffic.json
for N in yololite_m edge_m_seg yololite_m_v2; do
  cp $S/kernel_stats_$N.csv $D/${TAG}_kernel_stats_${N}_b32.csv
  cp $S/kernel_stats_full_batch_$N.csv $D/${TAG}_kernel_stats_full_batch_${N}_b32.csv
  cp $S/pmc_traffic_$N.json $D/${TAG}_pmc_traffic_${N}_b32.json
  cp $S/sq_$N.txt $D/${TAG}_sq_${N}_b32.txt
  grep -v amdgpu.ids $S/layers_$N.txt > $D/${TAG}_layer_table_${N}_b32.txt
  cp $S/bench_$N.json $D/${TAG}_bench_${N}_b32.json
done
cp $S/bench_yololite_m_winograd0.json $D/${TAG}_bench_yololite_m_b32_winograd0.json
cp $S/bench_yololite_m_winograd2.json $D/${TAG}_bench_yololite_m_b32_winograd2.json
cp $S/bench_yololite_m_v2_winograd0.json $D/${TAG}_bench_yololite_m_v2_b32_winograd0.json
grep -v amdgpu.ids $S/layers_yololite_m_winograd0.txt > $D/${TAG}_layer_table_yololite_m_b32_winograd0.txt
cp $S/bench_edge_m_seg_winograd0.json $D/${TAG}_bench_edge_m_seg_b32_winograd0.json
cp $S/bench_eval.json $D/${TAG}_bench_eval.json
cp $S/bench_track.json $D/${TAG}_bench_track.json
cp $S/calib/fetch_calibration.json $D/${TAG}_fetch_calibration.json
grep -v amdgpu.ids $S/layers.txt > $D/${TAG}_layer_table_edge_n_b64.txt
ls -la $D | grep $TAG
