#!/bin/bash
# Small evidence artefacts of a round, one gpurun call:  tools/evidence_round.sh r06
#   mfma_peak      tools/mfma_peak.hip: what v_mfma_f32_16x16x4_f32 sustains on the whole chip
#   pipeline_probe tools/pipeline_probe.py: lanes x chunk streams table (serving.ServingPipeline's choice of 2 x 1)
#   overlap_stats  tools/overlap_stats.py on a rocprofv3 kernel trace of the default bench loop
#   ir_stamps / wino_stamps   in-kernel cycle stamps of the stamp variant builds (built beforehand with tools/build_variant.sh)
#   stream_alias   throughput against the number of streams created before the benchmark (yl_streams_overlap at work)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/evid_$TAG; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
timeout 600 python tools/pipeline_probe.py --steps 40 --reps 3 2>&1 | grep -v amdgpu.ids > $O/pipeline_probe.txt
mkdir -p $O/tr
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o p --output-format csv -- python bench.py --no-cpu-baseline --other-configs 0 --steps 30 --layer-reps 0 --min-seconds 0.3 > $O/tr.log 2>&1
python tools/overlap_stats.py $(find $O/tr -name '*kernel_trace.csv' | head -1) 20 > $O/overlap_stats.txt; rm -rf $O/tr
[ -f _variants/libyololite_hip_wstamp.so ] && YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_wstamp.so timeout 300 python tools/ir_stamps.py 2>&1 | grep -v amdgpu.ids > $O/ir_stamps.txt
[ -f _variants/libyololite_hip_wstamp328.so ] && YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_wstamp328.so timeout 300 python tools/wino_stamps.py 2>&1 | grep -v amdgpu.ids | tail -40 > $O/wino_stamps.txt
{ echo "# images/s (two batches in flight | one batch in flight) of the default bench loop after PRE throw-away HIP streams were created"
  echo "# first (tools/order_exp.py): before round 6 the same sweep read 46.2k / 38.9k / 40.4k / 46.3k|27.9k / ... (gpurun logs of the round)"
  for n in 0 1 2 3 4 5 6 7 8; do echo -n "PRE=$n: "; PRE=$n LAYER_REPS=0 ORDER=edge_n timeout 200 python tools/order_exp.py 2>&1 | grep edge_n; done; } > $O/stream_alias.txt
ls -la $O; cat $O/mfma_peak.txt; tail -12 $O/pipeline_probe.txt; cat $O/stream_alias.txt
