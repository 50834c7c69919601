#!/bin/bash
# Final bench lines of a round on sources whose PMC passes are already under profiles/ (traffic fields resolve):
#   tools/bench_lines.sh r06     -> gpurun_out/lines_r06/*.json (+ layer tables), copied to profiles/ by hand
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
O=gpurun_out/lines_$TAG; [ "$ONLY_LOWP" == "1" ] || rm -rf $O; mkdir -p $O
[ "$ONLY_LOWP" == "1" ] || python bench.py --steps 20 --warmup 5 --layers > $O/bench_edge_n_b64.json 2> $O/layers_edge_n_b64.txt
[ "$ONLY_LOWP" == "1" ] || python bench.py --model yololite_m --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $O/bench_yololite_m_b32.json 2> $O/layers_yololite_m_b32.txt
[ "$ONLY_LOWP" == "1" ] || python bench.py --model edge_m --seg 1 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $O/bench_edge_m_seg_b32.json 2> $O/layers_edge_m_seg_b32.txt
[ "$ONLY_LOWP" == "1" ] || python bench.py --model yololite_m_v2 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers > $O/bench_yololite_m_v2_b32.json 2> $O/layers_yololite_m_v2_b32.txt
# reduced-precision lines (never the headline): fp16 operands + fp16 activation tensors in HBM, and fp16 operands only
for cfg in "edge_n 0 64" "yololite_m 0 32" "edge_m 1 32" "yololite_m_v2 0 32"; do
  set -- $cfg
  N=$1; [ "$2" == "1" ] && N=${1}_seg
  python bench.py --model $1 --seg $2 --batch $3 --steps 20 --warmup 3 --no-cpu-baseline --other-configs 0 --store-f16 1 > $O/bench_${N}_b$3_store_f16.json 2> /dev/null
  python bench.py --model $1 --seg $2 --batch $3 --steps 20 --warmup 3 --no-cpu-baseline --other-configs 0 --f16 1 > $O/bench_${N}_b$3_mfma_f16.json 2> /dev/null
done
[ "$ONLY_LOWP" == "1" ] || python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --in-flight 3 > $O/bench_edge_n_b64_in_flight3.json 2> /dev/null
[ "$ONLY_LOWP" == "1" ] || python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --in-flight 1 > $O/bench_edge_n_b64_in_flight1.json 2> /dev/null
for f in $O/*.json; do echo -n "$(basename $f): "; python tools/print_bench.py $f; done
