#!/bin/bash
# the four configurations' bench lines, short form (kernel A/B runs):  tools/quick4.sh TAG [extra bench args]
TAG=${1:-q}; shift
cd $GRAFT_REPO_ROOT; O=gpurun_out/quick_$TAG; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 "$@" > $O/edge_n.json 2>/dev/null
python bench.py --model yololite_m --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers "$@" > $O/yololite_m.json 2> $O/layers_yololite_m.txt
python bench.py --model edge_m --seg 1 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers "$@" > $O/edge_m_seg.json 2> $O/layers_edge_m_seg.txt
python bench.py --model yololite_m_v2 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline --layers "$@" > $O/yololite_m_v2.json 2> $O/layers_yololite_m_v2.txt
for f in edge_n yololite_m edge_m_seg yololite_m_v2; do echo -n "$f: "; python tools/print_bench.py $O/$f.json; done
