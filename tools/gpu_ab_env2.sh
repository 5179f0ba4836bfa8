#!/bin/bash
# Interleaved A/B of two environments on the headline bench:  bash tools/gpu_ab_env2.sh <tag> "ENV_A" "ENV_B" [rounds] [bench args]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
A=$2; B=$3; N=${4:-3}; shift 4
for r in $(seq 1 $N); do
  env $A timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 "$@" > $O/a${r}.json 2>/dev/null; python tools/bench_line.py "A[$A]" < $O/a${r}.json | cut -c1-90
  env $B timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 "$@" > $O/b${r}.json 2>/dev/null; python tools/bench_line.py "B[$B]" < $O/b${r}.json | cut -c1-90
done
