#!/bin/bash
# Interleaved A/B of one environment switch on the headline bench (pipelined):  bash tools/gpu_ab_env.sh <tag> VAR=VALUE [rounds]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
KV=${2:-EC_PIPE_HOLD=0}; N=${3:-3}
for r in $(seq 1 $N); do
  timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/a${r}.json 2>/dev/null; python tools/bench_line.py "default" < $O/a${r}.json | cut -c1-75
  env $KV timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/b${r}.json 2>/dev/null; python tools/bench_line.py "$KV" < $O/b${r}.json | cut -c1-75
done
