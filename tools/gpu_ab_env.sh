#!/bin/bash
# A/B of one environment switch: bash tools/gpu_ab_env.sh VAR  (VAR=1 vs VAR=0; timeline + three interleaved bench runs each)
cd $GRAFT_REPO_ROOT
V=$1; O=gpurun_out/ab_$V; mkdir -p $O
for v in 1 0; do
  echo "$V=$v"
  env $V=$v EC_TIMELINE=1 timeout 120 python tools/timeline_probe.py 2>&1 | grep timeline | tail -2 | cut -c1-330
done
for r in 1 2 3; do for v in 1 0; do
  env $V=$v timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/bench${r}_$v.json 2>/dev/null; echo -n "$V=$v "; python tools/bench_line.py < $O/bench${r}_$v.json | cut -c1-60
done; done
