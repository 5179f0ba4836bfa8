#!/bin/bash
# round 5: CU reservation experiment (EC_G8_RESERVE) on the pipelined headline and the episode leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5d
mkdir -p $OUT
cd $R
for rep in 1 2; do
for r in 0 8 16 32; do
  EC_G8_RESERVE=$r timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --no-alt --episode-images 64 > $OUT/bench_res$r.json 2> $OUT/bench$r.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_res$r.json"))
print("reserve $r: value", d["value"], "qkv frac", d["roofline"]["frac"], "episode", d["episode_cached"]["value"])
PY
done
done
