#!/bin/bash
# round 5: encoder as row chains (default) vs separate tiled GEMM + LayerNorm launches (EC_ENC_CHAIN=0) under the pipelined headline and the episode leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5e
mkdir -p $OUT
cd $R
for rep in 1 2; do
for c in 1 0; do
  EC_ENC_CHAIN=$c timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --no-alt --episode-images 64 > $OUT/bench_enc$c.json 2> $OUT/bench$c.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_enc$c.json"))
print("EC_ENC_CHAIN=$c: value", d["value"], "qkv frac", d["roofline"]["frac"], "episode", d["episode_cached"]["value"])
PY
done
done
