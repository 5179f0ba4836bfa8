#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
SHAPES=qkv:20800:2304:768,sq4096:4096:4096:4096,fc2:20800:768:3072 VARIANTS=1000,9192,9384,1000,9192,9384 REPS=3 python tools/g8_lab.py 2>/dev/null > $O/setprio.txt; cat $O/setprio.txt
for i in 1 2 3; do
  EC_HEAD_PRE=0 python bench.py --no-cpu-baseline --no-episode --no-alt --no-pipeline --steps 30 2>/dev/null | python tools/bench_line.py ec_forward_pre0 | cut -c1-100
  python bench.py --no-cpu-baseline --no-episode --no-alt --no-pipeline --steps 30 2>/dev/null | python tools/bench_line.py ec_forward_pre1 | cut -c1-100
done > $O/head_pre_ab.txt; cat $O/head_pre_ab.txt
python -m pytest tests -m gpu -q --durations=15 > $O/tests.log 2>&1; tail -n 28 $O/tests.log
