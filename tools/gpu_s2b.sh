#!/bin/bash
# zero-filled vs random operands on the same binary (DVFS: the instruction stream is identical, the sustained clock is not)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2b; mkdir -p $O
VARIANTS=0,0 REPS=3 SHAPES="qkv:20800:2304:768,sq4096:4096:4096:4096,fc2:20800:768:3072" timeout 200 python tools/g8_lab.py 2>&1 | tee $O/lab_random.txt | tail -n 9
ZERO=1 VARIANTS=0,0 REPS=3 SHAPES="qkv:20800:2304:768,sq4096:4096:4096:4096,fc2:20800:768:3072" timeout 200 python tools/g8_lab.py 2>&1 | tee $O/lab_zero.txt | tail -n 9
timeout 300 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json | cut -c1-200
