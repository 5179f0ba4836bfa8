#!/bin/bash
# the bench lines of every configuration with the CURRENT bench.py and library (no tests, no profiling)   -> gpurun_out/<tag>/
TAG=${1:-lines}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py < $OUT/bench.json | cut -c1-200
python bench.py --shots 5 --batch 16 --no-alt --steps 10 --cpu-batches 16 --cpu-runs 3 > $OUT/cfg4_5shot_b16.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vitl14 --image-size 384 --batch 8 --no-episode --no-alt --steps 10 --cpu-batches 8 --cpu-runs 3 > $OUT/cfg5_vitl_384_b8.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vits14 --image-size 224 --no-alt --steps 10 --cpu-batches 32 --cpu-runs 3 > $OUT/ref_vits_224_b32.json 2>> $OUT/bench.err
python bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 10 > $OUT/bench_bf16x3.json 2>> $OUT/bench.err
for f in cfg4_5shot_b16 cfg5_vitl_384_b8 ref_vits_224_b32 bench_bf16x3; do python tools/bench_line.py $f < $OUT/$f.json | cut -c1-200; done
