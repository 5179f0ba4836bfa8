#!/bin/bash
# The conforming mode (fp16x2 / bf16x3 since round 6; MODE="bf16x3" for round 5's) on every benched configuration: pairwise + episode protocol   -> gpurun_out/<tag>/
TAG=${1:-x2cfg}
MODE=${MODE:-fp16x2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="python bench.py --precision $MODE --head-precision bf16x3 --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 10"
$B > $OUT/x2_cfg2.json 2>> $OUT/bench.err
$B --shots 5 --batch 16 > $OUT/x2_cfg4_5shot_b16.json 2>> $OUT/bench.err
$B --arch dinov2_vitl14 --image-size 384 --batch 8 --no-episode > $OUT/x2_cfg5_vitl_384_b8.json 2>> $OUT/bench.err
$B --arch dinov2_vits14 --image-size 224 > $OUT/x2_ref_vits_224_b32.json 2>> $OUT/bench.err
for f in x2_cfg2 x2_cfg4_5shot_b16 x2_cfg5_vitl_384_b8 x2_ref_vits_224_b32; do python tools/bench_line.py $f < $OUT/$f.json | cut -c1-160; done
