#!/bin/bash
# Round-end measurement set on one MI355X (run through gpurun): tests, PMC refresh of the north-star kernel for every benched
# configuration, the bench lines (headline + BASELINE configs 4 / 5 + the reference's shipped ViT-S configuration, each WITH the CPU
# baseline / parity sample and with `traffic` from the PMC summary of the same library), rocprofv3 kernel trace + stats.
#   usage: bash tools/gpu_round_end.sh <tag> [--no-tests]   -> gpurun_out/<tag>/...   (copy what should be judged into profiles/)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT $OUT/pmc
cd $R
if [ "$2" != "--no-tests" ]; then python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -n 3 $OUT/tests.log; fi
# PMC first: bench.py only reports `traffic` from a summary whose source hash equals the running library's
python tools/refresh_pmc.py --out $OUT/pmc > $OUT/pmc.log 2>&1; tail -n 3 $OUT/pmc.log
python tools/refresh_pmc.py --out $OUT/pmc --shots 5 --batch 16 >> $OUT/pmc.log 2>&1
python tools/refresh_pmc.py --out $OUT/pmc --arch dinov2_vitl14 --image-size 384 --batch 8 >> $OUT/pmc.log 2>&1
python tools/refresh_pmc.py --out $OUT/pmc --arch dinov2_vits14 --image-size 224 >> $OUT/pmc.log 2>&1
python tools/refresh_pmc.py --out $OUT/pmc --precision fp16x2 >> $OUT/pmc.log 2>&1
cp $OUT/pmc/qkv_gemm_pmc*.json profiles/
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py < $OUT/bench.json | cut -c1-400
python bench.py --shots 5 --batch 16 --no-alt --steps 10 --cpu-batches 16 --cpu-runs 3 > $OUT/cfg4_5shot_b16.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vitl14 --image-size 384 --batch 8 --no-episode --no-alt --steps 10 --cpu-batches 8 --cpu-runs 3 > $OUT/cfg5_vitl_384_b8.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vits14 --image-size 224 --no-alt --steps 10 --cpu-batches 32 --cpu-runs 3 > $OUT/ref_vits_224_b32.json 2>> $OUT/bench.err
python bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 10 > $OUT/bench_bf16x3.json 2>> $OUT/bench.err
python bench.py --precision fp16x2 --head-precision bf16x3 --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 10 > $OUT/bench_fp16x2.json 2>> $OUT/bench.err
for f in cfg4_5shot_b16 cfg5_vitl_384_b8 ref_vits_224_b32 bench_bf16x3 bench_fp16x2; do python tools/bench_line.py $f < $OUT/$f.json | cut -c1-200; done
bash tools/gpu_profile_legs.sh $TAG
rm -rf $OUT/pmc/pmc_*
ls $OUT
