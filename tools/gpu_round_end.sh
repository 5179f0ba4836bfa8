#!/bin/bash
# Round-end measurement set on one MI355X (run through gpurun): tests, headline bench line, other configurations, rocprofv3 kernel
# trace + stats, PMC refresh of the north-star kernel.  usage: bash tools/gpu_round_end.sh <tag>   -> gpurun_out/<tag>/...
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -n 3 $OUT/tests.log
python bench.py > $OUT/bench_prepmc.json 2> $OUT/bench.err   # (before the PMC refresh below: its traffic field may be stale; the judged line is re-run at the end)
# BASELINE.json configs 4 and 5 and the reference's shipped configuration (same binary; parity-test cases, not the bench line)
python bench.py --shots 5 --batch 16 --no-cpu-baseline --no-episode --no-alt --steps 10 > $OUT/cfg4_5shot_b16.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vitl14 --image-size 384 --batch 8 --no-cpu-baseline --no-episode --no-alt --steps 10 > $OUT/cfg5_vitl_384_b8.json 2>> $OUT/bench.err
python bench.py --arch dinov2_vits14 --image-size 224 --no-cpu-baseline --no-episode --no-alt --steps 10 > $OUT/ref_vits_224_b32.json 2>> $OUT/bench.err
python bench.py --precision bf16x3 --no-cpu-baseline --no-episode --no-alt --steps 10 > $OUT/bench_bf16x3.json 2>> $OUT/bench.err
for f in cfg4_5shot_b16 cfg5_vitl_384_b8 ref_vits_224_b32 bench_bf16x3; do python tools/bench_line.py $f < $OUT/$f.json | cut -c1-160; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $R/bench.py --no-cpu-baseline --no-episode --no-alt --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
cd $R
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
head -n 12 $OUT/kernel_stats.csv | cut -c1-150
python tools/refresh_pmc.py --out $OUT/pmc > $OUT/pmc.log 2>&1; tail -n 12 $OUT/pmc.log
ls $OUT
# the headline line again, now that profiles/qkv_gemm_pmc.json on this box matches the library (copy $OUT/pmc/qkv_gemm_pmc.json into profiles/ first)
cp $OUT/pmc/qkv_gemm_pmc.json profiles/qkv_gemm_pmc.json
python bench.py > $OUT/bench.json 2>> $OUT/bench.err; python tools/bench_line.py < $OUT/bench.json | cut -c1-400
