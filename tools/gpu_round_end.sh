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
# rocprofv3 kernel traces, ONE LEG PER RUN, statistics over the leg's timed steps only (tools/rocpd_stats.py <db> <csv> <last steps>: the
# engine's build, the weight uploads and the warm-up steps lie in front of them and are cut off; per-step columns in the CSV)
prof_leg () {   # name, last steps, bench arguments...
  local name=$1 steps=$2; shift 2
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-alt --sustained-seconds 0 "$@" > $OUT/prof_${name}.json 2> $OUT/prof_${name}.err
  cd $R
  local DB=$(ls $OUT/prof_$name/*/*results.db $OUT/prof_$name/*results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $DB $OUT/kernel_stats_${name}.csv $steps
  python tools/trace_step.py $DB 0 -1 > $OUT/step_trace_${name}.txt 2>/dev/null          # the leg's last step / call, launch by launch
  head -n 10 $OUT/kernel_stats_${name}.csv | cut -c1-150
  rm -rf $OUT/prof_$name
}
prof_leg headline 6 --no-episode --steps 6 --warmup 3                                     # 6 pipelined headline steps (fp16 / mixed)
prof_leg episodes 48 --episode-images 64 --steps 1 --warmup 0                             # 6 passes x 8 ec_forward_episodes calls (60 queries + 4 supports)
prof_leg conforming 6 --no-episode --steps 6 --warmup 3 --precision fp16x2 --head-precision bf16x3   # the conforming mode's kernels
rm -rf $OUT/pmc/pmc_*
ls $OUT
