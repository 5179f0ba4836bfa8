#!/bin/bash
# quick check of the bf16x3 / bf16x3 mode: attention + backbone tests, two bench runs, kernel stats   -> gpurun_out/<tag>/
TAG=${1:-x3q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="python bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-episode --no-alt --sustained-seconds 0 --steps 10"
timeout 900 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "(parity_mode_bf16x3 and (cfg1 or cfg2)) or kconcat or attention" > $OUT/tests.log 2>&1; tail -n 3 $OUT/tests.log
for i in 1 2; do $B > $OUT/bench_$i.json 2>> $OUT/bench.err; python tools/bench_line.py x < $OUT/bench_$i.json | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $R/bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-episode --no-alt --sustained-seconds 0 --steps 4 --warmup 2 > $OUT/prof.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
head -n 9 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof
