#!/bin/bash
# K-concatenated bf16x3 backbone (EC_BB_X3, round 5): parity tests, interleaved bench A/B against the split-on-load kernel, kernel stats.
# (profiles/r05_x3_ab.txt also has a "gen" leg - the generic epilogue behind a switch, EC_G8_X3EPI, that is gone from the library again)
#   usage: bash tools/gpu_x3_ab.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-x3a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="python bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-episode --no-alt --sustained-seconds 0 --steps 10"
if [ "$2" != "--no-tests" ]; then
  timeout 900 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "(parity_mode_bf16x3 and (cfg1 or cfg2)) or backbone_vs or kconcat or linear or attention" > $OUT/tests.log 2>&1; tail -n 5 $OUT/tests.log
fi
for i in 1 2; do
  EC_BB_X3=0 $B > $OUT/bench_x3off_$i.json 2>> $OUT/bench.err
  $B > $OUT/bench_x3on_$i.json 2>> $OUT/bench.err
done
for f in $OUT/bench_x3*.json; do echo $f; python tools/bench_line.py x < $f | cut -c1-250; done
cd /tmp && export TMPDIR=/tmp
for v in on; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o r -- python $R/bench.py --precision bf16x3 --head-precision bf16x3 --no-cpu-baseline --no-episode --no-alt --sustained-seconds 0 --steps 4 --warmup 2 > $OUT/prof_$v.json 2> $OUT/prof_$v.err
  DB=$(ls $OUT/prof_$v/*/*results.db $OUT/prof_$v/*results.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_$v.csv
  head -n 16 $OUT/kernel_stats_$v.csv | cut -c1-180
  rm -rf $OUT/prof_$v
done
