#!/bin/bash
# round 5, second GPU pass: new tests; bench default vs EC_COMPACT=1 (plain calls without compaction); kernel trace of the headline
# step and of an episode call.   usage: bash tools/gpu_r5_b.sh  -> gpurun_out/r5b/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_next_rows.py -m gpu -q -x -k "episodes_stream or one_shot_call or row_compaction or pipelined_bit_equal or pipelined_stress" --durations=10 > $OUT/tests_new.log 2>&1; tail -n 14 $OUT/tests_new.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_errors.py -m gpu -q -x -k "not switch_matrix" > $OUT/tests_model.log 2>&1; tail -n 3 $OUT/tests_model.log
for c in 2 1; do
  EC_COMPACT=$c timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > $OUT/bench_compact$c.json 2> $OUT/bench$c.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_compact$c.json"))
print("EC_COMPACT=$c value", d["value"], "unpipelined", d["unpipelined"]["value"], "episode", d["episode_cached"]["value"], "conforming", d.get("conforming_mode", {}).get("value"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $R/bench.py --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
python $R/tools/trace_step.py $DB 0 -1 > $OUT/episode_call_trace.txt 2>$OUT/trace.err
python $R/tools/trace_step.py $DB 0 -40 > $OUT/step_trace.txt 2>>$OUT/trace.err
tail -n 45 $OUT/episode_call_trace.txt
rm -rf $OUT/prof
