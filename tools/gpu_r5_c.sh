#!/bin/bash
# round 5, third GPU pass: fan-out by mask bits; compaction default A/B; episode call size sweep; kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5c
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_next_rows.py -m gpu -q -x -k "episodes_stream or one_shot_call or row_compaction or (pipelined_bit_equal and 224-4)" > $OUT/tests_new.log 2>&1; tail -n 4 $OUT/tests_new.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "golden" > $OUT/tests_model.log 2>&1; tail -n 2 $OUT/tests_model.log
for c in 2 1 2 1; do
  EC_COMPACT=$c timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --no-episode > $OUT/bench_compact$c.json 2> $OUT/bench$c.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_compact$c.json"))
print("EC_COMPACT=$c value", d["value"], "unpipelined", d["unpipelined"]["value"], "conforming", d.get("conforming_mode", {}).get("value"))
PY
done
for n in 64 128; do
  timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --no-alt --episode-images $n --steps 10 > $OUT/bench_ep$n.json 2>> $OUT/bench.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_ep$n.json"))
e = d["episode_cached"]
print("episode images/call $n: value", d["value"], "episode", e["value"], "q/call", e["queries_per_call"], "ms/call", e["ms_per_call"], "x", e["speedup_vs_value"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $R/bench.py --no-cpu-baseline --no-alt --sustained-seconds 0 --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
python $R/tools/trace_step.py $DB 0 -1 > $OUT/episode_call_trace.txt 2>$OUT/trace.err
python $R/tools/trace_step.py $DB 0 -35 > $OUT/step_trace.txt 2>>$OUT/trace.err
grep chain_kernel $OUT/step_trace.txt | cut -c1-100 | tail -30
rm -rf $OUT/prof
