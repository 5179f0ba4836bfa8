#!/bin/bash
# round 5, first GPU pass: new episode-stream tests, the ADVICE regression test, attention XCD map A/B (kernel trace + FETCH_SIZE),
# bench with the streaming episode leg.   usage: bash tools/gpu_r5_a.sh  -> gpurun_out/r5a/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_next_rows.py -m gpu -q -x -k "episodes_stream or one_shot_call or support_cache or row_compaction" --durations=10 > $OUT/tests_new.log 2>&1; tail -n 15 $OUT/tests_new.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "attention or backbone_vs or support_cache_matches" > $OUT/tests_attn.log 2>&1; tail -n 3 $OUT/tests_attn.log
timeout 600 python bench.py --no-cpu-baseline --no-alt > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py < $OUT/bench.json | cut -c1-400
timeout 600 python bench.py --shots 5 --batch 16 --no-cpu-baseline --no-alt --steps 10 > $OUT/cfg4.json 2>> $OUT/bench.err; python tools/bench_line.py cfg4 < $OUT/cfg4.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  EC_ATTN_PLAIN=$mode timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$mode -o r -- python $R/bench.py --no-cpu-baseline --no-episode --no-alt --steps 6 --warmup 3 > $OUT/prof_bench$mode.json 2> $OUT/prof$mode.err
  DB=$(ls $OUT/prof$mode/*/*results.db $OUT/prof$mode/*results.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py $DB $OUT/kernel_stats_plain$mode.csv
  grep -E "attn_bf16|gemm8_bf16_kernel<1, 1" $OUT/kernel_stats_plain$mode.csv | cut -c1-160
  EC_ATTN_PLAIN=$mode timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc$mode -o r -- python $R/bench.py --no-cpu-baseline --no-episode --no-alt --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc$mode.err
  DB=$(ls $OUT/pmc$mode/*/*results.db $OUT/pmc$mode/*results.db 2>/dev/null | head -1)
  python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import rocpd_pmc
for (kn, cn), (n, v, dur) in sorted(rocpd_pmc.summarise("$DB").items()):
    if "attn_bf16" in kn or "gemm8_bf16_kernel<1, 1" in kn:
        print("plain=$mode", rocpd_pmc.short(kn)[:60], cn, n, "mean", round(v / n, 1), "KiB  x2 ->", round(2 * v / n * 1024 / 1e6, 1), "MB ; dur us", round(dur / n / 1e3, 1))
PY
  rm -rf $OUT/prof$mode $OUT/pmc$mode
done
ls $OUT
