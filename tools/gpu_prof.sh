# usage: bash tools/gpu_prof.sh <tag> [env assignments...]  -> kernel-trace stats of a short bench run under gpurun_out/<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python $R/bench.py --no-cpu-baseline --no-episode --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
cd $R
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.reader(open("$OUT/kernel_stats.csv")))
print(rows[0])
for r in rows[1:26]: print(r[0][:70].ljust(70), r[1].rjust(5), f"{int(r[2])/9/1e3:9.1f} us/step", f"{float(r[3])/1e3:8.1f} avg", r[5], r[6])
PY
