# usage: bash tools/gpu_prof2.sh <tag> [bench args...]  -> kernel-trace stats of a short bench run under gpurun_out/<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python $R/bench.py --no-cpu-baseline --no-episode --no-alt --steps 6 --warmup 3 "$@" > $OUT/prof_bench.json 2> $OUT/prof.err
cd $R
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.reader(open("$OUT/kernel_stats.csv")))
for r in rows[1:14]: print(r[0][:60].ljust(60), r[1].rjust(5), f"{int(r[2])/9/1e3:9.1f} us/step", f"{float(r[3])/1e3:8.1f} avg", r[5], r[6])
PY
rm -rf $OUT/prof
