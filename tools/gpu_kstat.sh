# usage: bash tools/gpu_kstat.sh <tag> <kernel-name regex> [ENV=VAL ...] : rocprofv3 kernel stats of a short bench run, rows matching the regex
TAG=$1; RE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python $R/bench.py --no-cpu-baseline --no-episode --no-alt --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
cd $R
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
grep -E "$RE" $OUT/kernel_stats.csv | awk -F, '{n=NF; printf "%-90s calls %s avg %.1f us min %.1f\n", substr($0,1,90), $(n-5), $(n-3)/1000, $(n-1)/1000}'
rm -rf $OUT/prof
