#!/bin/bash
# rocprofv3 kernel-trace summaries of the three bench legs (called by tools/gpu_round_end.sh)   usage: bash tools/gpu_profile_legs.sh <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
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
