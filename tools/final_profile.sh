#!/bin/bash
# Round-end measurement set on one MI355X (run through gpurun): tests, bench line, rocprofv3 kernel trace, PMC passes.
# usage: bash tools/final_profile.sh <tag>     -> gpurun_out/<tag>/...
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_line.py < $OUT/bench.json
python bench.py --precision fp32 --head-precision fp32 --steps 5 --warmup 2 --no-episode > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err; python tools/bench_line.py < $OUT/bench_fp32.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-episode"
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- $B --steps 6 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o r1 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_w -o r1 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_s -o r1 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
ls $OUT
