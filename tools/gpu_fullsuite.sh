#!/bin/bash
# the whole -m gpu suite with the slowest tests listed   -> gpurun_out/suite/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/suite
mkdir -p $OUT
cd $R
T0=$(date +%s)
python -m pytest tests -m gpu -q --durations=40 > $OUT/tests.log 2>&1
echo "suite wall seconds: $(( $(date +%s) - T0 ))" >> $OUT/tests.log
tail -n 62 $OUT/tests.log
