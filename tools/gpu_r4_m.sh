#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
python tools/g8_ledger.py qkv fc1 proj fc2 sq4096 2>/dev/null > $O/ledger.txt; grep "^==" $O/ledger.txt | grep traced
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -n 4 $O/tests.log
bash tools/gpu_round_end.sh r04m --no-tests
