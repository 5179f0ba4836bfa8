#!/bin/bash
# Round-4 second measurement set: GPU tests, bench line, cycle ledger of the 8-phase GEMM, lab ablations, saturation-cost A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
python tools/g8_ledger.py > $O/ledger.txt 2>&1; head -n 40 $O/ledger.txt
ZERO=1 python tools/g8_ledger.py qkv sq4096 > $O/ledger_zero.txt 2>&1
VARIANTS=1000,1001,1008,2024 REPS=3 python tools/g8_lab.py > $O/lab_qkv.txt 2>&1; cat $O/lab_qkv.txt
python tools/g8_lib_ab.py tools/_lab_old.so edgecape_amd/libedgecape_hip_lab.so > $O/sat_ab.txt 2>&1; cat $O/sat_ab.txt
python bench.py > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json | cut -c1-600
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -n 15 $O/tests.log
