#!/bin/bash
# Round-4 fourth set: A/B of the epilogue changes (FP16_OVFL saturation, accumulators started from the bias), op tests, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $O/ops.log 2>&1; tail -n 5 $O/ops.log
ROUNDS=12 python tools/g8_lib_ab.py tools/_lab_old.so edgecape_amd/libedgecape_hip_lab.so 2>/dev/null > $O/epi_ab.txt; cat $O/epi_ab.txt
python tools/g8_ledger.py qkv fc1 2>/dev/null > $O/ledger.txt; grep -v "XCD" $O/ledger.txt | head -40
python bench.py > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json | cut -c1-300
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_ops.py > $O/tests.log 2>&1; tail -n 8 $O/tests.log
