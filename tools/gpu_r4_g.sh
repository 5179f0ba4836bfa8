#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
ROUNDS=12 python tools/g8_lib_ab.py tools/_lab_old.so edgecape_amd/libedgecape_hip_lab.so 2>/dev/null > $O/prologue_ab.txt; cat $O/prologue_ab.txt
SHAPES=fc1:20800:3072:768 VARIANTS=3000,3128,3000,3128 REPS=3 python tools/g8_lab.py 2>/dev/null > $O/fc1_regepi.txt; cat $O/fc1_regepi.txt
python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/ops.log 2>&1; tail -n 3 $O/ops.log
python -m pytest tests/test_gpu_next_rows.py tests/test_gpu_model.py -m gpu -q -x > $O/model.log 2>&1; tail -n 5 $O/model.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-episode --no-alt --steps 30 2>/dev/null | python tools/bench_line.py pipelined | cut -c1-120
  python bench.py --no-cpu-baseline --no-episode --no-alt --no-pipeline --steps 30 2>/dev/null | python tools/bench_line.py ec_forward | cut -c1-120
done > $O/bench3.txt; cat $O/bench3.txt
