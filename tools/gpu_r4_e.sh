#!/bin/bash
# Round-4 fifth set: NaN propagation with FP16_OVFL confined to the epilogue, row compaction of the token chains (tests + A/B), bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
python tools/_nan_debug.py > $O/nan_debug.txt 2>&1; cat $O/nan_debug.txt | tail -8
python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/ops.log 2>&1; tail -n 4 $O/ops.log
python -m pytest tests/test_gpu_next_rows.py -m gpu -q -x -k "compaction or pipelined_bit_equal or submit" > $O/compact.log 2>&1; tail -n 6 $O/compact.log
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "golden or full_size or cache" > $O/model.log 2>&1; tail -n 6 $O/model.log
SHAPES=qkv,fc1 ROUNDS=10 python tools/g8_lib_ab.py tools/_lab_old.so edgecape_amd/libedgecape_hip_lab.so 2>/dev/null > $O/ovfl_toggle_ab.txt; cat $O/ovfl_toggle_ab.txt
for i in 1 2 3; do
  EC_COMPACT=0 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 30 2>/dev/null | python tools/bench_line.py compact0 | cut -c1-120
  python bench.py --no-cpu-baseline --no-episode --no-alt --steps 30 2>/dev/null | python tools/bench_line.py compact1 | cut -c1-120
done > $O/compact_ab.txt; cat $O/compact_ab.txt
for i in 1 2; do
  EC_COMPACT=0 python bench.py --no-cpu-baseline --no-episode --no-alt --no-pipeline --steps 30 2>/dev/null | python tools/bench_line.py nopipe_compact0 | cut -c1-120
  python bench.py --no-cpu-baseline --no-episode --no-alt --no-pipeline --steps 30 2>/dev/null | python tools/bench_line.py nopipe_compact1 | cut -c1-120
done >> $O/compact_ab.txt; tail -4 $O/compact_ab.txt
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -n 8 $O/tests.log
