#!/bin/bash
# Round-4 sixth set: ViT-S/14 @ 224 tile experiments (the reference's shipped configuration), outlier-statistics test, full tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_precision_modes.py -m gpu -q -x -k "outlier" -s > $O/outlier.log 2>&1; grep -E "planted|outlier statistics|bf16x3|passed|failed" $O/outlier.log | cut -c1-700
B="python bench.py --arch dinov2_vits14 --image-size 224 --no-cpu-baseline --no-episode --no-alt --steps 30"
for i in 1 2; do
  $B 2>/dev/null | python tools/bench_line.py vits_default | cut -c1-110
  EC_GEMM8_OFF=1 $B 2>/dev/null | python tools/bench_line.py vits_gemm8off | cut -c1-110
  EC_GEMM8_OFF=1 EC_GEMM_TILE=128 $B 2>/dev/null | python tools/bench_line.py vits_gemm8off_t128 | cut -c1-110
  EC_GEMM8_OFF=1 EC_GEMM_TILE=256128 $B 2>/dev/null | python tools/bench_line.py vits_gemm8off_t256128 | cut -c1-110
done > $O/vits_tiles.txt; cat $O/vits_tiles.txt
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -n 6 $O/tests.log
