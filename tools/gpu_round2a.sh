set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "fp16 or gemm8_race or linear_bf16" > gpurun_out/r2a/ops.log 2>&1; echo "ops rc $?"
python -m pytest tests/test_gpu_precision_modes.py -m gpu -q -s > gpurun_out/r2a/modes.log 2>&1; echo "modes rc $?"
for p in bf16 fp16 bf16x3; do
  python bench.py --precision $p --no-cpu-baseline --no-episode --steps 10 > gpurun_out/r2a/bench_$p.json 2> gpurun_out/r2a/bench_$p.err; echo "bench $p rc $?"
done
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/all.log 2>&1; echo "all rc $?"
tail -5 gpurun_out/r2a/*.log; cat gpurun_out/r2a/bench_*.json
