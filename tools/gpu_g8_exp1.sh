cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
S=""
for K in 768 1536 3072; do for M in 1024 3584 7168 14336 20800; do S="$S,m${M}k${K}:$M:2304:$K"; done; done
S=${S#,}
NOCHECK=1 ITERS=50 SHAPES="$S" python tools/gemm_bench.py bf16 2>&1 | tee gpurun_out/g8/exp1.txt
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "fp16" 2>&1 | tail -3
