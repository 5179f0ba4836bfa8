cd $GRAFT_REPO_ROOT
O=gpurun_out/g8b; mkdir -p $O
VARIANTS=0,128,0,128 REPS=3 SHAPES="qkv:20800:2304:768,proj:20800:768:768,fc2:20800:768:3072,sq4096:4096:4096:4096" timeout 300 python tools/g8_lab.py 2>&1 | tee $O/lab1.txt | tail -n 13
VARIANTS=100,228,100,228 REPS=3 SHAPES="fc1:20800:3072:768" timeout 300 python tools/g8_lab.py 2>&1 | tee $O/lab2.txt | tail -n 3
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "linear or gemm8" 2>&1 | tail -n 5
python -m pytest tests/test_gpu_model.py tests/test_gpu_precision_modes.py -m gpu -q -x 2>&1 | tail -n 5
python bench.py --no-cpu-baseline --no-episode --steps 20 > $O/bench.json 2>/dev/null; python tools/bench_line.py < $O/bench.json
