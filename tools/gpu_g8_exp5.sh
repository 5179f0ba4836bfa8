cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -n 3
REPS=3 VARIANTS=0,64,0,64 SHAPES="qkv:20800:2304:768,proj:20800:768:768,fc2:20800:768:3072,sq4096:4096:4096:4096" python tools/g8_lab.py 2>&1 | tee gpurun_out/g8/exp5.txt
REPS=3 VARIANTS=100,164,100,164 SHAPES="fc1:20800:3072:768" python tools/g8_lab.py 2>&1 | tee -a gpurun_out/g8/exp5.txt
python bench.py --precision bf16 --no-cpu-baseline --no-episode --steps 20 > gpurun_out/g8/bench5.json; cut -c1-1400 gpurun_out/g8/bench5.json
