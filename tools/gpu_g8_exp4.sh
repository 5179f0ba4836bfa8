cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
REPS=4 VARIANTS=0,64,0,64 SHAPES="qkv:20800:2304:768,proj:20800:768:768,fc2:20800:768:3072,sq4096:4096:4096:4096" python tools/g8_lab.py 2>&1 | tee gpurun_out/g8/exp4.txt
REPS=4 VARIANTS=100,164,100,164 SHAPES="fc1:20800:3072:768" python tools/g8_lab.py 2>&1 | tee -a gpurun_out/g8/exp4.txt
