cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
VARIANTS=0 NT=0,1 SHAPES="qkv:20800:2304:768,fc2:20800:768:3072,sq4096:4096:4096:4096,one252k3072:7168:2304:3072" python tools/g8_lab.py 2>&1 | tee gpurun_out/g8/exp3.txt
