cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
SHAPES="qkv:20800:2304:768,one252:7168:2304:768,one36:1024:2304:768,one252k3072:7168:2304:3072,one36k3072:1024:2304:3072,fc2:20800:768:3072" python tools/g8_lab.py 2>&1 | tee gpurun_out/g8/exp2.txt
